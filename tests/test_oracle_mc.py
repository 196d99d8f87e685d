"""Pin the sequential MC33 oracle (oracle/mc33_oracle.c) against the installed skimage 0.18.3 binary's
outputs (tests/golden/mc_*.npz, made by tests/golden/make_mc_goldens.py): bit-exact verts and faces."""
import numpy as np
import pytest

from alignsdf_amd import synthetic as syn
from oracle import mc33

CORNER_POS = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]


def cell_volume(c):
    vol = np.zeros((2, 2, 2), np.float32)
    for ci, (z, y, x) in enumerate(CORNER_POS):
        vol[z, y, x] = c[ci]
    return vol


def checksums(v, f):
    w = np.arange(1, f.size + 1, dtype=np.int64)
    return int((f.reshape(-1).astype(np.int64) * w).sum()), float(v.astype(np.float64).sum())


def test_all_sign_patterns_bit_exact(golden_dir):
    """All 5 080 fixture cells (254 sign patterns x 20 magnitude draws), verts and faces bit for bit."""
    g = {k: v for k, v in np.load(golden_dir + "/mc_cells.npz").items()}        # decompress once, not per access
    for n in range(len(g["corners"])):
        v, f = mc33.marching_cubes_raw(cell_volume(g["corners"][n]))
        V, F = int(g["V"][n]), int(g["F"][n])
        assert (len(v), len(f)) == (V, F), n
        assert np.array_equal(f, g["faces"][n, :F]) and np.array_equal(v, g["verts"][n, :V]), n


def test_value_dependent_patterns(golden_dir):
    """ALL 19 200 extra draws on the 128 value-dependent sign patterns: counts and checksums of the installed binary."""
    g = {k: v for k, v in np.load(golden_dir + "/mc_cells_ambiguous.npz").items()}
    seen = set()
    for n in range(len(g["corners"])):
        v, f = mc33.marching_cubes_raw(cell_volume(g["corners"][n]))
        assert (len(v), len(f)) == (int(g["V"][n]), int(g["F"][n])), n
        fs, vs = checksums(v, f)
        assert fs == int(g["fsum"][n]) and vs == float(g["vsum"][n]), n
        seen.add(len(f))
    assert {2, 3, 4, 5, 6, 8, 9, 10, 12} <= seen   # the MC33-only triangle counts are exercised


def volume_names(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    return g, sorted({k.split(".")[0] for k in g.files if k.endswith(".verts")})


def test_volumes_bit_exact(golden_dir):
    g, names = volume_names(golden_dir)
    assert len(names) >= 19
    for nm in names:
        sp = g[nm + ".spacing"]
        sp = tuple(sp.astype(np.float32)) if nm.startswith("dec_") else tuple(float(s) for s in sp)
        v, f = mc33.marching_cubes_lewiner(g[nm + ".vol"], float(g[nm + ".level"][0]), sp)
        assert v.dtype == g[nm + ".verts"].dtype, nm
        assert np.array_equal(f, g[nm + ".faces"]), nm
        assert np.array_equal(v, g[nm + ".verts"]), nm


def test_large_noise_volume_checksum(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    vol = syn.uniform((48, 48, 48), 31400, -1.0, 1.0).astype(np.float32)
    v, f = mc33.marching_cubes_raw(vol)
    assert (len(v), len(f)) == (int(g["noise48.V"][0]), int(g["noise48.F"][0]))
    fs, vs = checksums(v, f)
    assert fs == int(g["noise48.fsum"][0]) and vs == float(g["noise48.vsum"][0])


def test_failure_modes_match_skimage(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    with pytest.raises(ValueError, match="Surface level must be within volume data range"):
        mc33.marching_cubes_lewiner(g["fail_allpos.vol"], float(g["fail_allpos.level"][0]))
    with pytest.raises(RuntimeError, match="No surface found"):
        mc33.marching_cubes_lewiner(g["fail_level_is_max.vol"], float(g["fail_level_is_max.level"][0]))
    assert str(g["fail_allpos.error"]).startswith("ValueError") and str(g["fail_level_is_max.error"]).startswith("RuntimeError")
