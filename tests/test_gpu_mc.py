"""GPU parity of the HIP marching cubes (K3-K6) against the skimage goldens and the CPU oracle:
identical triangle and vertex counts, identical faces, bit-identical vertices."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu

CORNER_POS = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]


def hip_mc(vol, level=0.0, spacing=(1.0, 1.0, 1.0)):
    from alignsdf_amd.marching_cubes import marching_cubes_lewiner
    return marching_cubes_lewiner(torch.as_tensor(np.ascontiguousarray(vol, dtype=np.float32)).cuda(), level, spacing)


def test_all_sign_patterns_vs_skimage(golden_dir):
    """ALL 5 080 fixture cells (254 non-trivial sign patterns x 20 magnitude draws) through the real kernel chain, one
    2 x 2 x 2 volume each: vertex and face arrays bit for bit those of the installed skimage binary."""
    g = {k: v for k, v in np.load(golden_dir + "/mc_cells.npz").items()}        # decompress once, not per access
    for n in range(len(g["corners"])):
        vol = np.zeros((2, 2, 2), np.float32)
        for ci, (z, y, x) in enumerate(CORNER_POS):
            vol[z, y, x] = g["corners"][n][ci]
        v, f = hip_mc(vol)
        V, F = int(g["V"][n]), int(g["F"][n])
        assert (len(v), len(f)) == (V, F), n
        assert np.array_equal(f, g["faces"][n, :F]) and np.array_equal(v, g["verts"][n, :V]), n


def test_value_dependent_patterns_vs_skimage(golden_dir):
    """ALL 19 200 draws on the value-dependent patterns (MC33 sub-cases): counts and checksums of the skimage binary."""
    g = {k: v for k, v in np.load(golden_dir + "/mc_cells_ambiguous.npz").items()}
    for n in range(len(g["corners"])):
        vol = np.zeros((2, 2, 2), np.float32)
        for ci, (z, y, x) in enumerate(CORNER_POS):
            vol[z, y, x] = g["corners"][n][ci]
        v, f = hip_mc(vol)
        assert (len(v), len(f)) == (int(g["V"][n]), int(g["F"][n])), n
        w = np.arange(1, f.size + 1, dtype=np.int64)
        assert int((f.reshape(-1).astype(np.int64) * w).sum()) == int(g["fsum"][n]), n
        assert float(v.astype(np.float64).sum()) == float(g["vsum"][n]), n


def test_volumes_vs_skimage_bit_exact(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    names = sorted({k.split(".")[0] for k in g.files if k.endswith(".verts")})
    for nm in names:
        sp = g[nm + ".spacing"]
        sp = tuple(sp.astype(np.float32)) if nm.startswith("dec_") else tuple(float(s) for s in sp)
        v, f = hip_mc(g[nm + ".vol"], float(g[nm + ".level"][0]), sp)
        assert v.dtype == g[nm + ".verts"].dtype, nm
        assert f.shape == g[nm + ".faces"].shape and v.shape == g[nm + ".verts"].shape, nm
        assert np.array_equal(f, g[nm + ".faces"]), nm
        assert np.array_equal(v, g[nm + ".verts"]), nm


def test_noise48_checksum_vs_skimage(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    v, f = hip_mc(syn.uniform((48, 48, 48), 31400, -1.0, 1.0).astype(np.float32))
    assert (len(v), len(f)) == (int(g["noise48.V"][0]), int(g["noise48.F"][0]))
    w = np.arange(1, f.size + 1, dtype=np.int64)
    assert int((f.reshape(-1).astype(np.int64) * w).sum()) == int(g["noise48.fsum"][0])
    assert float(v.astype(np.float64).sum()) == float(g["noise48.vsum"][0])


def test_failure_modes(golden_dir):
    g = np.load(golden_dir + "/mc_volumes.npz")
    with pytest.raises(ValueError, match="Surface level must be within volume data range"):
        hip_mc(g["fail_allpos.vol"], float(g["fail_allpos.level"][0]))
    with pytest.raises(RuntimeError, match="No surface found"):
        hip_mc(g["fail_level_is_max.vol"], float(g["fail_level_is_max.level"][0]))
    with pytest.raises(ValueError):
        hip_mc(np.zeros((1, 4, 4), np.float32))


@pytest.mark.parametrize("shape", [(64, 64, 64), (33, 70, 129), (128, 128, 128)])
def test_large_volumes_vs_oracle(shape):
    """Bigger / non-cubic volumes against the sequential oracle: smooth field + noise (ambiguous cells)."""
    from oracle import mc33
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij"), -1)
    smooth = np.sin(4.0 * g[..., 0]) * np.cos(3.0 * g[..., 1]) + 0.5 * np.sin(6.0 * g[..., 2] + 1.0)
    vol = (smooth + 0.3 * syn.uniform(shape, 123 + shape[0], -1.0, 1.0)).astype(np.float32)
    v, f = hip_mc(vol, 0.05)
    rv, rf = mc33.marching_cubes_raw(vol, 0.05)
    assert v.shape == rv.shape and f.shape == rf.shape
    assert np.array_equal(f, rf) and np.array_equal(v, rv)
    # mesh sanity: every vertex is used, every face index is valid
    assert f.min() == 0 and f.max() == len(v) - 1 and len(np.unique(f)) == len(v)


def test_full_size_closed_surface_properties():
    """N=256 (BASELINE size): an analytic sphere must come out closed (V - E + F = 2) with F = 2V - 4."""
    n = 256
    ax = torch.linspace(-1, 1, n, device="cuda")
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = torch.sqrt(zz * zz + yy * yy + xx * xx) - 0.63
    from alignsdf_amd.marching_cubes import marching_cubes_device
    v, f = marching_cubes_device(vol, 0.0)
    V, F = v.shape[0], f.shape[0]
    assert F == 2 * V - 4 and F > 100000
    fl = f.long()
    e = torch.cat([fl[:, [0, 1]], fl[:, [1, 2]], fl[:, [2, 0]]]).sort(1).values
    assert torch.unique(e, dim=0).shape[0] * 2 == 3 * F       # every edge shared by exactly two faces
    r = ((v / (n - 1) * 2 - 1) ** 2).sum(1).sqrt()
    assert (r - 0.63).abs().max().item() < 2.0 / (n - 1)
