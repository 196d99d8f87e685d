"""Randomised GPU-vs-oracle parity: many small volumes of odd shapes for marching cubes (bit-exact), random lattices
for the decoder (1e-5).  Seeds are fixed (repo-local generator), so failures reproduce."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _volume(case):
    dims = 2 + (syn.splitmix64(np.arange(3, dtype=np.uint64), 4000 + case) % np.uint64(19)).astype(int)
    kind = case % 5
    raw = syn.uniform(tuple(dims), 5000 + case, -1.0, 1.0)
    if kind == 0:
        vol = raw                                             # white noise: every MC33 case, heavy ambiguity
    elif kind == 1:
        vol = np.round(raw * 3.0)                             # quantised: exact zeros, ties, degenerate triangles
    elif kind == 2:
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, d) for d in dims], indexing="ij"), -1)
        vol = np.linalg.norm(g * (0.6 + syn.uniform((3,), 6000 + case)), axis=-1) - 0.5 + 0.05 * raw
    elif kind == 3:
        vol = raw * 10.0 ** syn.uniform(tuple(dims), 7000 + case, -3.0, 1.0)   # wide dynamic range
    else:
        vol = np.where(raw > 0.3, raw, -np.abs(raw) * 1e-3)   # thin negative sheets
    level = 0.0 if case % 3 else float(syn.uniform((1,), 8000 + case, -0.3, 0.3)[0])
    return vol.astype(np.float32), level


def test_marching_cubes_fuzz_vs_oracle():
    from alignsdf_amd.marching_cubes import marching_cubes_lewiner
    from oracle import mc33
    checked = 0
    for case in range(160):
        vol, level = _volume(case)
        try:
            rv, rf = mc33.marching_cubes_raw(vol, level)
            ref_err = None
        except (ValueError, RuntimeError) as e:
            ref_err = type(e)
        if ref_err is not None:
            with pytest.raises(ref_err):
                marching_cubes_lewiner(torch.from_numpy(vol).cuda(), level)
            continue
        v, f = marching_cubes_lewiner(torch.from_numpy(vol).cuda(), level)
        assert v.shape == rv.shape and f.shape == rf.shape, (case, vol.shape, v.shape, rv.shape)
        assert np.array_equal(f, rf) and np.array_equal(v, rv), (case, vol.shape)
        checked += 1
    assert checked > 120


def test_decoder_fuzz_vs_oracle():
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from oracle import sdf_oracle as orc
    specs, sd = syn.specs_for("nerf3"), syn.full_state_dict("nerf3")
    dec = HipSdfDecoder(sd, 256, 3, "nerf", device="cuda:0")
    for case in range(12):
        lat = torch.from_numpy(syn.latent_code(20 + case))
        dec.set_sample(lat)
        N = int(3 + syn.splitmix64(np.arange(1, dtype=np.uint64), 900 + case)[0] % np.uint64(30))
        origin = syn.uniform((3,), 910 + case, -1.2, 0.2).astype(np.float32)
        vs = float(np.float32(syn.uniform((1,), 920 + case, 0.005, 0.08)[0]))
        mode = case % 2
        h, o, bbox = dec.decode_grid(N, origin, vs, grid_mode=mode)
        c = orc.grid_coords(N, torch.tensor(vs), torch.from_numpy(origin), integer_mode=bool(mode))
        rh, ro = orc.decode_points(sd, lat, c, specs)
        assert (h.cpu().reshape(-1) - rh).abs().max().item() <= 1e-5 and (o.cpu().reshape(-1) - ro).abs().max().item() <= 1e-5
        b = bbox.cpu().numpy()
        for k, vol in enumerate((h, o)):
            nz = torch.nonzero(vol < 0)
            assert int(b[8 * k + 6]) == nz.shape[0]
            if nz.shape[0]:
                assert list(b[8 * k: 8 * k + 3]) == nz.min(0).values.tolist() and list(b[8 * k + 3: 8 * k + 6]) == nz.max(0).values.tolist()
