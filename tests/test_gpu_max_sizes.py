"""The largest lattices the C ABI accepts (N <= 1024; the reference's CLIs stop at 256): index arithmetic, workspace sizes and
list capacities beyond the benchmark sizes.  Size-independent properties only - grid sweep == point-list evaluation, box
count == number of negative voxels, closed surfaces (Euler characteristic 2 per component, every edge shared by two faces)."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _free_gb():
    free, _ = torch.cuda.mem_get_info()
    return free / 2 ** 30


def _closed(v, f):
    """every undirected edge of the triangle list occurs exactly twice; returns V - E + F"""
    f64 = f.long()
    e = torch.cat([f64[:, [0, 1]], f64[:, [1, 2]], f64[:, [2, 0]]], 0)
    key = torch.minimum(e[:, 0], e[:, 1]) * v.shape[0] + torch.maximum(e[:, 0], e[:, 1])
    _, counts = torch.unique(key, return_counts=True)
    assert int(counts.min()) == 2 and int(counts.max()) == 2
    return v.shape[0] - counts.numel() + f.shape[0]


def test_decoder_and_marching_cubes_at_N_512():
    if _free_gb() < 24:
        pytest.skip("needs ~24 GB of device memory")
    from alignsdf_amd import _native
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.marching_cubes import marching_cubes_device
    # (the integer lattice: beyond N = 256 the reference's own fp32 index arithmetic - utils/mesh.py:32-40, reproduced bit for
    # bit by the default grid mode - no longer resolves single voxels, 2^27 indices in 24-bit significands)
    G = _native.GRID_INTEGER
    specs = syn.specs_for("nerf3")
    hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    hip.set_sample(torch.from_numpy(syn.latent_code(2)).cuda())
    N = 512
    origin, vs = [-0.62, -0.36, -0.37], 1.21 / (N - 1)
    vh, vo, bbox = hip.decode_grid(N, origin, vs, G)
    b = bbox.cpu().numpy()
    assert b[7] == 0 and b[15] == 0 and hip.math == "f16x3"
    for vol, rec in ((vh, b[0:7]), (vo, b[8:15])):
        neg = vol < 0
        assert int(neg.sum()) == int(rec[6])
        nz = neg.nonzero()
        assert [int(x) for x in nz.min(0).values] == [int(x) for x in rec[0:3]] and [int(x) for x in nz.max(0).values] == [int(x) for x in rec[3:6]]
    # the sweep against explicit points (the fp32 kernel) at the far end of the index range
    idx = torch.from_numpy(syn.uniform((4096, 3), 77, 0.0, 1.0)).mul(N).long().clamp_(0, N - 1).cuda()
    idx[:8] = N - 1
    pts = idx.float() * np.float32(vs) + torch.tensor(origin, dtype=torch.float32, device="cuda")
    ph, po = hip.decode_points(pts)
    assert (vh[idx[:, 0], idx[:, 1], idx[:, 2]] - ph).abs().max().item() <= 4e-6
    assert (vo[idx[:, 0], idx[:, 1], idx[:, 2]] - po).abs().max().item() <= 4e-6
    # marching cubes on 134 M cells; the surfaces are clipped by nothing on this lattice: closed
    for vol in (vh, vo):
        v, f = marching_cubes_device(vol, 0.0)
        assert f.shape[0] > 200000 and int(f.max()) == v.shape[0] - 1 and int(f.min()) == 0
        assert _closed(v, f) % 2 == 0
    # the one-plane sweeps at this size: boxes equal; the band list may overflow its capacity (then the sweep is refused)
    hip.coarse_mode, hip.fine_mode = "box", "band"
    hip.coarse_finish(hip.coarse_begin(N, origin, vs, G))              # calibration
    got = hip.coarse_finish(hip.coarse_begin(N, origin, vs, G))
    assert [int(x) for x in got[0:6]] == [int(x) for x in b[0:6]] and [int(x) for x in got[8:14]] == [int(x) for x in b[8:14]]
    bh, bo, ticket = hip.fine_begin(N, origin, vs, G, mc_only=True)
    if not hip.fine_needs_repeat(ticket):
        assert int(((bh < 0) != (vh < 0)).sum()) == 0 and int(((bo < 0) != (vo < 0)).sum()) == 0
    else:
        assert hip.band_stats["fallback"] == 1
    hip.close()


def test_marching_cubes_at_the_largest_lattice():
    """1024^3 (2^30 voxels, 4 GiB of input, ~21 GiB of workspace): two spheres, closed surfaces with the right topology."""
    if _free_gb() < 60:
        pytest.skip("needs ~60 GB of device memory")
    from alignsdf_amd.marching_cubes import marching_cubes_device
    N = 1024
    ax = torch.linspace(-1, 1, N, device="cuda")
    vol = torch.full((N, N, N), 10.0, device="cuda")
    for (c0, c1, c2, r) in ((-0.4, -0.1, 0.2, 0.45), (0.55, 0.5, -0.5, 0.3)):
        d2 = ((ax - c0) ** 2)[:, None, None] + ((ax - c1) ** 2)[None, :, None] + ((ax - c2) ** 2)[None, None, :]
        vol = torch.minimum(vol, d2.sqrt_() - r)
        del d2
    v, f = marching_cubes_device(vol, 0.0)
    assert int(f.max()) == v.shape[0] - 1
    assert _closed(v, f) == 4                             # two spheres: Euler characteristic 2 each
    # areas: 4 pi r^2 in voxel units, to half a percent
    tri = v[f.long()].double()
    area = 0.5 * torch.linalg.norm(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1).sum().item()
    want = 4 * np.pi * (0.45 ** 2 + 0.3 ** 2) * ((N - 1) / 2.0) ** 2
    assert abs(area / want - 1) < 5e-3, (area, want)
    # the last voxels of the index range take part: a surface that touches the far corner
    del vol, v, f, tri
    torch.cuda.empty_cache()
