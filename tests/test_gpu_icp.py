"""GPU parity of the eval-mode ICP (K7) against the reference class' goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["small", "noisy", "full30k"])
def test_icp_vs_reference_goldens(case, golden_dir):
    from alignsdf_amd.icp import icp_trans_scale
    g = np.load(golden_dir + "/ref_icp.npz")
    r = icp_trans_scale(g[case + ".src"], g[case + ".tgt"], g[case + ".verts"])
    assert abs(r["scale"] - g[case + ".scale"][0]) <= 1e-9
    assert np.abs(r["trans"] - g[case + ".trans"]).max() <= 1e-9
    assert abs(r["all_scale"] - g[case + ".all_scale"][0]) <= 1e-9
    assert np.abs(r["all_trans"] - g[case + ".all_trans"]).max() <= 1e-9
    assert np.abs(r["vertices"] - g[case + ".verts_out"]).max() <= 1e-9


def test_icp_iteration_count_and_ragged_sizes():
    from alignsdf_amd import synthetic as syn
    from alignsdf_amd.icp import icp_trans_scale
    from oracle import icp_oracle
    for ns, nt, seed in ((2, 5, 1), (129, 1023, 2), (1025, 77, 3), (3000, 2999, 4)):
        tgt = syn.normal((nt, 3), 50 + seed) * np.array([0.1, 0.06, 0.04]) + 0.3
        src = (syn.normal((ns, 3), 60 + seed) * np.array([0.1, 0.06, 0.04]) + 0.3 - 0.02) / 1.1
        verts = syn.normal((40, 3), 70 + seed)
        r = icp_trans_scale(src, tgt, verts, max_iter=25)
        ref = icp_oracle.icp_trans_scale(src, tgt, verts, max_iter=25)
        assert r["iterations"] == ref["iterations"]
        assert abs(r["scale"] - ref["scale"]) <= 1e-9 and np.abs(r["trans"] - ref["trans"]).max() <= 1e-9
        assert abs(r["error"] - ref["errors"][-1]) <= 1e-12


def test_eval_mode_alignment_recovers_known_transform(tmp_path):
    """Mesh-level flow: OBJ ground truth, seeded surface sampling, ICP, transformed vertices."""
    from alignsdf_amd.icp import align_to_ground_truth, load_obj
    from oracle import mc33
    n = 40
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    vol = (np.linalg.norm(g * np.array([1.0, 1.4, 2.0]), axis=-1) - 0.6).astype(np.float32)
    v, f = mc33.marching_cubes_raw(vol)
    gt_v = v * 0.01 + np.array([0.1, -0.05, 0.4])
    with open(tmp_path / "gt.obj", "w") as fh:
        for p in gt_v:
            fh.write("v %.9f %.9f %.9f\n" % tuple(p))
        for t in f:
            fh.write("f %d %d %d\n" % tuple(t + 1))
    lv, lf = load_obj(str(tmp_path / "gt.obj"))
    assert lv.shape == gt_v.shape and np.array_equal(lf, f)
    pred = (gt_v - np.array([0.01, 0.02, -0.015])) / 1.12           # predicted mesh = GT under a similarity
    aligned, trans, scale, info = align_to_ground_truth(pred, f, lv, lf, samples=20000)
    assert abs(scale - 1.12) < 5e-3 and np.abs(aligned - gt_v).max() < 1e-3
    assert info["iterations"] <= 100


def _sphere_volume(n=64, r=0.55):
    ax = torch.linspace(-1, 1, n)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    return (torch.sqrt(zz * zz * 1.0 + yy * yy * 1.6 + xx * xx * 2.3) - r).cuda()


def test_device_surface_sampler_is_the_host_sampler_bit_for_bit():
    """The eval-mode hook samples the predicted surface on the device (alignsdf_amd.icp.sample_surface_device): same faces picked,
    same points, bit for bit, as the host sampler on the same mesh - for a marching-cubes surface, for a mesh padded to a larger
    capacity with its face count on the device (the largest-component filter's output), and for other seeds / counts."""
    from alignsdf_amd.icp import sample_surface, sample_surface_device, sample_surface_native, sample_surface_torch
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.mesh_post import keep_largest_component_device
    v, f = marching_cubes_device(_sphere_volume(), 0.0)
    vs, org = np.float32(2.0 / 63), [-1.0, -1.0, -1.0]
    placed = v * float(vs) + torch.tensor(org, dtype=torch.float32, device="cuda")
    host_v, host_f = placed.cpu().numpy(), f.cpu().numpy()
    for count, seed in ((30000, 0), (30000, 1), (1000, 7)):
        want = sample_surface(host_v, host_f, count, seed)
        got = sample_surface_device(placed, f, count, seed)
        assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), want), (count, seed)
        # the three device forms: K9 on positions (what the line above ran: fp32 vertices, int32 faces), K9 placing lattice-unit
        # vertices itself (the exporter's fp32 multiply + add), and the elementwise torch form other input types take
        assert f.dtype == torch.int32 and placed.dtype == torch.float32
        assert np.array_equal(sample_surface_native(v, f, count, seed, placement=(vs, org)).cpu().numpy(), want), (count, seed)
        assert np.array_equal(sample_surface_torch(placed, f, count, seed).cpu().numpy(), want), (count, seed)
        assert np.array_equal(sample_surface_device(placed.double(), f.long(), count, seed).cpu().numpy(), want), (count, seed)
    # K8's output: capacity of the input, the kept counts on the device
    kv, kf, counts = keep_largest_component_device(v, f, vs, org)
    c = counts.cpu().numpy()
    kept_placed = kv * float(vs) + torch.tensor(org, dtype=torch.float32, device="cuda")
    want = sample_surface(kept_placed[:c[0]].cpu().numpy(), kf[:c[1]].cpu().numpy(), 30000, 0)
    got = sample_surface_device(kept_placed, kf, 30000, 0, counts[1:2])
    assert np.array_equal(got.cpu().numpy(), want)
    # a padded copy with garbage behind the live faces gives the same samples
    pad = torch.cat([kf[:c[1]], torch.randint(0, int(c[0]), (5000, 3), dtype=kf.dtype, device="cuda")], 0)
    got = sample_surface_device(kept_placed, pad, 30000, 0, counts[1:2])
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(sample_surface_torch(kept_placed, pad, 30000, 0, counts[1:2]).cpu().numpy(), want)
    assert np.array_equal(sample_surface_native(kv, pad, 30000, 0, counts[1:2], placement=(vs, org)).cpu().numpy(), want)
    # meshes around the scan's chunk size (4096 faces per round of the one-workgroup scan) and a single face
    rng = np.random.default_rng(5)
    for nf in (1, 2, 4095, 4096, 4097, 12289):
        hv = rng.standard_normal((max(3, nf // 2 + 3), 3)).astype(np.float32)
        hf = rng.integers(0, hv.shape[0], (nf, 3)).astype(np.int32)
        hf[0] = [0, 1, 2]
        want = sample_surface(hv, hf, 5000, 3)
        got = sample_surface_device(torch.from_numpy(hv).cuda(), torch.from_numpy(hf).cuda(), 5000, 3)
        assert np.array_equal(got.cpu().numpy(), want), nf


def test_device_normalisation_equals_the_host_normalisation():
    """asdf_icp_normalise (K9: mean / RMS radius of both sample sets and the moved source set, one launch) against normalise_source
    (icp_trans_scale.py:25-31 in numpy): statistics and points to 1e-12 relative (the sums are associated differently), and the same
    bits run to run."""
    from alignsdf_amd.icp import normalise_source, start_icp_device, finish_icp
    rng = np.random.default_rng(11)
    ps = rng.standard_normal((30000, 3)) * [0.3, 0.2, 0.5] + [0.1, -0.2, 0.05]
    pt = ps * 1.3 + [0.4, 0.1, -0.3] + rng.standard_normal((30000, 3)) * 1e-3
    want, (os_, ss, ot, st) = normalise_source(ps, pt)
    runs = []
    for _ in range(2):
        job = start_icp_device(torch.from_numpy(ps).cuda(), torch.from_numpy(pt).cuda(), max_iter=1)
        finish_icp(job, ps[:4])
        runs.append((job.src.cpu().numpy(), job.host[0].numpy().copy()))
    got, stats = runs[0]
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert np.allclose(stats, np.concatenate([os_, [ss], ot, [st]]), rtol=1e-12, atol=1e-15)
    assert np.abs(got - want).max() <= 1e-12


def test_device_alignment_equals_the_host_alignment(tmp_path):
    """start_alignment_device (placement, sampling, normalisation and ICP enqueued on the device, nothing waited for) against the
    host-prepared start_alignment on the same surface and ground truth: same samples, so the same transform to 1e-9."""
    from alignsdf_amd.icp import finish_icp, sample_surface, start_alignment, start_alignment_device
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.mesh_post import keep_largest_component_device
    v, f = marching_cubes_device(_sphere_volume(), 0.0)
    vs, org = np.float32(2.0 / 63), [-1.0, -1.0, -1.0]
    kv, kf, counts = keep_largest_component_device(v, f, vs, org)
    c = counts.cpu().numpy()
    placed = (kv[:c[0]] * float(vs) + torch.tensor(org, dtype=torch.float32, device="cuda")).cpu().numpy()
    faces = kf[:c[1]].cpu().numpy()
    gt_v = placed.astype(np.float64) * 1.07 + np.array([0.02, -0.03, 0.01])
    host = finish_icp(start_alignment(placed, faces, gt_v, faces), placed)
    target = sample_surface(gt_v, faces, 30000, 1)
    dev = finish_icp(start_alignment_device(kv, kf, counts, org, vs, target), placed)
    assert dev["iterations"] == host["iterations"]
    assert abs(dev["all_scale"] - host["all_scale"]) <= 1e-9 and np.abs(dev["all_trans"] - host["all_trans"]).max() <= 1e-9
    assert np.abs(dev["vertices"] - host["vertices"]).max() <= 1e-9
    assert abs(dev["all_scale"] - 1.07) < 5e-3


def test_obj_reader_forms(tmp_path):
    from alignsdf_amd.icp import load_obj
    p = tmp_path / "a.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1/1/1 3/3/1 4/4/1\n")
    v, f = load_obj(str(p))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3]]
    p.write_text("v 0 0 0 1 0 0\nv 1 0 0 0 1 0\nv 1 1 0 0 0 1\nv 0 1 0 1 1 1\nf 1 2 3 4\nf -4 -3 -2\n")      # colours, a quad, negative indices
    v, f = load_obj(str(p))
    assert v.tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]] and f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]


def _nn_both_ways(a, b, mode):
    """Nearest-neighbour structure through the C ABI under a search mode: the Chamfer sums and one ICP run."""
    import ctypes
    from alignsdf_amd import _native
    from alignsdf_amd.icp import run_icp_f
    L = _native.lib()
    _native.check(L.asdf_icp_set_search(mode), "asdf_icp_set_search")
    try:
        A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        nbytes = ctypes.c_size_t()
        _native.check(L.asdf_icp_workspace_bytes(len(a), len(b), ctypes.byref(nbytes)), "ws")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
        out = (ctypes.c_double * 2)()
        _native.check(L.asdf_chamfer(A.data_ptr(), len(a), B.data_ptr(), len(b), ws.data_ptr(), ws.numel(), out,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "asdf_chamfer")
        return (out[0], out[1]), run_icp_f(a, b)
    finally:
        L.asdf_icp_set_search(0)


@pytest.mark.parametrize("case", ["surfaces", "offset", "clustered", "ties", "small"])
def test_grid_search_equals_brute_force(case):
    """The uniform-grid nearest-neighbour search (round 3) against the brute-force kernel: identical Chamfer sums (bit for bit: the
    same neighbours, the same fixed-order reduction) and an identical ICP run, on overlapping surfaces, on sets that barely
    overlap (long shell walks, queries outside the grid), on clustered points, on lattice points with many exact distance ties,
    and on a set smaller than the grid resolution."""
    rng = np.random.default_rng(11)
    if case == "surfaces":
        u = rng.normal(size=(30000, 3)); a = 0.35 * u / np.linalg.norm(u, axis=1, keepdims=True) + np.array([-0.25, 0.0, 0.0])
        w = rng.normal(size=(30000, 3)); b = 0.37 * w / np.linalg.norm(w, axis=1, keepdims=True) + np.array([-0.23, 0.01, 0.0])
    elif case == "offset":
        a = rng.random((20000, 3)) * 0.3
        b = rng.random((15000, 3)) * 0.3 + np.array([0.5, -0.4, 0.2])
    elif case == "clustered":
        a = np.concatenate([rng.normal(size=(10000, 3)) * 0.01, rng.normal(size=(10000, 3)) * 0.2 + 0.5])
        b = np.concatenate([rng.normal(size=(12000, 3)) * 0.02 + 0.01, rng.random((3000, 3))])
    elif case == "ties":
        g = np.stack(np.meshgrid(*[np.arange(24) / 8.0] * 3, indexing="ij"), -1).reshape(-1, 3)
        a, b = g + 1.0 / 16.0, g.copy()                    # every query has 8 equidistant neighbours
    else:
        a, b = rng.random((1500, 3)), rng.random((1100, 3)) * np.array([1.0, 1e-3, 1.0])
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    brute, icp_brute = _nn_both_ways(a, b, 1)
    grid, icp_grid = _nn_both_ways(a, b, 2)
    assert brute == grid, (brute, grid)
    assert icp_brute[0] == icp_grid[0] and np.array_equal(icp_brute[1], icp_grid[1]) and icp_brute[2:] == icp_grid[2:]
