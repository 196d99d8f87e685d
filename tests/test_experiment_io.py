"""On-disk contract of the reconstruction driver: <exp>/specs.json, <exp>/ModelParameters/latest.pth
with `module.decoder.*` keys (networks/model_utils.py:40-47), split json, Eval_<task>/meshes/<id>_{hand,obj}.ply."""
import json
import os

import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn


def make_experiment(root, tag, names):
    os.makedirs(os.path.join(root, "ModelParameters"), exist_ok=True)
    specs = syn.specs_for(tag)
    json.dump(specs, open(os.path.join(root, "specs.json"), "w"))
    sd = {"module.decoder." + k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()}
    sd["module.encoder.conv1.weight"] = torch.zeros(4, 3, 7, 7)          # encoder tensors are ignored
    torch.save({"epoch": 1600, "model_state_dict": sd}, os.path.join(root, "ModelParameters", "latest.pth"))
    split = os.path.join(root, "split.json")
    json.dump({"filenames": ["data/obman/test/rgb/%s.jpg" % n for n in names]}, open(split, "w"))
    return specs, split


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9"])
def test_load_experiment_reads_decoder_tensors(tag, tmp_path):
    from alignsdf_amd.reconstruct import load_experiment
    specs, _ = make_experiment(str(tmp_path), tag, ["a"])
    specs2, dec = load_experiment(str(tmp_path))
    assert specs2 == specs
    ref = syn.full_state_dict(tag)
    got = dec.state_dict()
    assert set(got) == set(ref)
    for k in ref:
        assert np.array_equal(got[k].numpy(), ref[k]), k
    # the module's own forward (host check only) agrees with the oracle
    from oracle import sdf_oracle as orc
    pts = torch.from_numpy(syn.uniform((64, 3), 9, -1, 1).astype(np.float32))
    lat = torch.from_numpy(syn.latent_code(0))
    if tag != "both9":
        feats = orc.point_features(pts, specs, None, None)
        with torch.no_grad():
            h, o, _ = dec(torch.cat([lat.expand(64, -1), feats], 1))
        rh, ro = orc.decode_points(ref, lat, pts, specs)
        assert (h.squeeze(1) - rh).abs().max() <= 1e-6 and (o.squeeze(1) - ro).abs().max() <= 1e-6


@pytest.mark.gpu
def test_reconstruct_cli_writes_reference_layout(tmp_path):
    from alignsdf_amd import reconstruct as rc
    from alignsdf_amd.ply import read_ply
    names = ["00000012", "00000047", "00000100"]
    specs, split = make_experiment(str(tmp_path), "nerf3", names)
    with pytest.raises(SystemExit):       # no silent default: without --codes / --synthetic there is nothing to reconstruct
        rc.main(["-e", str(tmp_path), "-s", split, "-t", "obman", "--start_point", "1", "--end_point", "3", "--cube_dim", "32"])
    import json
    # round 6: WITHOUT --fast every voxel of both passes is evaluated at <= 1e-5 (ordinary sweeps), like utils/mesh.py:27-115
    recs0 = rc.main(["-e", str(tmp_path), "-s", split, "-t", "obman", "--start_point", "1", "--end_point", "3", "--cube_dim", "32",
                     "--synthetic"])
    sw0 = json.load(open(os.path.join(str(tmp_path), "Eval_obman", "sweeps_1_3.json")))["sweeps"]
    assert sw0["sweeps_audited"] == 0 and sw0["coarse_pass"]["mode_now"] == "exact" and sw0["fine_pass"]["mode_now"] == "exact"
    assert sw0["coarse_pass"]["ordinary_sweeps"] == 2 and sw0["fine_pass"]["ordinary_sweeps"] == 2 and sw0["whole_lattice_comparisons"] == {"coarse_lattice": 0, "zoom_lattice": 0}
    recs = rc.main(["-e", str(tmp_path), "-s", split, "-t", "obman", "--start_point", "1", "--end_point", "3", "--cube_dim", "32",
                    "--synthetic", "--fast"])
    assert [(r["V_hand"], r["F_hand"], r["V_obj"], r["F_obj"]) for r in recs] == [(r["V_hand"], r["F_hand"], r["V_obj"], r["F_obj"]) for r in recs0]
    assert [r["name"] for r in recs] == names[1:] and [r["index"] for r in recs] == [1, 2]
    mesh_dir = os.path.join(str(tmp_path), "Eval_obman", "meshes")
    assert sorted(os.listdir(mesh_dir)) == sorted(["%s_%s.ply" % (n, p) for n in names[1:] for p in ("hand", "obj")])
    for r in recs:
        v, f = read_ply(os.path.join(mesh_dir, r["name"] + "_hand.ply"))
        assert len(f) > 100 and f.max() < len(v) and r["F_hand"] >= len(f)
    # round 5: next to meshes/ the run says which sweeps produced the volumes behind its files
    rep = json.load(open(os.path.join(str(tmp_path), "Eval_obman", "sweeps_1_3.json")))
    assert rep["range"] == [1, 3] and rep["samples"] == 2 and rep["cube_dim"] == 32
    # f1 (parity unpinned against trimesh): what the largest-component filter dropped as open - the only place fill_holes could differ
    assert rep["dropped_open_components"] == 0 and rep["surfaces_filtered"] == 4 and rep["dropped_small_components"] >= 0
    sw = rep["sweeps"]
    assert sw["evaluator"].startswith("hip kernels") and sw["arithmetic"] == {"at_start": "f16x3", "now": "f16x3", "fell_back_to_fp32_chain": False}
    assert sw["sweeps_refused"] == 0 and sw["sweeps_repeated"] == 0 and sw["modes_switched_off"] == []
    # sample 1: both lattices compared as a whole (ordinary sweeps); sample 2: the audited one-plane sweeps
    assert sw["whole_lattice_comparisons"] == {"coarse_lattice": 1, "zoom_lattice": 1}
    assert sw["coarse_pass"]["one_plane_box_sweeps_accepted"] == 1 and sw["fine_pass"]["one_plane_band_sweeps_accepted"] == 1
    assert sw["sweeps_audited"] == 2 and sw["coarse_pass"]["mode_now"] == "box" and sw["fine_pass"]["mode_now"] == "band"


@pytest.mark.gpu
def test_dist_reconstruct_cli_two_ranks(tmp_path):
    """`torchrun -m alignsdf_amd.dist_reconstruct` with 2 ranks (sharing the GPU over gloo on a 1-GPU box): contiguous
    shards, every sample exactly once, records gathered on rank 0, files in the reference layout."""
    import subprocess
    import sys
    names = ["%08d" % i for i in (3, 14, 15, 92, 65)]
    specs, split = make_experiment(str(tmp_path), "nerf3", names)
    env = dict(os.environ, ASDF_DIST_BACKEND="gloo", ASDF_SHARE_DEVICE="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", "-m", "alignsdf_amd.dist_reconstruct", "-e", str(tmp_path), "-t", "obman", "--split", split,
           "--cube_dim", "32", "--synthetic", "--allow_missing_gt", "--data_root", str(tmp_path / "no_such_data"), "--fast"]
    # (a stale shard report of an earlier run with another world size must not enter this run's totals: ADVICE r05)
    os.makedirs(os.path.join(str(tmp_path), "Eval_obman"), exist_ok=True)
    json.dump({"range": [0, 5], "samples": 5, "cube_dim": 32, "sweeps": {"sweeps_audited": 1000, "sweeps_refused": 7, "sweeps_repeated": 7}},
              open(os.path.join(str(tmp_path), "Eval_obman", "sweeps_0_5.json"), "w"))
    subprocess.run(cmd, check=True, timeout=600, env=env)
    whole = json.load(open(os.path.join(str(tmp_path), "Eval_obman", "reconstruct_summary.json")))
    assert whole["complete"] is True and whole["samples"] == 5
    assert [(s["rank"], s["range"], s["status"], s["samples"]) for s in whole["shards"]] == [(0, [0, 2], "ok", 2), (1, [2, 5], "ok", 3)]
    summary = whole["records"]
    assert [r["index"] for r in summary] == [0, 1, 2, 3, 4]
    # every rank left its records next to the meshes BEFORE the gather
    for a, b, n in ((0, 2, 2), (2, 5, 3)):
        shard = json.load(open(os.path.join(str(tmp_path), "Eval_obman", "records_%d_%d.json" % (a, b))))
        assert shard["status"] == "ok" and shard["range"] == [a, b] and [r["index"] for r in shard["records"]] == list(range(a, b)) and len(shard["records"]) == n
    assert [r["rank"] for r in summary] == [0, 0, 1, 1, 1]                 # len // W per rank, remainder to the last
    meshes = sorted(os.listdir(os.path.join(str(tmp_path), "Eval_obman", "meshes")))
    assert meshes == sorted(["%s_%s.ply" % (n, p) for n in names for p in ("hand", "obj")])
    assert all(r["F_hand"] > 100 and r["F_obj"] > 100 for r in summary)
    assert all(r["icp_skipped"] == 1 for r in summary)          # eval mode without ground-truth meshes is visible in the records
    # each shard's sweeps_<start>_<end>.json and rank 0's merged sweeps.json, next to meshes/
    out = os.path.join(str(tmp_path), "Eval_obman")
    assert os.path.exists(os.path.join(out, "sweeps_0_2.json")) and os.path.exists(os.path.join(out, "sweeps_2_5.json"))
    merged = json.load(open(os.path.join(out, "sweeps.json")))
    assert [s["range"] for s in merged["shards"]] == [[0, 2], [2, 5]] and merged["totals"]["samples"] == 5
    assert merged["totals"]["sweeps_refused"] == 0 and merged["totals"]["modes_switched_off"] == [] and not merged["totals"]["fell_back_to_fp32_chain"]
    assert merged["totals"]["sweeps_audited"] == 2 * (1 + 2)            # every sample but each shard's first runs two audited sweeps


@pytest.mark.gpu
def test_sweeps_json_counts_a_provoked_refusal(tmp_path, monkeypatch):
    """a sweep that was refused and repeated is visible AFTER the run, next to the meshes - not only in a log
    line.  Six samples at N = 64; while samples 1 / 2 are in flight the capacity of the narrow-band list is set to 10 voxels, so
    the band sweep that is judged then is refused ("voxels listed, capacity") and repeated as an ordinary sweep; the run carries on
    with the audited sweeps.  The meshes are those of a run that never heard of the one-plane sweeps."""
    from alignsdf_amd import hip_decoder as hd
    from alignsdf_amd import reconstruct as rc
    from alignsdf_amd.ply import read_ply
    from alignsdf_amd.utils.utils import decoder_for
    names = ["%08d" % i for i in range(6)]
    specs, split = make_experiment(str(tmp_path), "nerf3", names)
    specs, decoder = rc.load_experiment(str(tmp_path))
    hip = decoder_for(decoder, specs)
    base = rc.synthetic_code_source("nerf3")
    real_cap = hd.BAND_CAP

    def sabotaging_source(name, index):          # (the pipeline fetches sample k + 2 while sample k is being finished)
        monkeypatch.setattr(hd, "BAND_CAP", 10 if index in (3, 4) else real_cap)
        return base(name, index)

    out = str(tmp_path / "Eval_obman")
    recs = rc.reconstruct(decoder, specs, split, out, 0, 6, cube_dim=64, code_source=sabotaging_source, fast=True)
    monkeypatch.setattr(hd, "BAND_CAP", real_cap)
    rep = json.load(open(os.path.join(out, "sweeps_0_6.json")))["sweeps"]
    assert 1 <= rep["sweeps_refused"] <= 2 and rep["sweeps_repeated"] >= rep["sweeps_refused"], rep
    assert rep["fine_pass"]["refused_and_repeated"] == rep["sweeps_refused"] and rep["coarse_pass"]["refused_and_repeated"] == 0
    assert rep["refusals_for_error"] == 0 and rep["whole_lattice_comparisons"] == {"coarse_lattice": 1, "zoom_lattice": 1}, rep
    assert rep["modes_switched_off"] == [] and rep["coarse_pass"]["mode_now"] == "box" and rep["fine_pass"]["mode_now"] == "band"
    assert rep["fine_pass"]["one_plane_band_sweeps_accepted"] >= 2 and rep["arithmetic"]["fell_back_to_fp32_chain"] is False
    # the files are the ordinary sweeps' files
    out2 = str(tmp_path / "Eval_plain")
    recs2 = rc.reconstruct(decoder, specs, split, out2, 0, 6, cube_dim=64, code_source=base, fast=False)
    rep2 = json.load(open(os.path.join(out2, "sweeps_0_6.json")))["sweeps"]
    assert rep2["sweeps_audited"] == 0 and rep2["coarse_pass"]["ordinary_sweeps"] == 6 and rep2["fine_pass"]["ordinary_sweeps"] == 6
    assert rep2["coarse_pass"]["mode_now"] == "exact"
    for a, b in zip(recs, recs2):
        assert (a["V_hand"], a["F_hand"], a["V_obj"], a["F_obj"]) == (b["V_hand"], b["F_hand"], b["V_obj"], b["F_obj"])
        for part in ("hand", "obj"):
            va, fa = read_ply(os.path.join(out, "meshes", "%s_%s.ply" % (a["name"], part)))
            vb, fb = read_ply(os.path.join(out2, "meshes", "%s_%s.ply" % (a["name"], part)))
            assert np.array_equal(va, vb) and np.array_equal(fa, fb)
    assert (hip.coarse_mode, hip.fine_mode) == ("exact", "exact")


def test_reconstruct_requires_a_code_source(tmp_path):
    """reconstruct() used to fall back to synthetic latents and write <real sample>_hand.ply files from them."""
    from alignsdf_amd import reconstruct as rc
    specs, split = make_experiment(str(tmp_path), "nerf3", ["a", "b"])
    with pytest.raises(ValueError, match="code_source"):
        rc.reconstruct(object(), specs, split, str(tmp_path / "out"), 0, 2)


def test_evaluation_summary_marks_uncomputed_metrics(tmp_path):
    """The joint / vertex / object-pose errors come from the encoder (outside this build): nan, not 0.0."""
    from alignsdf_amd.evaluate import write_summary
    os.makedirs(tmp_path / "Eval_obman")
    path = write_summary(str(tmp_path), "obman", [("x", 1.5, float("nan"), float("nan"))], 2)
    text = open(path).read()
    assert "mean joints error:nan" in text and "mean verts error:nan" in text and "failure count:1" in text


@pytest.mark.gpu
def test_rccl_backend_executes_the_collectives_of_the_path():
    """backend "nccl" (RCCL) with world size 1 on the box's GPU: the rank count, the MAX of the elapsed time, the barrier and
    gather_records - so that the RCCL code path of dist_reconstruct / bench.py has run on an MI355X, not only over gloo."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_single_rank_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl_ok"] and line["backend"] == "nccl" and line["n_ranks"] == 1 and line["max_elapsed"] == 1.25
    assert [m["index"] for m in line["merged"]] == [1, 3] and line["merged"][1]["icp_skipped"] == 1 and line["empty"] == []


@pytest.mark.gpu
def test_code_upload_does_not_wait_for_the_queued_passes(tmp_path):
    """The per-sample codes reach the device through CodeUploader (pinned ring, side stream, device-side wait): the values arrive,
    the ring survives more uploads than it has slots, the compute stream sees the data (a consumer kernel queued right behind), and
    the CALL returns while a long kernel queue is still running - a plain `.to(device)` waited for all of it (50 ms per sample
    with a sample's two passes queued: the eval-mode flow then ran with 4.4 ms of idle GPU per sample)."""
    import time
    from alignsdf_amd.reconstruct import CodeUploader, npz_code_source
    up = CodeUploader("cuda")
    rng = np.random.default_rng(0)
    arrays = [rng.standard_normal((1, 256)).astype(np.float32) for _ in range(3 * CodeUploader.SLOTS)]
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        a = (a @ a) * 1e-3
    queued = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = [up(x) * 2.0 for x in arrays]                       # (the multiply is the consumer on the compute stream)
    call = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    for x, y in zip(arrays, out):
        assert np.array_equal(y.cpu().numpy(), x * 2.0)
    assert total > 4 * call, "the uploads waited for the queue: %.1f ms of calls, queue drained after %.1f ms (enqueue %.1f ms)" % (1e3 * call, 1e3 * total, 1e3 * queued)
    # the product's code source goes through it
    np.savez(tmp_path / "s.npz", latent=arrays[0], global_trans=np.tile(np.eye(4, dtype=np.float32), (1, 16, 1, 1)),
             rot_center=np.zeros((1, 1, 3), np.float32), obj_trans=np.eye(4, dtype=np.float32)[None])
    # (round 6: by default the code sources leave host-side codes on the HOST - the HIP decoder reads them from pinned memory;
    # on_host=False asks for device tensors through the uploader)
    lat, mano, obj = npz_code_source(str(tmp_path))("s", 0)
    assert lat.device.type == "cpu" and lat.dtype == torch.float32 and np.array_equal(lat.numpy(), arrays[0]) and mano["global_trans"].device.type == "cpu"
    lat, mano, obj = npz_code_source(str(tmp_path), on_host=False)("s", 0)
    torch.cuda.synchronize()
    assert lat.is_cuda and np.array_equal(lat.cpu().numpy(), arrays[0]) and mano["global_trans"].shape == (1, 16, 4, 4) and obj["obj_trans"].shape == (1, 4, 4)


def test_code_sources_on_a_cpu_device_need_no_stream(tmp_path):
    """CodeUploader on a non-CUDA device is a plain conversion (the CPU tests and tools construct code sources with device="cpu")."""
    from alignsdf_amd.reconstruct import CodeUploader, npz_code_source
    up = CodeUploader("cpu")
    x = np.arange(12, dtype=np.float32).reshape(3, 4)[:, ::2]          # (not contiguous)
    y = up(x)
    assert y.device.type == "cpu" and y.dtype == torch.float32 and np.array_equal(y.numpy(), x) and up.stream is None
    np.savez(tmp_path / "a.npz", latent=np.ones((1, 256), np.float64), obj_trans=np.eye(4)[None])
    lat, mano, obj = npz_code_source(str(tmp_path), device="cpu")("a", 0)
    assert lat.dtype == torch.float32 and mano is None and obj["obj_trans"].dtype == torch.float32 and obj["obj_trans"].shape == (1, 4, 4)


@pytest.mark.gpu
def test_reconstruct_strided_range(tmp_path):
    """reconstruct(..., start, end, stride): every stride-th sample of the range, indices and names of the split, one sweep report named
    after the range (dist_reconstruct --shard strided deals rank r the slice r, r + W, ...)."""
    from alignsdf_amd import reconstruct as rc
    names = ["%08d" % i for i in (7, 11, 13, 17, 19, 23)]
    specs, split = make_experiment(str(tmp_path), "nerf3", names)
    specs, decoder = rc.load_experiment(str(tmp_path))
    out = str(tmp_path / "Eval_obman")
    recs = rc.reconstruct(decoder, specs, split, out, 1, 6, cube_dim=32, code_source=rc.synthetic_code_source("nerf3"), stride=2)
    assert [r["index"] for r in recs] == [1, 3, 5] and [r["name"] for r in recs] == [names[1], names[3], names[5]]
    assert sorted(os.listdir(os.path.join(out, "meshes"))) == sorted("%s_%s.ply" % (names[i], p) for i in (1, 3, 5) for p in ("hand", "obj"))
    rep = json.load(open(os.path.join(out, "sweeps_1_6.json")))
    assert rep["range"] == [1, 6] and rep["stride"] == 2 and rep["samples"] == 3
    # the same samples one by one through contiguous ranges: the same meshes (a sample does not depend on its neighbours in the shard)
    out2 = str(tmp_path / "Eval_single")
    for i in (1, 3, 5):
        rc.reconstruct(decoder, specs, split, out2, i, i + 1, cube_dim=32, code_source=rc.synthetic_code_source("nerf3"))
    from alignsdf_amd.ply import read_ply
    for i in (1, 3, 5):
        for part in ("hand", "obj"):
            a = read_ply(os.path.join(out, "meshes", "%s_%s.ply" % (names[i], part)))
            b = read_ply(os.path.join(out2, "meshes", "%s_%s.ply" % (names[i], part)))
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
