"""End-to-end GPU parity of the drop-in entry points (utils.mesh.create_mesh_combined_decoder,
deep_sdf.mesh.create_mesh) against the reference goldens and the CPU oracles."""
import os

import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _module(tag):
    from alignsdf_amd.networks.model import build_decoder
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    mano = obj = None
    if tag == "both9":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
        obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    return specs, dec, lat, mano, obj


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9"])
@pytest.mark.parametrize("N", [32, 64])
def test_two_pass_matches_reference(tag, N, golden_dir):
    """Pass 1 -> bbox -> zoom cube -> pass 2, all computed by the product, vs the reference's own run."""
    from alignsdf_amd.utils.mesh import decode_two_pass
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    specs, dec, lat, mano, obj = _module(tag)
    r = decode_two_pass(True, True, dec, lat, mano, obj, specs, N)
    assert np.array_equal(np.stack([r["bbox"][0:6], r["bbox"][8:14]]), g["bbox_%d" % N])
    assert np.array_equal(r["voxel_size"].numpy().reshape(1), g["new_voxel_size_%d" % N])      # bit-equal zoom cube
    assert np.array_equal(np.array(r["origin"]), g["mc_origin_%d" % N])
    sel = g["probe_sel_%d" % N]
    assert np.abs(r["vol_hand"].cpu().numpy().reshape(-1)[sel] - g["p2_hand_%d" % N]).max() <= TOL
    assert np.abs(r["vol_obj"].cpu().numpy().reshape(-1)[sel] - g["p2_obj_%d" % N]).max() <= TOL


@pytest.mark.parametrize("tag", ["nerf3", "both9"])
def test_create_mesh_combined_decoder_files(tag, tmp_path, golden_dir):
    """Files written by the drop-in entry point: triangle / vertex counts identical to skimage on the
    reference's volumes; vertices within 1e-5 (they inherit the <=1e-5 SDF difference through interpolation)."""
    from alignsdf_amd.ply import read_ply
    from alignsdf_amd.utils.mesh import create_mesh_combined_decoder
    gm = np.load(golden_dir + "/mc_volumes.npz")
    gd = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    specs, dec, lat, mano, obj = _module(tag)
    prefix = str(tmp_path / "sample0")
    stats = create_mesh_combined_decoder(True, True, False, dec, lat, mano, obj, None, specs, prefix, N=32, max_batch=2 ** 18, return_stats=True)
    for part in ("hand", "obj"):
        v, f = read_ply("%s_%s.ply" % (prefix, part))
        ref_v, ref_f = gm["dec_%s_%s32.verts" % (tag, part)], gm["dec_%s_%s32.faces" % (tag, part)]
        assert stats[part] == (len(ref_v), len(ref_f))
        assert f.shape == ref_f.shape and np.array_equal(f, ref_f)
        world = ref_v + gd["mc_origin_32"].astype(np.float32)
        # an SDF difference d moves a vertex by d / (SDF change per voxel ~ 0.5 * voxel) along its edge
        assert np.abs(v - world).max() <= 5e-5


def test_mc_failure_is_skipped_like_reference(tmp_path, capsys):
    from alignsdf_amd.utils.mesh import convert_sdf_samples_to_ply
    out = convert_sdf_samples_to_ply(torch.ones(8, 8, 8).cuda(), [-1, -1, -1], 0.1, str(tmp_path / "none.ply"))
    assert out[0] is None and out[1] is None and list(out[2]) == [0, 0, 0] and list(out[3]) == [1]
    assert not os.path.exists(tmp_path / "none.ply")
    assert "Surface level must be within volume data range" in capsys.readouterr().out


def test_get_higher_res_cube_and_empty_branch():
    from alignsdf_amd.utils.mesh import get_higher_res_cube
    from oracle import sdf_oracle as orc
    N = 24
    vs = 2.0 / (N - 1)
    a = torch.from_numpy(syn.uniform((N, N, N), 3, -0.02, 1.0).astype(np.float32))
    b = torch.ones(N, N, N)
    for hb, ob in ((True, True), (True, False), (False, True)):
        nvs, norg = get_higher_res_cube(hb, ob, a.cuda(), b.cuda(), N, [-1, -1, -1], vs)
        rvs, rorg, _ = orc.get_higher_res_cube(hb, ob, a, b, N, vs)
        assert torch.equal(nvs, rvs) and torch.equal(norg, rorg)


def test_legacy_create_mesh(tmp_path, golden_dir):
    from alignsdf_amd.deep_sdf.mesh import create_mesh
    from alignsdf_amd.ply import read_ply
    from oracle import mc33
    g = np.load(golden_dir + "/ref_legacy.npz")
    specs, dec, lat, _, _ = _module("nerf3")
    create_mesh(dec, lat, str(tmp_path / "legacy"), N=32, max_batch=32 ** 3)
    v, f = read_ply(str(tmp_path / "legacy.ply"))
    rv, rf = mc33.marching_cubes_lewiner(g["vol_32"], 0.0, [2.0 / 31] * 3)
    assert f.shape == rf.shape and np.array_equal(f, rf)
    assert np.abs(v - (rv - 1.0)).max() <= 1e-4


def test_decode_sdf_multi_output_dropin(golden_dir):
    from alignsdf_amd.utils.utils import decode_sdf_multi_output
    for tag in ("nerf3", "both9", "comb3", "nerf9"):
        g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
        specs, dec, lat, mano, obj = _module(tag)
        h, o, _ = decode_sdf_multi_output(dec, lat, torch.from_numpy(g["rand_pts"]).cuda(), mano, None, specs, obj_results=obj)
        assert h.shape == (4096, 1)
        assert np.abs(h.squeeze(1).cpu().numpy() - g["rand_hand"]).max() <= TOL
        assert np.abs(o.squeeze(1).cpu().numpy() - g["rand_obj"]).max() <= TOL


@pytest.mark.parametrize("tag", ["nerf3", "both9"])
def test_pipelined_samples_equal_one_at_a_time(tag):
    """The software pipeline (pass 1 of sample k+1 queued before the MC of sample k, async embedding upload) must give
    bit-identical volumes and meshes to processing every sample on its own."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, reconstruct_sample, synthetic_code_source
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    src = synthetic_code_source(tag, "cuda")
    N = 48
    samples = [(i,) + src("s%d" % i, i) for i in (0, 5, 9, 12, 17)]
    got = {k: r for k, r in pipelined_two_pass(dec, specs, iter(samples), N)}
    assert list(got) == [0, 5, 9, 12, 17]
    for i, lat, mano, obj in samples:
        ref = reconstruct_sample(dec, specs, lat, mano, obj, N)
        r = got[i]
        assert r["origin"] == ref["origin"] and float(r["voxel_size"]) == ref["voxel_size"]
        for part in ("hand", "obj"):
            assert (r["V_" + part], r["F_" + part]) == (ref["V_" + part], ref["F_" + part])
            assert torch.equal(r["verts_" + part], ref["verts_" + part]) and torch.equal(r["faces_" + part], ref["faces_" + part])


@pytest.mark.parametrize("tag,branches", [("nerf3", (True, True)), ("nerf3", (True, False)), ("both9", (False, True)), ("comb3", (True, True))])
def test_ordinary_sweeps_enqueued_in_one_go_equal_the_step_by_step_run(tag, branches, monkeypatch):
    """Round 6: the product's DEFAULT (ordinary sweeps, every voxel) takes the one-go form too - coarse sweep, zoom cube on the device
    (asdf_zoom_cube), fine sweep reading its lattice from those words (asdf_decode_grid_dev), capacity-bounded marching cubes - from the
    second sample of a decoder on (the first calibrates the activation scales).  Zoom cubes, VOLUMES and meshes equal the step-by-step
    run's (ASDF_SPECULATE=0) bit for bit, for both branches, one branch alone, and a CombinedDecoder."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    for k in ("ASDF_FAST", "ASDF_COARSE", "ASDF_FINE", "ASDF_MATH"):
        monkeypatch.delenv(k, raising=False)
    hb, ob = branches
    specs = dict(syn.specs_for(tag), HandBranch=hb, ObjectBranch=ob)
    sd = {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()}
    N = 64
    src = synthetic_code_source(tag if tag != "comb3" else "nerf3", "cuda")

    def run():
        dec = build_decoder(specs, sd)
        items = [(i,) + src("s%d" % i, i) for i in range(7)]
        out = {}
        for k, r in pipelined_two_pass(dec, specs, iter(items), N):
            out[k] = {key: (r[key].clone() if torch.is_tensor(r[key]) else r[key]) for key in r
                      if key.startswith(("verts_", "faces_", "vol_")) or key in ("origin", "voxel_size")}
        return out, decoder_for(dec, specs, items[0][2])

    monkeypatch.setenv("ASDF_SPECULATE", "0")
    want, hip0 = run()
    assert hip0.events["samples_in_one_go"] == 0 and (hip0.coarse_mode, hip0.fine_mode) == ("exact", "exact")
    monkeypatch.setenv("ASDF_SPECULATE", "1")
    got, hip = run()
    assert hip.events["samples_in_one_go"] == 6 and hip.box_stats["exact"] == 7 and hip.band_stats["exact"] == 7, (hip.events, hip.box_stats)
    assert hip.box_stats["box"] == 0 and hip.band_stats["band"] == 0 and hip.events["repeated_sweeps"] == 0
    for k in want:
        a, b = want[k], got[k]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"]), k
        for part, on in (("hand", hb), ("obj", ob)):
            if on:
                assert torch.equal(a["vol_" + part], b["vol_" + part]), (k, part)
                assert torch.equal(a["verts_" + part], b["verts_" + part]) and torch.equal(a["faces_" + part], b["faces_" + part]), (k, part)


def test_eval_mode_icp_in_file_flow(tmp_path):
    """convert_sdf_samples_to_ply(eval_mode=True): the written hand mesh is aligned to data/<task>/test/mesh_hand/<id>.obj
    and (trans, scale) are returned like utils/mesh.py:385-395."""
    from alignsdf_amd.ply import read_ply
    from alignsdf_amd.utils.mesh import convert_sdf_samples_to_ply, ground_truth_mesh_path
    n = 48
    ax = torch.linspace(-1, 1, n)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = torch.sqrt(zz * zz * 1.0 + yy * yy * 2.0 + xx * xx * 3.5) - 0.55
    vs = 2.0 / (n - 1)
    out = str(tmp_path / "Eval_obman" / "meshes" / "00001234_hand.ply")
    # unaligned run gives the "prediction"; the ground truth is that mesh under a known similarity
    v0, f0, t0, s0 = convert_sdf_samples_to_ply(vol.cuda(), [-1, -1, -1], vs, out)
    pv, pf = read_ply(out)
    assert list(t0) == [0, 0, 0] and list(s0) == [1]
    gt = pv.astype(np.float64) * 1.09 + np.array([0.02, -0.01, 0.03])
    gt_path = ground_truth_mesh_path(out, "obman", str(tmp_path / "data"))
    assert gt_path.endswith("data/obman/test/mesh_hand/00001234.obj")
    os.makedirs(os.path.dirname(gt_path))
    with open(gt_path, "w") as fh:
        for p in gt:
            fh.write("v %.9f %.9f %.9f\n" % tuple(p))
        for t in pf:
            fh.write("f %d %d %d\n" % tuple(t + 1))
    v1, f1, trans, scale = convert_sdf_samples_to_ply(vol.cuda(), [-1, -1, -1], vs, out, eval_mode=True, task="obman",
                                                      data_root=str(tmp_path / "data"))
    av, af = read_ply(out)
    assert np.array_equal(af, pf) and np.array_equal(v1, v0)
    assert abs(float(scale[0]) - 1.09) < 5e-3 and np.abs(av - gt).max() < 2e-3
    assert np.abs(np.asarray(trans).reshape(3) - np.array([0.02, -0.01, 0.03])).max() < 5e-3


def test_packed_weights_follow_parameter_updates():
    """The per-module cache of packed weights is refreshed after an in-place parameter update."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.utils import decode_sdf_multi_output, hip_decoder_for
    specs = syn.specs_for("nerf3")
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    pts = torch.from_numpy(syn.uniform((256, 3), 3, -1, 1).astype(np.float32)).cuda()
    h0, _, _ = decode_sdf_multi_output(dec, lat, pts, None, None, specs)
    first = hip_decoder_for(dec)
    assert hip_decoder_for(dec) is first                       # unchanged parameters: cache hit
    with torch.no_grad():
        dec.linh4.bias.add_(0.25)
    h1, _, _ = decode_sdf_multi_output(dec, lat, pts, None, None, specs)
    assert hip_decoder_for(dec) is not first
    expect = torch.tanh(torch.atanh(h0.double()) + 0.25).float()
    assert (h1 - expect).abs().max().item() <= 2e-6
