"""GPU parity of the part classifier / label pass (K1's label variants behind asdf_decode_points_cls) against the
reference's own outputs (ref_cls_combcls3.npz) and, for the SeparateDecoder configurations the reference cannot
build, against the CPU oracle."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 2e-5       # scores are O(3): 1e-5 relative, like the SDF bar


def _decisive(scores, margin=1e-4):
    top2 = np.sort(scores, axis=1)[:, -2:]
    return (top2[:, 1] - top2[:, 0]) > margin


def _module(tag):
    from alignsdf_amd.networks.model import build_decoder
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    mano = obj = None
    if specs["PointFeatSize"] == 9:
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
        obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    return specs, dec, lat, mano, obj


def test_scores_match_reference(golden_dir):
    from alignsdf_amd.utils.utils import decode_sdf_multi_output
    g = np.load(golden_dir + "/ref_cls_combcls3.npz")
    specs, dec, lat, mano, obj = _module("combcls3")
    h, o, scores = decode_sdf_multi_output(dec, lat, torch.from_numpy(g["rand_pts"]).cuda(), mano, None, specs)
    assert scores.shape == (4096, 6)
    assert np.abs(scores.cpu().numpy() - g["rand_scores"]).max() <= TOL
    assert np.abs(h[:, 0].cpu().numpy() - g["rand_hand"]).max() <= 1e-5
    assert np.abs(o[:, 0].cpu().numpy() - g["rand_obj"]).max() <= 1e-5


@pytest.mark.parametrize("tag", ["cls3", "bothcls9", "combcls3"])
@pytest.mark.parametrize("M", [1, 31, 4096, 70001])
def test_classify_points_matches_oracle(tag, M):
    from alignsdf_amd.utils.utils import hip_decoder_for, sample_embedding
    from oracle import sdf_oracle as orc
    specs, dec, lat, mano, obj = _module(tag)
    hip = hip_decoder_for(dec)
    hip.set_sample(lat, sample_embedding(specs, mano, obj, hip.combined))
    pts = syn.uniform((M, 3), 4321 + M, -1.0, 1.0).astype(np.float32)
    hand, ob, scores, labels = hip.classify_points(torch.from_numpy(pts).cuda())
    cpu = lambda d: None if d is None else {k: v.cpu() for k, v in d.items()}
    want, want_labels = orc.classify_points(syn.full_state_dict(tag), syn.latent_code(0), pts, specs, cpu(mano), cpu(obj))
    wh, wo = orc.decode_points(syn.full_state_dict(tag), syn.latent_code(0), pts, specs, cpu(mano), cpu(obj))
    assert np.abs(scores.cpu().numpy() - want.numpy()).max() <= TOL
    assert np.abs(hand.cpu().numpy() - wh.numpy()).max() <= 1e-5 and np.abs(ob.cpu().numpy() - wo.numpy()).max() <= 1e-5
    s = scores.cpu().numpy()
    assert np.array_equal(labels.cpu().numpy(), s.argmax(1))            # the kernel's argmax of its own scores
    ok = _decisive(want.numpy())
    assert np.array_equal(labels.cpu().numpy()[ok], want_labels.numpy()[ok])
    # scores without the SDF outputs are the same numbers
    _, _, s2, l2 = hip.classify_points(torch.from_numpy(pts).cuda(), want_sdf=False)
    assert torch.equal(s2, scores) and torch.equal(l2, labels)


def test_label_pass_files_match_reference(tmp_path, golden_dir):
    """create_mesh_combined_decoder(label_out=True, viz=True): the label files of the hand mesh vs the reference's
    label pass on the same surface (vertices differ by the <=1e-5 SDF difference through interpolation)."""
    from alignsdf_amd.utils.mesh import create_mesh_combined_decoder
    g = np.load(golden_dir + "/ref_cls_combcls3.npz")
    specs, dec, lat, mano, obj = _module("combcls3")
    prefix = str(tmp_path / "s0")
    stats = create_mesh_combined_decoder(True, True, True, dec, lat, mano, obj, None, specs, prefix, N=32, max_batch=2 ** 18, return_stats=True,
                                         label_out=True, viz=True)
    z = np.load(prefix + "_hand_label.npz")
    assert z["points"].shape == g["lab_points"].shape and stats["hand"][0] == len(g["lab_points"])
    assert np.abs(z["points"] - g["lab_points"]).max() <= 2e-5
    from oracle import sdf_oracle as orc
    ref_scores, _ = orc.classify_points(syn.full_state_dict("combcls3"), syn.latent_code(0), g["lab_points"], specs)
    ok = _decisive(ref_scores.numpy(), 1e-3)                               # vertex positions differ slightly as well
    assert ok.mean() > 0.95 and np.array_equal(z["labels"][ok], g["lab_labels"][ok])
    assert z["labels"].dtype == np.float32
    assert os.path.exists(prefix + "_hand_label.obj") and os.path.exists(prefix + "_hand_color.ply")
    assert len(open(prefix + "_hand_label.obj").read().splitlines()) == len(g["lab_points"])


@pytest.mark.parametrize("tag", ["combcls3", "bothcls9"])
def test_pipelined_label_pass_equals_single_sample_path(tag, tmp_path):
    """reconstruct(label_out=True) re-binds sample k for its label pass while sample k+1 is in flight; every file must
    equal what the one-sample-at-a-time entry point writes."""
    from alignsdf_amd.reconstruct import reconstruct, synthetic_code_source
    from alignsdf_amd.utils.mesh import create_mesh_combined_decoder
    specs, dec, _, _, _ = _module(tag)
    split = tmp_path / "split.json"
    names = ["scene/%04d.jpg" % k for k in range(4)]
    split.write_text(json.dumps({"filenames": names}))
    src = synthetic_code_source("both9" if tag == "bothcls9" else "nerf3")
    recs = reconstruct(dec, specs, str(split), str(tmp_path / "out"), 0, 4, cube_dim=32, label_out=True, viz=False, code_source=src)
    assert len(recs) == 4 and all("labels_hand" in r for r in recs)
    for k in range(4):
        lat, mano, obj = src("%04d" % k, k)
        prefix = str(tmp_path / ("one_%d" % k))
        create_mesh_combined_decoder(True, True, True, dec, lat, mano, obj, None, specs, prefix, N=32, label_out=True)
        a = np.load(prefix + "_hand_label.npz")
        b = np.load(str(tmp_path / "out" / "meshes" / ("%04d_hand_label.npz" % k)))
        assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["labels"], b["labels"])
        for part in ("hand", "obj"):
            assert open(prefix + "_%s.ply" % part, "rb").read() == open(str(tmp_path / "out" / "meshes" / ("%04d_%s.ply" % (k, part))), "rb").read()


def test_classifier_abi_errors(native_lib):
    """asdf_decode_points_cls without a classifier, with both outputs NULL, and bad class counts -> ASDF_EINVAL."""
    from alignsdf_amd.utils.utils import hip_decoder_for
    specs, dec, lat, _, _ = _module("nerf3")
    hip = hip_decoder_for(dec)
    hip.set_sample(lat)
    pts = torch.zeros((8, 3), device="cuda")
    lab = torch.zeros(8, dtype=torch.int32, device="cuda")
    L = native_lib
    assert L.asdf_decode_points_cls(hip._h, pts.data_ptr(), 8, None, None, None, lab.data_ptr(), None) == -1
    with pytest.raises(ValueError):
        hip.classify_points(pts)
    w = np.zeros((9, 512), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.asdf_decoder_set_classifier(hip._h, vp(w), vp(w), 9) == -1
    assert L.asdf_decoder_set_classifier(hip._h, vp(w), vp(w), 0) == -1
    assert L.asdf_decoder_set_classifier(hip._h, None, vp(w), 6) == -1
    assert L.asdf_decoder_set_classifier(hip._h, vp(w), vp(w), 6) == 0
    assert L.asdf_decode_points_cls(hip._h, pts.data_ptr(), 8, None, None, None, None, None) == -1
    assert L.asdf_decode_points_cls(hip._h, pts.data_ptr(), 8, None, None, None, lab.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(lab.abs().sum()) == 0           # all-zero classifier: first maximum is class 0
