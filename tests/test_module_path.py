"""Decoder variants outside the HIP kernels ("otherwise falls back to calling the module"): use_tanh, the
LayerNorm form, xyz_in_all, a pose-aligned model without mano_results, PixelAlign.  The reference's own outputs for each
are tests/golden/ref_variant_<name>.npz (make_r2_goldens.py).  CPU part: our module containers reproduce the reference's
modules; GPU part: the drop-in functions route these decoders through the module on PyTorch-ROCm and match the goldens."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn


def _module(name):
    from alignsdf_amd.networks import model as arch
    specs, cls, sd, mano, obj, cam, latent = syn.variant_config(name)
    dec = getattr(arch, cls)(256, specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"]).eval()
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    return specs, dec, t(mano), t(obj), None if cam is None else torch.from_numpy(cam), torch.from_numpy(latent)


@pytest.mark.parametrize("name", ["tanh", "layernorm", "xyzall"])
def test_module_containers_reproduce_the_reference_modules_cpu(name, golden_dir):
    g = np.load("%s/ref_variant_%s.npz" % (golden_dir, name))
    specs, dec, _, _, _, latent = _module(name)
    pts = torch.from_numpy(g["rand_pts"])
    with torch.no_grad():
        h, o, _ = dec(torch.cat([latent.expand(pts.shape[0], -1), pts], 1))
    assert np.abs(h[:, 0].numpy() - g["rand_hand"]).max() <= 1e-6 and np.abs(o[:, 0].numpy() - g["rand_obj"]).max() <= 1e-6


def test_which_decoders_take_the_module_path():
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.torch_decoder import needs_module_path
    for name, word in (("tanh", "use_tanh"), ("layernorm", "LayerNorm"), ("xyzall", "xyz_in_all"), ("pixelalign", "PixelAlign")):
        specs, dec, mano, *_ = _module(name)
        assert word in needs_module_path(dec, specs, mano)
    specs, dec, mano, *_ = _module("nomano")
    assert "without mano_results" in needs_module_path(dec, specs, None)
    m, _ = syn.pose_inputs(0)
    assert needs_module_path(dec, specs, m) is None                      # with poses the same decoder runs on the HIP kernels
    specs3 = syn.specs_for("nerf3")
    assert needs_module_path(build_decoder(specs3, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()}), specs3, None) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(syn.VARIANTS))
def test_variants_through_the_drop_in_functions(name, golden_dir):
    """decode_sdf_multi_output and the two-pass flow at N = 32 for every variant: volumes within 1e-5 of the reference's,
    negative-voxel boxes and zoom cube equal."""
    from alignsdf_amd.torch_decoder import TorchModuleDecoder
    from alignsdf_amd.utils.mesh import decode_two_pass
    from alignsdf_amd.utils.utils import decode_sdf_multi_output, decoder_for
    g = np.load("%s/ref_variant_%s.npz" % (golden_dir, name))
    specs, dec, mano, obj, cam, latent = _module(name)
    assert isinstance(decoder_for(dec, specs, mano), TorchModuleDecoder)
    h, o, _ = decode_sdf_multi_output(dec, latent.cuda(), torch.from_numpy(g["rand_pts"]).cuda(), mano, cam, specs, obj_results=obj)
    assert np.abs(h[:, 0].cpu().numpy() - g["rand_hand"]).max() <= 1e-5 and np.abs(o[:, 0].cpu().numpy() - g["rand_obj"]).max() <= 1e-5
    r = decode_two_pass(True, True, dec, latent.cuda(), mano, obj, specs, 32, cam_intr=cam)
    assert np.array_equal(np.stack([r["bbox"][0:6], r["bbox"][8:14]]), g["bbox_32"])
    assert np.array_equal(r["voxel_size"].numpy().reshape(1), g["new_voxel_size_32"])
    assert np.array_equal(np.array(r["origin"], dtype=np.float32), g["new_origin_32"])
    assert np.abs(r["vol_hand"].cpu().numpy() - g["vol2_hand_32"]).max() <= 1e-5
    assert np.abs(r["vol_obj"].cpu().numpy() - g["vol2_obj_32"]).max() <= 1e-5


@pytest.mark.gpu
def test_variant_through_the_sample_pipeline_and_files(tmp_path):
    """create_mesh_combined_decoder writes both files for a module-path decoder; the pipeline yields its surfaces."""
    from alignsdf_amd.reconstruct import pipelined_two_pass
    from alignsdf_amd.utils.mesh import create_mesh_combined_decoder
    specs, dec, mano, obj, cam, latent = _module("tanh")
    stats = create_mesh_combined_decoder(True, True, False, dec, latent.cuda(), None, None, None, specs, str(tmp_path / "s"), N=32)
    assert stats["hand"][1] > 100 and stats["obj"][1] > 100 and (tmp_path / "s_hand.ply").exists() and (tmp_path / "s_obj.ply").exists()
    out = list(pipelined_two_pass(dec, specs, [(k, latent.cuda(), None, None) for k in range(2)], 32))
    assert [k for k, _ in out] == [0, 1] and all(r["F_hand"] == stats["hand"][1] and r["F_obj"] == stats["obj"][1] for _, r in out)


def test_model_output_adapter_contract():
    """model_output_code_source: decode_model_output's return triple -> the (latent, mano, obj) the drivers consume."""
    from alignsdf_amd.frontend import model_output_code_source
    m, o = syn.pose_inputs(0)
    mano = {k: torch.from_numpy(v) for k, v in m.items()}
    mano.update(verts=torch.zeros(1, 778, 3), joints=torch.zeros(1, 21, 3), shape=torch.zeros(1, 10), pcas=torch.zeros(1, 15))
    obj = {k: torch.from_numpy(v) for k, v in o.items()}
    src = model_output_code_source(lambda name, i: (torch.from_numpy(syn.latent_code(i)).double(), mano, obj), device="cpu")
    lat, mm, oo = src("00000012", 3)
    assert lat.dtype == torch.float32 and lat.shape == (1, 256) and set(mm) == {"global_trans", "rot_center", "joints"} and set(oo) == {"obj_trans"}
    lat, mm, oo = model_output_code_source(lambda name, i: (torch.zeros(1, 256), None, None), device="cpu")("x", 0)
    assert mm is None and oo is None


def test_resnet18_like_shapes_cpu():
    from alignsdf_amd.frontend import ResNet18Like
    enc = ResNet18Like().eval()
    with torch.no_grad():
        z = enc(torch.zeros(1, 3, 64, 64))
    assert z.shape == (1, 256) and sum(p.numel() for p in enc.parameters()) > 11_000_000


@pytest.mark.gpu
def test_encoder_in_the_sample_pipeline(monkeypatch):
    """Codes produced on the device by an encoder in the loop (no host synchronisation) give the same surfaces as the same
    codes fed as resident tensors."""
    monkeypatch.setenv("ASDF_FINE", "exact")      # the pass-2 VOLUMES are compared below: ordinary sweeps (the pipeline's default
                                                  # narrow-band volumes are only defined where marching cubes reads values)
    from alignsdf_amd.frontend import ResNet18Like, encoder_code_source
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.reconstruct import pipelined_two_pass
    specs = syn.specs_for("nerf3")
    dec = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    torch.manual_seed(1)
    enc = ResNet18Like().cuda().eval()
    imgs = [torch.rand(1, 3, 128, 128) for _ in range(3)]
    src = encoder_code_source(enc, lambda name, i: imgs[i])
    with torch.no_grad():
        fixed = [enc(im.cuda()).clone() for im in imgs]
    a = list(pipelined_two_pass(dec, specs, ((i, *src("s", i)) for i in range(3)), 32))
    b = list(pipelined_two_pass(dec, specs, ((i, fixed[i], None, None) for i in range(3)), 32))
    for (_, ra), (_, rb) in zip(a, b):        # (the convolutions need not be bit-reproducible from call to call)
        assert (ra["vol_hand"] - rb["vol_hand"]).abs().max().item() <= 1e-5 and abs(ra["F_obj"] - rb["F_obj"]) <= 16
    dec.close()
