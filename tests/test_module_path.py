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
    dec = getattr(arch, cls)(specs["LatentSize"], specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"]).eval()
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    return specs, dec, t(mano), t(obj), None if cam is None else torch.from_numpy(cam), torch.from_numpy(latent)


@pytest.mark.parametrize("name", ["tanh", "layernorm", "xyzall", "narrow"])
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
    for name, word in (("tanh", "use_tanh"), ("layernorm", "LayerNorm"), ("xyzall", "xyz_in_all"), ("pixelalign", "PixelAlign"),
                       ("narrow", "LatentSize 128")):
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
    stats = create_mesh_combined_decoder(True, True, False, dec, latent.cuda(), None, None, None, specs, str(tmp_path / "s"), N=32, return_stats=True)
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


# ---- round 3: decoder generality --------------------------------------------------------------------------

def test_shape_routing_of_legal_network_specs():
    """NetworkSpecs other than dims [512] * 4 / latent_in [2] / LatentSize 256 are legal (networks/model.py:192-282): they must be
    ROUTED to the module path, not refused with ASDF_EINVAL."""
    from alignsdf_amd.hip_decoder import unsupported_reason
    from alignsdf_amd.networks.model import CombinedDecoder, SeparateDecoder
    from alignsdf_amd.torch_decoder import needs_module_path
    wn = dict(weight_norm=True)
    cases = [
        (SeparateDecoder(128, 3, "nerf", [256] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], **wn), "LatentSize 128"),
        (SeparateDecoder(256, 3, "nerf", [512] * 3, latent_in=[2], norm_layers=[0, 1, 2], **wn), "five layers"),
        (SeparateDecoder(256, 3, "nerf", [512] * 5, latent_in=[2], norm_layers=[0, 1, 2, 3, 4], **wn), "five layers"),
        (SeparateDecoder(256, 3, "nerf", [512] * 4, latent_in=[1], norm_layers=[0, 1, 2, 3], **wn), "linh0"),
        (SeparateDecoder(256, 3, "nerf", [512, 768, 512, 512], latent_in=[2], norm_layers=[0, 1, 2, 3], **wn), "linh1"),
        (SeparateDecoder(256, 3, "nerf", [512] * 4, latent_in=[], norm_layers=[0, 1, 2, 3], **wn), "linh1"),
        (SeparateDecoder(256, 21, "nerf", [512] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], **wn), "PointFeatSize 21"),
        (CombinedDecoder(256, 3, "nerf", [384] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], **wn), "lin0"),
        (torch.nn.Linear(259, 1), "not a SeparateDecoder"),
    ]
    for dec, word in cases:
        why = needs_module_path(dec, None, None)
        assert why is not None and word in why, (word, why)
    ok = SeparateDecoder(256, 3, "nerf", [512] * 4, latent_in=[2], norm_layers=[0, 1, 2, 3], **wn)
    assert needs_module_path(ok, None, None) is None
    assert unsupported_reason(ok.state_dict(), 256, 3, "nerf") is None
    with pytest.raises(NotImplementedError, match="module path"):
        from alignsdf_amd.hip_decoder import HipSdfDecoder
        HipSdfDecoder(cases[0][0].state_dict(), 128, 3, "nerf")          # (raised before any device is touched)


def test_latent_free_module_matches_the_reference_cpu(golden_dir):
    """The fixture's network is what the reference evaluated (same parameters, same outputs on the CPU)."""
    g = np.load(golden_dir + "/ref_legacy_free.npz")
    net = syn.latent_free_module()
    with torch.no_grad():
        out = net(torch.from_numpy(g["rand_pts"]))
    assert np.abs(out[:, 0].numpy() - g["rand_sdf"]).max() <= 1e-6


@pytest.mark.gpu
def test_legacy_entry_points_take_any_single_output_module(golden_dir, tmp_path):
    """deep_sdf.utils.decode_sdf / deep_sdf.mesh.create_mesh with latent_vector None and a plain nn.Module
    (deep_sdf/utils.py:64-75, deep_sdf/mesh.py:14-61): values within 1e-5 of the reference's run, the mesh written."""
    from alignsdf_amd.deep_sdf import mesh as legacy
    from alignsdf_amd.deep_sdf.utils import decode_sdf
    from alignsdf_amd.marching_cubes import marching_cubes_lewiner
    g = np.load(golden_dir + "/ref_legacy_free.npz")
    net = syn.latent_free_module()                       # CPU module: evaluated through a device copy, left where it is
    sdf = decode_sdf(net, None, torch.from_numpy(g["rand_pts"]).cuda())
    assert sdf.shape == (4096, 1) and np.abs(sdf[:, 0].cpu().numpy() - g["rand_sdf"]).max() <= 1e-5
    assert next(net.parameters()).device.type == "cpu" and not net.training
    seen = {}
    real = legacy.convert_sdf_samples_to_ply
    legacy.convert_sdf_samples_to_ply = lambda vol, origin, vs, path: seen.setdefault("vol", vol.cpu().numpy()) is None or real(vol, origin, vs, path)
    try:
        pts, faces = legacy.create_mesh(net, None, str(tmp_path / "free"), N=32)
    finally:
        legacy.convert_sdf_samples_to_ply = real
    assert np.abs(seen["vol"] - g["vol_32"]).max() <= 1e-5
    v, f = marching_cubes_lewiner(torch.from_numpy(g["vol_32"]).cuda(), 0.0, spacing=[float(g["voxel_size"][0])] * 3)
    assert abs(len(faces) - len(f)) <= 8 and (tmp_path / "free.ply").exists()
    # a module in TRAINING mode comes back in training mode, and a latent-carrying generic module works the same way
    net.train()
    decode_sdf(net, None, torch.from_numpy(g["rand_pts"][:64]).cuda())
    assert net.training
    wide = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1)).eval()
    lat = torch.randn(1, 4)
    q = torch.from_numpy(g["rand_pts"][:100])
    with torch.no_grad():
        want = wide(torch.cat([lat.expand(100, -1), q], 1))
    got = decode_sdf(wide, lat.cuda(), q.cuda())
    assert np.abs(got.cpu().numpy() - want.numpy()).max() <= 1e-5


@pytest.mark.gpu
def test_module_path_does_not_keep_or_move_the_callers_module():
    """the module path must not move the caller's module to the GPU or pin it (and its device copy) forever."""
    import gc
    import weakref
    from alignsdf_amd.utils.utils import decoder_for
    specs, dec, mano, obj, cam, latent = _module("tanh")
    dec.train()
    ev = decoder_for(dec, specs, mano)
    ev.set_sample(latent.cuda())
    ev.decode_points(torch.zeros(8, 3).cuda())
    assert dec.training and all(p.device.type == "cpu" for p in dec.parameters())
    first = ev.module
    with torch.no_grad():
        dec.linh0.bias.add_(1.0)                  # an optimiser step: the device copy must follow
    assert ev.module is not first
    ref = weakref.ref(dec)
    del dec, first
    gc.collect()
    assert ref() is None                          # nothing holds the module but the caller


# ---- round 3: image files -> encoder -> sample pipeline (reconstruct.py:54-84) -----------------------------------

def _write_images(root, names, size=(300, 280), fmt="JPEG"):
    from PIL import Image
    rng = np.random.default_rng(5)
    for n in names:
        a = (rng.random(size + (3,)) * 255).astype(np.uint8)
        Image.fromarray(a).save(str(root / (n + (".jpg" if fmt == "JPEG" else ".png"))), fmt, quality=95)


def test_image_loader_reproduces_the_reference_transform_cpu(tmp_path):
    """Centre crop to ImageSize, [0, 1], ImageNet normalisation (utils/data.py:217-244); a picture smaller than the crop is
    padded with black; the prefetcher hands the images out in request order."""
    from PIL import Image
    from alignsdf_amd.frontend import IMAGENET_MEAN, IMAGENET_STD, ImageFilePrefetcher, load_image_tensor
    _write_images(tmp_path, ["a", "b", "c"], fmt="PNG")
    raw = np.asarray(Image.open(str(tmp_path / "a.png")).convert("RGB"))
    want = (torch.from_numpy(raw[22:278, 12:268].copy()).permute(2, 0, 1).float() / 255 - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / \
        torch.tensor(IMAGENET_STD).view(3, 1, 1)
    got = load_image_tensor(str(tmp_path / "a.png"), (256, 256))
    assert got.shape == (1, 3, 256, 256) and torch.equal(got[0], want)
    small = load_image_tensor(str(tmp_path / "a.png"), (320, 320))
    assert torch.equal(small[0, :, 10:310, 20:300], (torch.from_numpy(raw.copy()).permute(2, 0, 1).float() / 255 - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) /
                       torch.tensor(IMAGENET_STD).view(3, 1, 1))
    assert torch.allclose(small[0, :, 0, 0], (0 - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD))
    pre = ImageFilePrefetcher(str(tmp_path), ["a", "b", "c"], ext=".png", device="cpu")
    outs = [pre(n, i) for i, n in enumerate(["a", "b", "c"])]
    pre.close()
    assert torch.equal(outs[0], got) and not torch.equal(outs[1], outs[0]) and pre.decode_seconds > 0


@pytest.mark.gpu
def test_reconstruct_from_image_files(tmp_path):
    """reconstruct() fed from JPEG files: prefetching loader thread -> encoder on the device -> sample pipeline -> PLY files; the
    surfaces equal those obtained with the same images decoded up front."""
    import json
    from alignsdf_amd import reconstruct as rc
    from alignsdf_amd.frontend import ImageFilePrefetcher, ResNet18Like, encoder_code_source, load_image_tensor
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.ply import read_ply
    names = ["%08d" % i for i in range(4)]
    img_root = tmp_path / "rgb"
    img_root.mkdir()
    _write_images(img_root, names)
    split = tmp_path / "split.json"
    split.write_text(json.dumps({"filenames": ["x/%s.jpg" % n for n in names]}))
    specs = syn.specs_for("nerf3")
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
    torch.manual_seed(3)
    enc = ResNet18Like().cuda().eval()
    pre = ImageFilePrefetcher(str(img_root), names, image_size=(256, 256))
    recs = rc.reconstruct(dec, specs, str(split), str(tmp_path / "out"), 0, 4, cube_dim=32, code_source=encoder_code_source(enc, pre))
    pre.close()
    assert [r["name"] for r in recs] == names and all(r["F_hand"] > 0 for r in recs)
    fixed = {n: load_image_tensor(str(img_root / (n + ".jpg")), (256, 256)) for n in names}
    recs2 = rc.reconstruct(dec, specs, str(split), str(tmp_path / "out2"), 0, 4, cube_dim=32,
                           code_source=encoder_code_source(enc, lambda name, i: fixed[name]))
    for a, b in zip(recs, recs2):          # (the convolutions need not be bit-reproducible from call to call)
        assert abs(a["F_hand"] - b["F_hand"]) <= 16 and abs(a["F_obj"] - b["F_obj"]) <= 16
    v, f = read_ply(str(tmp_path / "out" / "meshes" / (names[2] + "_hand.ply")))
    assert len(f) == recs[2]["F_hand"] or len(f) > 0
