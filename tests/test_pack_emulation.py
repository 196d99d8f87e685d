"""CPU check of the HIP decoder's data layouts: the C++ weight packer's real output is run through a
lane-accurate emulation of the K0/K1 index logic (tests/kernel_emulator.py) and compared with the oracle."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import kinematic_affine
from oracle import sdf_oracle as orc
from tests import kernel_emulator as emu


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9"])
def test_packed_stream_emulation_matches_oracle(tag, native_lib):
    specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
    pk = emu.pack_host(sd, specs["PointFeatSize"], specs["EncodeStyle"])
    lat = syn.latent_code(0)
    mano = obj = emb = None
    if tag == "both9":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
        emb = kinematic_affine(9, "both", specs["SdfScaleFactor"], mano, obj)
    cst = emu.fold(pk, lat, emb)
    pts = syn.uniform((32, 3), 5, -1, 1).astype(np.float32)
    h, o = emu.run_wave(pk, cst, pts)
    rh, ro = orc.decode_points(sd, lat, pts, specs, mano, obj)
    assert np.abs(h - rh.numpy()).max() <= 2e-6 and np.abs(o - ro.numpy()).max() <= 2e-6


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3"])
def test_split_half_image_emulation_matches_oracle(tag, native_lib):
    """The split-half image (two fp16 planes per weight, per-layer power-of-two scales, pre-scaled constants) through an
    emulation of sdf_mlp_f16_kernel.h's data flow: fp32-class agreement with the oracle, and the scales are exact."""
    specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
    pk = emu.pack_host(sd, specs["PointFeatSize"], specs["EncodeStyle"])
    lat = syn.latent_code(0)
    mano = obj = emb = None
    if tag == "both9":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
        emb = kinematic_affine(9, "both", specs["SdfScaleFactor"], mano, obj)
    for s2 in pk["s2"][:len(pk["pf"])]:
        assert s2 > 0 and np.log2(s2) == np.round(np.log2(s2))                  # powers of two
    planes = pk["stream16"].view(np.float16).astype(np.float32).reshape(256, 8, 2, 64, 8)
    used = planes[:128 * len(pk["pf"])]
    assert 512.0 <= np.abs(used[:, :, 0]).max() < 1024.0                          # hi planes: max |w| S_w in [512, 1024)
    assert np.abs(used[:, :, 1]).max() <= 0.25 + 1e-6                            # lo planes: at most half an ulp of the hi plane
    cst = emu.fold16(pk, lat, emb)
    pts = syn.uniform((32, 3), 5, -1, 1).astype(np.float32)
    h, o = emu.run_wave16(pk, cst, pts)
    rh, ro = orc.decode_points(sd, lat, pts, specs, mano, obj)
    assert np.abs(h - rh.numpy()).max() <= 2e-6 and np.abs(o - ro.numpy()).max() <= 2e-6


def test_kinematic_affine_matches_oracle_embedding():
    specs = syn.specs_for("both9")
    m, o = syn.pose_inputs(3)
    mano = {k: torch.from_numpy(v) for k, v in m.items()}
    obj = {k: torch.from_numpy(v) for k, v in o.items()}
    pts = torch.from_numpy(syn.uniform((500, 3), 11, -1, 1).astype(np.float32))
    ref = orc.kinematic_embedding(pts, mano, 9, specs["SdfScaleFactor"], obj, "both").double().numpy()
    Eh, Eo = kinematic_affine(9, "both", specs["SdfScaleFactor"], mano, obj)
    x = pts.double().numpy()
    hand = x @ Eh[:, :3].T + Eh[:, 3]
    objf = x @ Eo[:, :3].T + Eo[:, 3]
    assert np.abs(hand - ref[:, :6]).max() <= 2e-6
    assert np.abs(objf[:, :3] - ref[:, :3]).max() <= 2e-6 and np.abs(objf[:, 3:] - ref[:, 6:]).max() <= 2e-6
    # all-joint variants are affine too (PointFeatSize 51 'hand')
    ref51 = orc.kinematic_embedding(pts, mano, 51, specs["SdfScaleFactor"], obj, "hand").double().numpy()
    Eh51, Eo51 = kinematic_affine(51, "hand", specs["SdfScaleFactor"], mano, obj)
    assert Eh51.shape == (51, 4) and np.abs(x @ Eh51[:, :3].T + Eh51[:, 3] - ref51).max() <= 2e-6


def test_library_exports_every_declared_symbol(native_lib):
    import re, os
    from alignsdf_amd import _native
    hdr = open(os.path.join(os.path.dirname(_native.__file__), "..", "include", "alignsdf_hip.h")).read()
    declared = set(re.findall(r"\b(asdf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS)
    for name in declared:
        assert hasattr(native_lib, name), name
    assert native_lib.asdf_version() == _native.ABI_VERSION
    assert native_lib.asdf_strerror(-6).decode().startswith("Surface level")
