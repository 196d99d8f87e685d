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


@pytest.mark.parametrize("tag", ["nerf3", "both9"])
def test_w_form_image_is_the_same_weights_in_its_slot_order(tag, native_lib):
    """Round 6: the weight stream of the W form (v_mfma_f32_16x16x32_f16; pack.h: pack_decoder_f16, second image) holds exactly the fp16
    planes of the 32-wide form's image - same scales, same hi / lo values per (output row, input feature) - in the record and slot order
    sdf_mlp_f16w_kernel.h documents: record (tile t, feature half fh, K32-block j), lane l = row 32 t + 16 fh + (l & 15), slot
    (q = l >> 4, e) = input feature 32 j + 16 (e >> 2) + 4 q + (e & 3); layers 1 / 3 feature half outer (i = 16 fh + j), layer 2
    K32-block outer (i = 2 j + fh).  Both images are unpacked into dense [row][feature] matrices here and compared bit for bit."""
    specs = syn.specs_for(tag)
    pk = emu.pack_host(syn.full_state_dict(tag), specs["PointFeatSize"], specs["EncodeStyle"])
    old = pk["stream16"].reshape(2, 1024, 2, 64, 8)          # [head][record = (tile, K16-block)][plane][lane][e]
    new = pk["stream16w"].reshape(2, 1024, 2, 64, 8)
    lane = np.arange(64)[:, None]
    e = np.arange(8)[None, :]
    tile_row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
    layers = ((8, 32, 0), (16, 16, 256), (16, 32, 512))      # (tiles, records per tile, first record) of layers 1, 2, 3
    for head in range(2):
        for tiles, per, first in layers:
            K = per * 16
            dense_old = np.zeros((2, tiles * 32, K), np.uint16)
            dense_new = np.zeros((2, tiles * 32, K), np.uint16)
            for t in range(tiles):
                for i in range(per):
                    rec = first + t * per + i
                    # 32-wide form: record i = K16-block i; lane l: row 32 t + (l & 31), element e: feature 32 (i >> 1) + tile_row(8 (i & 1) + e, l >> 5)
                    rows = 32 * t + (lane & 31) + 0 * e
                    feats = 32 * (i >> 1) + tile_row(8 * (i & 1) + e, lane >> 5)
                    for plane in range(2):
                        dense_old[plane, rows, feats] = old[head, rec, plane]
                    fh, j = (i // 16, i % 16) if per == 32 else (i & 1, i >> 1)
                    rows = 32 * t + 16 * fh + (lane & 15) + 0 * e
                    feats = 32 * j + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3)
                    for plane in range(2):
                        dense_new[plane, rows, feats] = new[head, rec, plane]
            assert np.array_equal(dense_old, dense_new), (tag, head, tiles, per)
            assert dense_old[0].any()
