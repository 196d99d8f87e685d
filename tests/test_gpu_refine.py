"""Near-level refinement of split-half sweeps (asdf_decoder_set_refine): voxels within tau of the iso level are recomputed on
the fp32 MFMA chain, so the SIGN pattern - all that the zoom cube (utils/mesh.py:208-237) and marching cubes
(utils/mesh.py:354) read - is the fp32 kernel's, voxel for voxel, and with it boxes, counts and surfaces."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _bound(tag, sample=0):
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.utils.utils import sample_embedding
    specs = syn.specs_for(tag)
    hip = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"])
    mano = obj = None
    if specs["EncodeStyle"] != "nerf":
        m, o = syn.pose_inputs(sample)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
        obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    hip.set_sample(torch.from_numpy(syn.latent_code(sample)).cuda(), sample_embedding(specs, mano, obj, hip.combined))
    return hip


@pytest.mark.parametrize("tag,N", [("nerf3", 128), ("both9", 128), ("nerf3", 256), ("comb3", 96), ("nerf9", 64)])
def test_signs_boxes_and_surfaces_equal_the_fp32_chain(tag, N):
    from alignsdf_amd.marching_cubes import marching_cubes_device
    hip = _bound(tag)
    assert hip.math == "f16x3" and hip.refine_tau == pytest.approx(4e-6)
    lattices = [([-1.0, -1.0, -1.0], 2.0 / (N - 1)), ([-0.62, -0.36, -0.37], 1.21 / (N - 1))]
    for origin, vs in lattices:
        out = {}
        for math in ("f32", "f16x3"):
            hip.set_math(math)
            out[math] = hip.decode_grid(N, origin, vs)
        for k in (0, 1):
            a, b = out["f32"][k], out["f16x3"][k]
            assert int(((a < 0) != (b < 0)).sum()) == 0
            d = (a - b).abs()
            assert d.max().item() <= 2e-6
            # inside the window the values ARE the fp32 kernel's (same kernel, same coordinates)
            near = a.abs() < 3e-6
            assert int(near.sum()) == 0 or d[near].max().item() == 0.0
            va, fa = marching_cubes_device(a, 0.0)
            vb, fb = marching_cubes_device(b, 0.0)
            assert torch.equal(fa, fb) and va.shape == vb.shape
        ba, bb = out["f32"][2].cpu().numpy(), out["f16x3"][2].cpu().numpy()
        assert np.array_equal(ba, bb)                 # boxes, counts - and the range-report words, both zero
    assert hip.range_violations() == 0
    hip.close()


def test_refinement_can_be_switched_off_and_argument_checks(native_lib):
    hip = _bound("nerf3")
    N = 64
    hip.set_refine(0.0)
    a = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
    hip.set_refine(4e-6)
    b = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
    changed = int((a[0] != b[0]).sum()) + int((a[1] != b[1]).sum())
    near = int(((a[0].abs() < 4e-6) | (a[1].abs() < 4e-6)).sum())
    assert changed <= 2 * near                        # only listed voxels may change (both heads are recomputed there)
    assert native_lib.asdf_decoder_set_refine(hip._h, -1.0) == -1 and native_lib.asdf_decoder_set_refine(None, 0.0) == -1
    hip.close()


def test_single_head_and_odd_sizes():
    """A head that is switched off is not touched by the refinement; N^3 not a multiple of 4 takes the scalar path."""
    hip = _bound("nerf3")
    for N in (33, 47):
        h, o, b = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), obj=False)
        hip.set_math("f32")
        h32, _, b32 = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), obj=False)
        hip.set_math("f16x3")
        assert o is None and int(((h < 0) != (h32 < 0)).sum()) == 0
        assert np.array_equal(b.cpu().numpy(), b32.cpu().numpy())
    hip.close()
