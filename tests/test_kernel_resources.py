"""Register / scratch budget of the shipped decoder kernel, checked at compile time (hipcc cross-compiles without a
GPU).  A spill to scratch in the MFMA stream costs several percent and is invisible to the parity tests."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alignsdf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    """One device-only compile of every decoder translation unit (side by side): (resource-usage remarks, ISA text)."""
    out_dir = tmp_path_factory.mktemp("isa")
    from alignsdf_amd.build_native import TU_FLAGS          # (the per-unit flags of the shipped build: MFMA accumulators in VGPRs for some)
    units = ["decoder.hip", "k1_kernels.hip", "k1_cls_kernels.hip", "k1h_kernels.hip", "k1h_nerf_kernels.hip", "k1s_kernels.hip", "k1s_nerf_kernels.hip"]
    procs = [(u, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *TU_FLAGS.get(u, []), "-S",
                                   "--cuda-device-only", u, "-o", str(out_dir / (u + ".s")), "-Rpass-analysis=kernel-resource-usage"],
                                  cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for u in units]
    remarks, isa = "", ""
    for u, proc in procs:
        _, err = proc.communicate(timeout=900)
        assert proc.returncode == 0, (u, err[-2000:])
        remarks += err
        isa += (out_dir / (u + ".s")).read_text()
    return remarks, isa


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_main_decoder_kernel_has_no_scratch(compiled):
    text = compiled[0]
    blocks = re.split(r"remark: Function Name: ", text)
    stats = {}
    for b in blocks[1:]:
        name = b.split()[0]
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        stats[name] = dict(vgpr=get(r"VGPRs"), agpr=get(r"AGPRs"), scratch=get(r"ScratchSize \[bytes/lane\]"),
                           occupancy=get(r"Occupancy \[waves/SIMD\]"))
    main = [v for k, v in stats.items() if "14sdf_mlp_kernelE" in k]
    assert len(main) == 1, list(stats)
    assert main[0]["scratch"] == 0 and main[0]["vgpr"] + main[0]["agpr"] <= 512 and main[0]["occupancy"] == 1, main[0]
    # the label-pass kernel as well
    k = [v for name, v in stats.items() if "18sdf_mlp_cls_kernelE" in name]
    assert len(k) == 1 and k[0]["scratch"] == 0 and k[0]["occupancy"] == 1, k
    # the split-half kernel (the default arithmetic) sits exactly at the 512-register limit and keeps a few per-thread
    # constants (addresses computed once per kernel) in scratch: allowed only outside the MFMA stream - see below
    k = [v for name, v in stats.items() if "18sdf_mlp_f16_kernelE" in name]
    assert len(k) == 1 and k[0]["scratch"] == 0 and k[0]["vgpr"] + k[0]["agpr"] <= 512 and k[0]["occupancy"] == 1, k
    # round 2: the CombinedDecoder form and the NeRF-encoded forms (PointFeatSize 9: the one any config could plausibly
    # use) are scratch-free too; only the 8-K-step form of PointFeatSize 15 still spills (a few hundred bytes)
    for frag in ("27sdf_mlp_f16_combined_kernelE", "24sdf_mlp_f16_nerf9_kernelE", "33sdf_mlp_f16_combined_nerf9_kernelE",
                 "34sdf_mlp_f16_combined_nerf15_kernelE"):
        k = [v for name, v in stats.items() if frag in name]
        assert len(k) == 1 and k[0]["scratch"] == 0 and k[0]["occupancy"] == 1, (frag, k)
    k = [v for name, v in stats.items() if "25sdf_mlp_f16_nerf15_kernelE" in name]
    assert len(k) == 1 and k[0]["scratch"] <= 512, k
    # the one-plane kernels of the box-only / narrow-band sweeps (opt-in paths): the CombinedDecoder form is scratch-free; the
    # SeparateDecoder form carries two point groups per wave (the register footprint of the split-half kernel plus a second
    # set of accumulators) and keeps a few dozen per-thread constants in scratch
    k = [v for name, v in stats.items() if "29sdf_mlp_f16p1_combined_kernelE" in name]
    assert len(k) == 1 and k[0]["scratch"] == 0 and k[0]["occupancy"] == 1, k
    k = [v for name, v in stats.items() if "20sdf_mlp_f16p1_kernelE" in name]
    assert len(k) == 1 and k[0]["scratch"] <= 256 and k[0]["vgpr"] + k[0]["agpr"] <= 512 and k[0]["occupancy"] == 1, k
    # round 4: the one-plane instantiations of the NeRF-encoded decoders (SeparateDecoder: two point groups per wave) - scratch-free
    # except the forms with 8 K-steps of point features (a few dozen / hundred bytes of per-thread constants)
    for frag in ("26sdf_mlp_f16p1_nerf9_kernelE", "35sdf_mlp_f16p1_combined_nerf9_kernelE"):
        k = [v for name, v in stats.items() if frag in name]
        assert len(k) == 1 and k[0]["scratch"] == 0 and k[0]["vgpr"] + k[0]["agpr"] <= 512 and k[0]["occupancy"] == 1, (frag, k)
    for frag, limit in (("27sdf_mlp_f16p1_nerf15_kernelE", 64), ("36sdf_mlp_f16p1_combined_nerf15_kernelE", 256)):
        k = [v for name, v in stats.items() if frag in name]
        assert len(k) == 1 and k[0]["scratch"] <= limit and k[0]["occupancy"] == 1, (frag, k)
    # the streaming kernels must not touch scratch either
    for k, v in stats.items():
        if "fold_sample" in k or "neg_bbox" in k:
            assert v["scratch"] == 0, (k, v)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_scratch_traffic_inside_the_mfma_stream(compiled):
    """In the ISA of the shipped decoder kernels every scratch load / store must come before the first MFMA of the kernel
    body (kernel / head-loop prologue) - a spill between MFMAs costs whole percents and no parity test sees it."""
    text = compiled[1]
    for mangled in ("_ZN4asdf14sdf_mlp_kernelENS_12DecodeParamsE", "_ZN4asdf18sdf_mlp_f16_kernelENS_12DecodeParamsE",
                    "_ZN4asdf27sdf_mlp_f16_combined_kernelENS_12DecodeParamsE", "_ZN4asdf24sdf_mlp_f16_nerf9_kernelENS_12DecodeParamsE",
                    "_ZN4asdf33sdf_mlp_f16_combined_nerf9_kernelENS_12DecodeParamsE",
                    "_ZN4asdf34sdf_mlp_f16_combined_nerf15_kernelENS_12DecodeParamsE"):
        body = text[text.index(mangled + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        mfma = [i for i, l in enumerate(body) if "v_mfma_" in l]
        scratch = [i for i, l in enumerate(body) if "scratch_" in l and not l.strip().startswith(";")]
        assert len(mfma) > 2000
        inside = [i for i in scratch if mfma[0] < i < mfma[-1]]
        assert not inside, "%s: %d scratch accesses inside the MFMA stream, first at line %d: %s" % (
            mangled, len(inside), inside[0], body[inside[0]].strip())
    # the two-group one-plane kernel: at most a handful of spill reloads between its 2048 MFMAs per tile (tile boundaries)
    body = text[text.index("_ZN4asdf20sdf_mlp_f16p1_kernelENS_12DecodeParamsE:"):]
    body = body[:body.index("s_endpgm")].splitlines()
    mfma = [i for i, l in enumerate(body) if "v_mfma_f32_32x32x16" in l]
    scratch = [i for i, l in enumerate(body) if "scratch_" in l and not l.strip().startswith(";")]
    # (2048 of the three 512-wide layers + 32 of layer 0, whose point features are an fp16 operand since round 3)
    assert len(mfma) == 2048 + 32 and len([i for i in scratch if mfma[0] < i < mfma[-1]]) <= 8
    # round 4: built with MFMA accumulators in VGPRs (build_native.TU_FLAGS) the epilogues read them in place - the default form read
    # every accumulator back out of the AGPRs (1 862 v_accvgpr_read per tile body, 4.6 % of the kernel's time)
    assert len([l for l in body if "v_accvgpr_read" in l]) <= 200 and not scratch
