"""Register / scratch budget of the shipped decoder kernel, checked at compile time (hipcc cross-compiles without a
GPU).  A spill to scratch in the MFMA stream costs several percent and is invisible to the parity tests."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alignsdf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_main_decoder_kernel_has_no_scratch(tmp_path):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", "decoder.hip",
                          "-o", str(tmp_path / "d.o"), "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr
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
    # the streaming kernels must not touch scratch either
    for k, v in stats.items():
        if "fold_sample" in k or "neg_bbox" in k:
            assert v["scratch"] == 0, (k, v)
