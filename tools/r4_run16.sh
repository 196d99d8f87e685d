#!/bin/bash
cp tools/bin/libalignsdf_hip_poison.so alignsdf_amd/csrc/libalignsdf_hip.so
timeout 900 python -m pytest tests/test_gpu_coarse_box.py tests/test_gpu_default_sweeps.py tests/test_gpu_split_half_adversarial.py -q -x 2>&1 | tail -6
bash tools/r4_ab_many.sh inf0 poison
cp tools/bin/libalignsdf_hip_poison.so alignsdf_amd/csrc/libalignsdf_hip.so
