#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_coarse_box.py tests/test_gpu_default_sweeps.py tests/test_gpu_sweep_fallbacks.py tests/test_gpu_split_half_adversarial.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_math_modes.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r4/pytest7.log
cat gpurun_out/r4/pytest7.log
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs > gpurun_out/r4/bench7.json 2> gpurun_out/r4/bench7.err
tail -c 600 gpurun_out/r4/bench7.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench7.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
r = d["roofline"]; r.pop("note")
print(json.dumps(r, indent=1))
print(json.dumps(d["sweeps"]["certificate"]))
print(json.dumps(d["parity_in_run"]["against_ordinary_sweeps_f16x3"]))
PY
