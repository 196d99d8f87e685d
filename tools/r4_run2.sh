#!/bin/bash
# round 4, GPU call 2: the GPU suite on the new audit / state machine, a bench line with the certificate + clock
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_default_sweeps.py 2>&1 | tail -40 > gpurun_out/r4/pytest2.log
cat gpurun_out/r4/pytest2.log
timeout 900 python -m pytest tests/test_gpu_default_sweeps.py -q -s 2>&1 | tail -60 > gpurun_out/r4/pytest2b.log
cat gpurun_out/r4/pytest2b.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench2.json 2> gpurun_out/r4/bench2.err
tail -c 1500 gpurun_out/r4/bench2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench2.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print("roofline", json.dumps(d["roofline"], indent=1))
print("certificate", json.dumps(d["sweeps"]["certificate"], indent=1))
print("parity", json.dumps(d.get("parity_in_run"), indent=1)[:3000])
for c in d.get("other_configs", []):
    print(c["config"], c["ms_per_step"], c["V_F_sample0"], c["V_F_sample0_reference_golden"], c["sweeps"]["refused_sweeps"], c["sweeps"]["certificate"])
PY
