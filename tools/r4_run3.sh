#!/bin/bash
# round 4, GPU call 3: short-list form parity, fallback tests, power trace, small-lattice bench lines
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_short_list.py tests/test_gpu_sweep_fallbacks.py tests/test_gpu_default_sweeps.py::test_audit_record_of_the_c_abi tests/test_gpu_refine.py tests/test_gpu_coarse_box.py -q -x 2>&1 | tail -40 > gpurun_out/r4/pytest3.log
cat gpurun_out/r4/pytest3.log
timeout 300 python tools/power_trace.py --seconds 10 > gpurun_out/r4/power_trace.txt 2> gpurun_out/r4/power_trace.err
cat gpurun_out/r4/power_trace.txt; tail -3 gpurun_out/r4/power_trace.err
for cfg in "--grid 64 --branches hand" "--grid 128"; do
  timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-other-configs --no-other-math --no-other-sweeps $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], d['ms_per_step'], d['value'], d['roofline']['launch_ms'])"
done
