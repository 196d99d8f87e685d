#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r4/pytest5.log
cat gpurun_out/r4/pytest5.log
timeout 300 python tools/power_trace.py --seconds 10 > gpurun_out/r4/power_trace.txt 2> gpurun_out/r4/power_trace.err
cat gpurun_out/r4/power_trace.txt; tail -3 gpurun_out/r4/power_trace.err
timeout 300 python tools/one_plane_error_bound.py > gpurun_out/r4/one_plane_error_bound.txt 2> gpurun_out/r4/one_plane_error_bound.err
cat gpurun_out/r4/one_plane_error_bound.txt; tail -3 gpurun_out/r4/one_plane_error_bound.err
for cfg in "--grid 64 --branches hand" "--grid 128"; do
  timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-other-configs --no-other-math --no-other-sweeps $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], d['ms_per_step'], d['value'], d['roofline']['launch_ms'])"
done
