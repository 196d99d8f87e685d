#!/bin/bash
mkdir -p gpurun_out/r4
bash tools/profile_bench_r4.sh > gpurun_out/r4/profile.log 2>&1
tail -60 gpurun_out/r4/profile.log
rm -rf gpurun_out/r4/prof/stats gpurun_out/r4/prof/pmc_[0-9]
timeout 900 python bench.py --steps 256 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps > gpurun_out/r4/bench_sustained_256.json 2>/dev/null
timeout 900 python bench.py --tag grasp3 --steps 256 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps > gpurun_out/r4/bench_sustained_256_grasp3.json 2>/dev/null
timeout 900 python bench.py --tag grasp9 --steps 256 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps > gpurun_out/r4/bench_sustained_256_grasp9.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_sustained_256", "bench_sustained_256_grasp3", "bench_sustained_256_grasp9"):
    d = json.loads(open("gpurun_out/r4/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["shader_clock_ghz"], d["sweeps"]["refused_sweeps"], d["sweeps"]["certificate"])
PY
