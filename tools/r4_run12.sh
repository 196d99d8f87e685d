#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_coarse_box.py tests/test_gpu_default_sweeps.py tests/test_gpu_sweep_fallbacks.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -8
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['shader_clock_ghz'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps > /dev/null 2>&1
python - <<'PY'
import glob, csv
for f in glob.glob("/tmp/st/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
