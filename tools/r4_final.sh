#!/bin/bash
# final measurements of round 4 on the committed tree: GPU suite, profiles (kernel stats, PMC, HBM traffic), the default bench line
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6
bash tools/profile_bench_r4.sh > gpurun_out/r4/profile.log 2>&1
rm -rf gpurun_out/r4/prof/stats gpurun_out/r4/prof/pmc_[0-9]
cp gpurun_out/r4/prof/hbm_traffic_f16p1.json profiles/r04_hbm_traffic_f16p1.json     # (so that the bench line below finds the traffic record of THIS kernel source)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_default_N256.json 2> gpurun_out/r4/bench_default_N256.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_default_N256.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
r = dict(d["roofline"]); r.pop("note"); print(json.dumps(r))
print("other_sweeps", d["other_sweeps"]["value"], d["other_sweeps"]["ms_per_step"], "other_math", d["other_math"]["value"])
for c in d["other_configs"]:
    print(c["config"][:60], round(c["ms_per_step"], 3), c["V_F_equal_reference"], c["sweeps"]["refused_sweeps"])
print(json.dumps(d["sweeps"]["certificate"]))
print("cpu", d["cpu_baseline"]["seconds_per_sample"], d["cpu_baseline"]["gpu_over_cpu"])
PY
grep -A26 "^asdf::sdf_mlp_f16p1_kernel" gpurun_out/r4/prof/pmc_summary.txt | grep -E "INSTS|WAIT_ANY|WAVE_CYCLES|MFMA_BUSY"
head -3 gpurun_out/r4/prof/kernel_stats.csv | cut -c1-110
