#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_default_N256.json 2> gpurun_out/r4/bench_default_N256.err
tail -c 400 gpurun_out/r4/bench_default_N256.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_default_N256.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "dtype", d["dtype"][:60])
r = dict(d["roofline"]); r.pop("note")
print(json.dumps(r))
print("mc", json.dumps(d.get("roofline_marching_cubes")))
print("other_sweeps", d["other_sweeps"]["value"], d["other_sweeps"]["ms_per_step"], d["other_sweeps"]["launch_ms"])
print("other_math", d["other_math"]["value"], d["other_math"]["ms_per_step"], d["other_math"]["launch_ms"])
print("parity", json.dumps({k: v for k, v in d["parity_in_run"].items() if k != "against_reference_runs"}))
print("ref runs", [(x["sample"], x["zoom_cube_bit_equal"], x["V_F_equal_reference"]) for x in d["parity_in_run"]["against_reference_runs"]])
for c in d["other_configs"]:
    print(c["config"][:70], "ms", round(c["ms_per_step"], 3), "V/F == ref", c["V_F_equal_reference"], "refused", c["sweeps"]["refused_sweeps"], "tau", c["sweeps"]["allowance_now"])
cb = d["cpu_baseline"]
print("cpu", {k: cb[k] for k in ("value", "unit", "cores", "kind", "gpu_over_cpu") if k in cb})
PY
