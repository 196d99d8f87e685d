# Round-3 check on an MI355X: the whole -m gpu suite, then the default bench line.   gpurun -- 'bash tools/r3_check.sh [tag]'
cd $GRAFT_REPO_ROOT; T=${1:-a}; mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q > gpurun_out/r3/gpu_tests_$T.log 2>&1; echo "gpu tests rc=$?"
tail -n 25 gpurun_out/r3/gpu_tests_$T.log
python bench.py ${BENCH_ARGS:---steps 8 --warmup 2} > gpurun_out/r3/bench_$T.json 2> gpurun_out/r3/bench_$T.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3/bench_$T.err
python - <<PY
import json
b = json.load(open("gpurun_out/r3/bench_$T.json"))
print("value", b["value"], "ms/step", b["ms_per_step"], "launch_ms", b["roofline"]["launch_ms"], "frac", b["roofline"]["frac"], "refused", b["sweeps"]["refused_sweeps"])
print("parity", json.dumps(b.get("parity_in_run"))[:900])
for c in b.get("other_configs", []):
    print(c["config"], c["ms_per_step"], c["V_F_equal_reference"], c["sweeps"]["refused_sweeps"])
PY
