# the round's closing measurements on one box: profile (kernel trace + PMC), smoke, the default bench line and the variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2
bash tools/profile_bench_r2.sh > gpurun_out/r2/prof_run.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2/smoke.txt 2>&1
python bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
python bench.py --coarse box --no-cpu-baseline --no-other-math > gpurun_out/r2/bench_coarse_box_main.json 2>/dev/null
python bench.py --grid 128 --no-cpu-baseline --no-other-math > gpurun_out/r2/bench_N128.json 2>/dev/null
python bench.py --tag both9 --no-cpu-baseline --no-other-math > gpurun_out/r2/bench_both9.json 2>/dev/null
python bench.py --math f32 --no-cpu-baseline --no-other-math --no-other-coarse > gpurun_out/r2/bench_f32.json 2>/dev/null
tail -2 gpurun_out/r2/smoke.txt
for f in default coarse_box_main N128 both9 f32; do python - <<PY
import json
d=json.loads(open("gpurun_out/r2/bench_$f.json").read().strip().splitlines()[-1])
o=d.get("other_coarse_pass") or {}
print("$f", round(d["value"],2), round(d["ms_per_step"],1), round(d["roofline"]["launch_ms"],2), d["roofline"].get("traffic"), d["config"].get("coarse_pass"), "| other coarse:", o.get("value"), o.get("zoom_cubes_equal_to_main_run"), o.get("V_F_equal_to_main_run"), "| parity", (d.get("parity_in_run") or {}).get("sign_differences"), (d.get("roofline_marching_cubes") or {}).get("chain_ms_both_volumes"))
PY
done
