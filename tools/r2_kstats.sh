# per-kernel statistics of a short bench run (rocprofv3 --kernel-trace --stats), names shortened
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-math --steps ${STEPS:-3} --warmup 1 > /tmp/prof_ks.json 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/prof_ks/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("asdf::", "")
        if float(r["Percentage"]) > 0.001 or "mc_" in n or "bbox" in n:
            print("%-44s calls %4s avg %10.1f us  min %9.1f  max %9.1f  %6s%%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
python3 -c "
import json; d=json.load(open('/tmp/prof_ks.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline_marching_cubes']['chain_ms_both_volumes'])"
