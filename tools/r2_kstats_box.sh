# per-kernel statistics of the box-only coarse sweep (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_kb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -- python $GRAFT_REPO_ROOT/tools/time_coarse_box.py > /tmp/prof_kb.txt 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/prof_kb/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("asdf::", "")
        if float(r["Percentage"]) > 0.01:
            print("%-44s calls %4s avg %10.1f us  min %9.1f  max %9.1f  %6s%%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
