# Kernel trace of the eval-mode file flow: where does the time over the plain sample pipeline go - GPU work or idle gaps?
#   gpurun -- 'bash tools/trace_eval_flow.sh'    -> gpurun_out/r3/trace_eval/summary.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${R:-r3}/trace_eval; rm -rf $O; mkdir -p $O
ASDF_TIMING_REPS=1 ASDF_TIMING_FLOW_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python tools/time_reconstruct_files.py 256 ${SAMPLES:-24} eval > $O/run.log 2>&1
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections
rows = []
for f in glob.glob("$O/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the timed reconstruct() call = the last S samples' worth of lattice sweeps (two per sample).  Steady state: the window from the
# first sweep of the call's 3rd sample to the first sweep of its last sample - the first samples fill the pipeline and the last one's
# post-processing has no next sample to hide behind (both are in the whole-call figures printed after it)
S = ${SAMPLES:-24}
# (the sweep kernel of the flow: the split-half kernel under the product's default - ordinary sweeps - and the one-plane kernel under ASDF_FAST=1)
KEY = "f16p1" if "${ASDF_FAST:-0}" not in ("", "0") else "sdf_mlp_f16_kernel"
p1 = [i for i, r in enumerate(rows) if KEY in r[2] and "subset" not in r[2]]
def window(first, last, samples, label):
    win = rows[first:last]
    t0, t1 = win[0][0], max(r[1] for r in win)
    busy = collections.Counter()
    last_end, idle, gaps = t0, 0, []
    for s, e, n in win:
        busy[n] += e - s
        if s > last_end:
            idle += s - last_end
            gaps.append((s - last_end, n))
        last_end = max(last_end, e)
    print("%s: window %.1f ms for %d samples = %.2f ms/sample; GPU idle %.2f ms/sample; %d launches/sample" % (
        label, (t1 - t0) / 1e6, samples, (t1 - t0) / 1e6 / samples, idle / 1e6 / samples, len(win) // samples))
    return busy, gaps
busy, gaps = window(p1[-2 * S + 4], p1[-2], S - 3, "steady state")
for n, v in busy.most_common(18):
    print("  %-70s %.3f ms/sample" % (n[:70], v / 1e6 / (S - 3)))
print("largest gaps (ms, before kernel):", [(round(g / 1e6, 2), n[:40]) for g, n in sorted(gaps, reverse=True)[:12]])
window(p1[-2 * S], len(rows), S, "whole call")
PY
grep -v '^[WE]2026' $O/run.log | tail -n 3 | tee -a $O/summary.txt
