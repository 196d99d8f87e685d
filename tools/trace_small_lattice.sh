# Kernel trace of the sample pipeline on a small lattice: GPU time by kernel and idle gaps per sample.
#   gpurun -- 'bash tools/trace_small_lattice.sh 64 hand'   (N, branches)  -> gpurun_out/r3/trace_small_<tag>_<N>/summary.txt   (TAG=both9 for the DexYCB decoder)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-64}; BR=${2:-hand}; S=${3:-64}
O=$GRAFT_REPO_ROOT/gpurun_out/${R:-r3}/trace_small_${TAG:-nerf3}_$N; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python bench.py --tag ${TAG:-nerf3} --grid $N --branches $BR --steps $S --warmup 4 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 > $O/bench.json 2> $O/run.err
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections, json
S = $S
rows = []
for f in glob.glob("$O/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
b = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench: %.3f ms/step, parity_in_run keys %s" % (b["ms_per_step"], list(b.get("parity_in_run", {}).keys())[:4]))
# the timed region: S samples = 2 S one-plane sweeps; parity_in_run's ordinary sweeps follow it, so take the LAST run of 2 S
# consecutive f16p1 launches that is not interleaved with ordinary-sweep kernels
p1 = [i for i, r in enumerate(rows) if "f16p1" in r[2]]
first, last = p1[-2 * S], p1[-1]
win = rows[first:last + 1]
t0, t1 = win[0][0], max(r[1] for r in win)
busy, calls = collections.Counter(), collections.Counter()
last_end, idle, gaps = t0, 0, collections.Counter()
for s, e, n in win:
    busy[n] += e - s; calls[n] += 1
    if s > last_end:
        idle += s - last_end
        gaps[n] += s - last_end
    last_end = max(last_end, e)
print("window %.2f ms for %d samples = %.3f ms/sample; GPU busy %.3f, idle %.3f ms/sample; %d launches/sample" % (
    (t1 - t0) / 1e6, S, (t1 - t0) / 1e6 / S, (t1 - t0 - idle) / 1e6 / S, idle / 1e6 / S, len(win) // S))
for n, v in busy.most_common(16):
    print("  %-64s %7.1f us/sample  (%.1f launches)" % (n[:64], v / 1e3 / S, calls[n] / S))
print("idle time by the kernel that ended the gap (us/sample):")
for n, v in gaps.most_common(8):
    print("  %-64s %7.1f" % (n[:64], v / 1e3 / S))
PY
