# Round-6 profile of bench.py on an MI355X (run through gpurun): kernel-trace stats + PMC passes of the split-half kernel (the DEFAULT:
# ordinary sweeps, every voxel) AND of the one-plane kernel (--fast), summarised into gpurun_out/r6/prof/*; copy what should be judged
# into profiles/ (r06_*).      gpurun -- 'bash tools/profile_bench_r6.sh'
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6/prof; rm -rf $O; mkdir -p $O
ARGS="--no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 6 --warmup 2 $ARGS --details $O/bench_under_kernel_trace_details.json > $O/bench_under_kernel_trace.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fast -- python bench.py --steps 8 --warmup 2 --fast $ARGS --details $O/bench_fast_under_kernel_trace_details.json > $O/bench_fast_under_kernel_trace.json 2> $O/stats_fast.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python bench.py --steps 2 --warmup 1 $ARGS --details /tmp/d.json > /dev/null 2> $O/pmc_$i.err
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcf_$i -- python bench.py --steps 2 --warmup 2 --fast $ARGS --details /tmp/d.json > /dev/null 2> $O/pmcf_$i.err
done
python3 - <<PY
import csv, glob, collections, json, hashlib, os
O = "$O"
for tag in ("stats", "stats_fast"):
    for f in glob.glob(O + "/" + tag + "/**/*kernel_stats.csv", recursive=True):
        open(O + "/kernel_" + tag + ".csv", "w").write(open(f).read())
def collect(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(O + "/" + prefix + "_[0-9]")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
out = []
for prefix, title in (("pmc", "DEFAULT: ordinary sweeps (split-half kernel on every voxel of both lattices)"), ("pmcf", "--fast: audited one-plane sweeps (one-plane kernel dominant)")):
    acc = collect(prefix)
    out.append("==== " + title)
    for k in sorted(acc):
        if "sdf_mlp" in k:
            out.append(k)
            for c in sorted(acc[k]):
                v = acc[k][c]
                out.append("  %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    for name in ("sdf_mlp_f16p1_kernel", "sdf_mlp_f16_kernel", "sdf_mlp_f16w_kernel"):
        k = [n for n in acc if n.endswith(name) or (name + "E") in n or name == n.split("::")[-1]]
        if not k or "FETCH_SIZE" not in acc[k[0]] or (name != "sdf_mlp_f16p1_kernel") != (prefix == "pmc"):
            continue
        a = acc[k[0]]
        fetch = sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"]) * 1024 * 2      # KB -> B, x2: gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md)
        write = sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"]) * 1024
        h = hashlib.sha256()
        for src in (("sdf_mlp_f16w_kernel.h",) if name == "sdf_mlp_f16w_kernel" else ()) + ("sdf_mlp_f16_kernel.h", "sdf_mlp_common.h", "sdf_layout.h"):
            h.update(open(os.path.join("alignsdf_amd", "csrc", src), "rb").read())
        short = {"sdf_mlp_f16p1_kernel": "f16p1", "sdf_mlp_f16_kernel": "f16", "sdf_mlp_f16w_kernel": "f16w"}[name]
        json.dump({"kernel": name, "grid": 256, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
                   "source_sha256": h.hexdigest(),
                   "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --steps 2; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; mean over the launches of the run (coarse and fine sweeps)"},
                  open(O + "/hbm_traffic_%s.json" % short, "w"), indent=1)
open(O + "/pmc_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out)[:7000])
PY
head -12 $O/kernel_stats.csv; head -8 $O/kernel_stats_fast.csv
