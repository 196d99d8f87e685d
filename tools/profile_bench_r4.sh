# Round-4 profile of bench.py on an MI355X (run through gpurun): kernel-trace stats + PMC passes of the dominant kernel of the
# default sweeps (sdf_mlp_f16p1_kernel), summarised into gpurun_out/r4/prof/*; copy what should be judged into profiles/ (r04_*).
#   gpurun -- 'bash tools/profile_bench_r4.sh'
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4/prof; rm -rf $O; mkdir -p $O
ARGS="--no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 8 --warmup 2 $ARGS > $O/bench_under_kernel_trace.json 2> $O/stats.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $O/pmc_$i.err
done
python3 - <<PY
import csv, glob, collections, json, hashlib, os
O = "$O"
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    open(O + "/kernel_stats.csv", "w").write(open(f).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(O + "/pmc_[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k in sorted(acc):
    if "sdf_mlp" in k or "mc_" in k or "collect" in k or "neg_bbox" in k or "band_" in k or "audit" in k or "fold" in k:
        lines.append(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append("  %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
open(O + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
k = [n for n in acc if "sdf_mlp_f16p1_kernel" in n]
if k:
    a = acc[k[0]]
    fetch = sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"]) * 1024 * 2      # KB -> B, x2: gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md)
    write = sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"]) * 1024
    h = hashlib.sha256()
    for name in ("sdf_mlp_f16_kernel.h", "sdf_mlp_common.h", "sdf_layout.h"):
        h.update(open(os.path.join("alignsdf_amd", "csrc", name), "rb").read())
    busy = sum(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(a["SQ_VALU_MFMA_BUSY_CYCLES"]) if a.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
    json.dump({"kernel": "sdf_mlp_f16p1_kernel", "grid": 256, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch,
               "write_bytes": write, "source_sha256": h.hexdigest(),
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --steps 2; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; mean over the launches of the run (coarse and fine sweeps)"},
              open(O + "/hbm_traffic_f16p1.json", "w"), indent=1)
    print(open(O + "/hbm_traffic_f16p1.json").read())
print("\n".join(l for l in lines if "f16p1" in l or l.startswith("  ")) [:6000])
PY
head -24 $O/kernel_stats.csv
