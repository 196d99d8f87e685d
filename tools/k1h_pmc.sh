# PMC passes over the ablation binaries (real weights unless DATA says otherwise): one rocprofv3 run per counter set and variant.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r2/pmc_${TAG:-a}; rm -rf $O; mkdir -p $O
D=${DATA:-tools/bin/k1h_nerf3.bin}
for v in ${VARIANTS:-r1base new_pf1}; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${v}_$i -- tools/bin/k1h_$v 256 $D > $O/${v}_$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*_[0-9]")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-1], {k: sum(v) / len(v) for k, v in acc.items()})
PY
