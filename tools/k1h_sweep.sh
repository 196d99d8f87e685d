# Schedule sweep of the split-half kernel (timing only).  Build the variants HERE (hipcc cross-compiles; the GPU box would
# spend its minutes compiling), then run them in ONE gpurun call - boxes differ by 2-3 %, only same-run pairs compare:
#   bash tools/k1h_sweep.sh build && gpurun -- 'bash tools/k1h_sweep.sh run'
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
VARIANTS=("base:" "stage8:-DASDF16_STAGE_KB=8" "pf2:-DASDF16_PREFETCH=2" "sched4:-DASDF16_SCHED_KB=4" "nocheck:-DASDF16_NO_RANGE_CHECK")
if [ "$1" = "build" ]; then
  for v in "${VARIANTS[@]}"; do
    n=${v%%:*}; f=${v#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialignsdf_amd/csrc '-DABL_LIST=X(0) X(16) X(4)' $f tools/k1h_ablate.hip -o tools/bin/k1h_$n &
  done
  wait
else
  for r in 1 2; do for v in "${VARIANTS[@]}"; do n=${v%%:*}; echo "== $n"; tools/bin/k1h_$n ${N:-256} | tail -3; done; done
fi
