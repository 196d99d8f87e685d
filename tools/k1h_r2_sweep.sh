# Round-2 schedule sweep of the split-half kernel: variants built locally (tools/bin/k1h_*), run interleaved in ONE gpurun call.
#   VARIANTS="r1base new_pf1" DATA="tools/bin/k1h_nerf3.bin zero small" bash tools/k1h_r2_sweep.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2
for r in 1 2 ${ROUNDS:-3}; do
  for d in ${DATA:-tools/bin/k1h_nerf3.bin}; do
    for v in ${VARIANTS:-r1base new_pf1}; do
      echo "== $v $d"; tools/bin/k1h_$v ${N:-256} $d | grep ABL
    done
  done
done 2>&1 | tee gpurun_out/r2/k1h_sweep_${TAG:-a}.txt
