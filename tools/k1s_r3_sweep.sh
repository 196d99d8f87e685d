# Round-3 schedule sweep of the one-plane kernel (two point groups per wave): variants built locally (tools/bin/k1s_*), run
# interleaved in ONE gpurun call.   VARIANTS="base pk w4" bash tools/k1s_r3_sweep.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
for r in 1 ${ROUNDS:-2}; do
  for v in ${VARIANTS:-base pk w4 pf3 tail all all3}; do
    echo "== $v"; tools/bin/k1s_$v ${N:-256} ${DATA:-tools/bin/k1h_nerf3.bin}
  done
done 2>&1 | tee gpurun_out/r3/k1s_sweep_${TAG:-a}.txt
