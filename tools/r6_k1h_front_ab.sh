#!/bin/bash
# Round 6: K1h, how many layer-0 tiles are computed in FRONT of layer 1 (ASDF16_L0_FRONT; the others ride one per K-block in layer 1's
# first stage).  tools/bin/k1h_front<F> = tools/k1h_ablate.hip with -DASDF16_L0_FRONT=<F> -DABL_LIST="X(0)"; interleaved, one box,
# the product's weights, status record and box fold as in the product.
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
export K1H_STATUS=1
for r in 1 2 3 ${ROUNDS:-4}; do
  for v in ${VARIANTS:-front8 front4 front2 front1}; do
    echo "== $v round $r"; timeout 120 tools/bin/k1h_$v 256 tools/bin/k1h_nerf3.bin
  done
done 2>&1 | tee $O/k1h_front_${TAG:-a}.txt | grep -E "^==|ABL +0|fault"
