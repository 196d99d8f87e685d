# schedule sweep of the split-half kernel on the GPU box (timing only)
cd $GRAFT_REPO_ROOT
for cfg in "1 4" "2 4" "2 5" "2 3" "3 4" "1 5"; do
  set -- $cfg
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialignsdf_amd/csrc -DASDF16_PREFETCH=$1 -DASDF16_BARRIER_KB=$2 ${EXTRA} tools/k1h_ablate.hip -o /tmp/k1h_$1_$2 2>/dev/null && /tmp/k1h_$1_$2 ${N:-256}
done
