# Round-5 timing of the file-producing flows (PLY export; eval mode = + K8, device surface sampling, ICP per hand mesh) under both kinds
# of sweep, the kernel trace of the eval-mode flow (GPU idle per sample) and the small-lattice traces.   gpurun -- 'bash tools/r5_files.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
(
echo "# tools/time_reconstruct_files.py 256 8 [eval]  (MI355X, 1 GPU, synthetic nerf3 decoder, PLY export on; K8 + surface sampling + ICP on the device,"
echo "# ground-truth parsing / sampling in a worker process, PLY writes on a writer thread; round 5: samples enqueued in one go)"
for mode in "ASDF_COARSE=box ASDF_FINE=band" "ASDF_COARSE=box ASDF_FINE=band ASDF_SPECULATE=0" "ASDF_COARSE=exact ASDF_FINE=exact"; do
  echo "== $mode"
  env $mode python tools/time_reconstruct_files.py 256 8 2>/dev/null | grep -v "^$"
  env $mode python tools/time_reconstruct_files.py 256 8 eval 2>/dev/null | grep -v "^$"
done ) | tee gpurun_out/r5/reconstruct_files_timing.txt
R=r5 bash tools/trace_eval_flow.sh > /dev/null 2>&1; cat gpurun_out/r5/trace_eval/summary.txt
( echo "# tools/trace_small_lattice.sh on one MI355X (rocprofv3 --kernel-trace over bench.py), round 5: samples enqueued in one go"
  echo "== hand-only, N = 64 (configs[0])"; R=r5 bash tools/trace_small_lattice.sh 64 hand 64 2>/dev/null
  echo; echo "== hand + object, N = 128 (configs[1])"; R=r5 bash tools/trace_small_lattice.sh 128 both 32 2>/dev/null ) | tee gpurun_out/r5/small_lattice_traces.txt
