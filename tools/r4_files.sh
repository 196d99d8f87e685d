# Round-4 timing of the file-producing flow (PLY export; eval mode = + ground-truth ICP per sample) under both kinds of sweep.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_icp.py tests/test_gpu_pipeline.py tests/test_experiment_io.py tests/test_gpu_chamfer.py tests/test_mesh_post.py -q 2>&1 | tail -12
(
echo "# tools/time_reconstruct_files.py 256 8 [eval]  (MI355X, 1 GPU, synthetic nerf3 decoder, PLY export on; K8 + surface sampling + ICP on the device,"
echo "# ground-truth parsing / sampling in a worker process, PLY writes on a writer thread)"
for mode in "ASDF_COARSE=box ASDF_FINE=band" "ASDF_COARSE=exact ASDF_FINE=exact"; do
  echo "== $mode"
  env $mode python tools/time_reconstruct_files.py 256 8 2>/dev/null | grep -v "^$"
  env $mode python tools/time_reconstruct_files.py 256 8 eval 2>/dev/null | grep -v "^$"
done ) | tee gpurun_out/r4/reconstruct_files_timing.txt
