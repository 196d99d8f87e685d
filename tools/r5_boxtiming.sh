cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
python tools/box_sweep_timing.py 64 hand 2>&1 | tee gpurun_out/r5/box_sweep_timing.txt
python tools/box_sweep_timing.py 128 both 2>&1 | tee -a gpurun_out/r5/box_sweep_timing.txt
