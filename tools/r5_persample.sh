cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
python tools/per_sample_times.py 64 hand 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/per_sample_times.txt
python tools/per_sample_times.py 128 both 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/per_sample_times.txt
