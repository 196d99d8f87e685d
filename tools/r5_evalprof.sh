cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
ASDF_TIMING_PROFILE=1 python tools/time_reconstruct_files.py 256 24 eval > gpurun_out/r5/eval_host_profile.txt 2>&1
head -120 gpurun_out/r5/eval_host_profile.txt | cut -c1-180
