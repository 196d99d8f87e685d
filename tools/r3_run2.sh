cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_default_sweeps.py::test_default_meshes_are_the_ordinary_sweeps_meshes_all_64_samples_n256[nerf3]" > gpurun_out/r3/t_all2.log 2>&1; echo "all rc=$?"
tail -n 40 gpurun_out/r3/t_all2.log
