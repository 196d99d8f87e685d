cd $GRAFT_REPO_ROOT; python -m pytest tests/test_module_path.py tests/test_experiment_io.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -15
