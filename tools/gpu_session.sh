set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_c5_c2.py tests/test_longform_golden.py tests/test_gpu_facade.py tests/test_gpu_weights_and_launch.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
