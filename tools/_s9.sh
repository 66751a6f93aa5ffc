cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_width.py tests/test_gpu_full_size.py tests/test_gpu_c5_c2.py -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
