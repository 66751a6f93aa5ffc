cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/_x16.py < /dev/null
timeout 200 python tools/beam_step_bench.py --clips 6,8 --beam 5 --new-tokens 32 2>/dev/null < /dev/null
timeout 900 python -m pytest tests/test_gpu_beam.py tests/test_gpu_prefill.py tests/test_gpu_c5_c2.py tests/test_gpu_fuzz.py -m gpu -q -x < /dev/null 2>&1 | tail -3
