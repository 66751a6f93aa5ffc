cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 200 python tools/beam_step_bench.py --clips 6,8 --beam 5 --new-tokens 32 2>/dev/null < /dev/null
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 2>/dev/null < /dev/null
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 --audio-ctx auto 2>/dev/null < /dev/null
timeout 600 python -m pytest tests/test_gpu_beam.py tests/test_gpu_wide_batch.py tests/test_gpu_c5_c2.py -m gpu -q -x < /dev/null 2>&1 | tail -3
