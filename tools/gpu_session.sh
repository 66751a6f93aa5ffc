set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_large_width.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --write-crc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json; tail -3 $O/bench.err
TTASR_ENC_RES_EPI=1 timeout 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_resepi.json 2> $O/bench_resepi.err; echo "bench rc=$?"
cat $O/bench_resepi.json; tail -3 $O/bench_resepi.err
cp profiles/bench_tokens_crc.json $O/
