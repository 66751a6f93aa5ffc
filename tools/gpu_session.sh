set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --write-crc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json | head -c 900; tail -3 $O/bench.err
cp profiles/bench_tokens_crc.json $O/
