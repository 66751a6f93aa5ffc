set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --write-crc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json; echo
cp profiles/bench_tokens_crc.json $O/
timeout 600 python bench.py --compute f16 --write-crc --no-cpu-baseline > $O/bench_f16.json 2>> $O/bench.err; echo "bench f16 rc=$?"; head -c 300 $O/bench_f16.json; echo
cp profiles/bench_tokens_crc.json $O/
