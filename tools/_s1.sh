set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | head -c 6000; echo; tail -5 $O/bench.err
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 > $O/stream8.json 2> $O/stream8.err; echo "stream rc=$?"; cat $O/stream8.json
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 --audio-ctx auto > $O/stream8_auto.json 2>> $O/stream8.err; cat $O/stream8_auto.json
