set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/beam_step_bench.py > $O/beam_new.txt 2>&1; cat $O/beam_new.txt
timeout 600 python tools/folder_bench.py > $O/folder_new.txt 2>&1; tail -1 $O/folder_new.txt
timeout 600 python tools/stream_bench.py > $O/stream_new.txt 2>&1; tail -4 $O/stream_new.txt
