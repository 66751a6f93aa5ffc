set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/stream_bench.py > $O/stream_new.txt 2>&1; tail -1 $O/stream_new.txt | cut -c1-400
TTASR_PREFILL_TILED=1 timeout 600 python tools/stream_bench.py > $O/stream_tiled.txt 2>&1; tail -1 $O/stream_tiled.txt | cut -c1-400
timeout 300 python tools/decode_variants.py --variants auto,prefill_sot,auto,prefill_sot > $O/variants.txt 2>&1; cat $O/variants.txt
timeout 600 python tools/folder_bench.py > $O/folder_new.txt 2>&1; tail -1 $O/folder_new.txt
TTASR_PREFILL_TILED=1 timeout 600 python tools/folder_bench.py > $O/folder_tiled.txt 2>&1; tail -1 $O/folder_tiled.txt
