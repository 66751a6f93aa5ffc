cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_beam.py tests/test_gpu_c5_c2.py tests/test_gpu_align.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/decode_variants.py --variants auto,prefill_sot,auto,prefill_sot > $O/variants.jsonl 2> $O/variants.err; cat $O/variants.jsonl; tail -3 $O/variants.err
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 2>/dev/null
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 2>/dev/null
timeout 300 python tools/prefill_bench.py 2>/dev/null
