set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest.log
timeout 600 python tools/decode_variants.py --variants auto,dual,qkv2 > $O/variants.jsonl 2> $O/variants.err; echo "variants rc=$?"
cat $O/variants.jsonl; tail -3 $O/variants.err
timeout 900 python bench.py --cpu-full --write-crc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json; tail -5 $O/bench.err
cp profiles/cpu_baseline_full.json profiles/bench_tokens_crc.json $O/ 2>/dev/null
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_g2.json 2> $O/bench_g2.err; echo "bench2 rc=$?"
cat $O/bench_g2.json; tail -5 $O/bench_g2.err
