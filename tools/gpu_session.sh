set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
./tools/microbench/bin/skinny_bench2 > $O/skinny.txt 2>&1; cat $O/skinny.txt
./tools/microbench/bin/layer_bench2 -1 > $O/layer.txt 2>&1; cat $O/layer.txt
timeout 600 python tools/decode_variants.py --variants auto,w_nt > $O/variants.jsonl 2> $O/variants.err; echo "variants rc=$?"
cat $O/variants.jsonl; tail -3 $O/variants.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.log
