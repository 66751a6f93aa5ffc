set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for rep in 1 2; do
timeout 300 python tools/decode_variants.py --variants auto > $O/new_$rep.txt 2>&1; cat $O/new_$rep.txt
timeout 300 python scratch_ab/old/tools/decode_variants.py --variants auto > $O/old_$rep.txt 2>&1; cat $O/old_$rep.txt
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
grep -i "select" $O/kernel_stats.csv | cut -c1-40,100-200
