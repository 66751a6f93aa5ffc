set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -4 $O/kernel_stats.csv | cut -c1-300; rm -rf $O/prof
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $p | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $p --output-format csv -d $O/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err; echo "pmc $tag rc=$?"
done
python tools/microbench/pmc_report.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES r3 $O > $O/pmc_report.txt 2>&1; tail -5 $O/pmc_report.txt
find $O -name "*counter_collection.csv" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
ls -la $O
