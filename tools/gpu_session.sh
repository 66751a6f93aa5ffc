set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $p | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $p --output-format csv -d $O/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err; echo "pmc $tag rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/microbench/pmc_report.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES r2 $O > $O/pmc_report.txt 2>&1; tail -5 $O/pmc_report.txt
cat $O/xattn_pmc.json
find $O -name "*counter_collection.csv" -delete
