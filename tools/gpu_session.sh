# One GPU-box session that re-validates the tree and refreshes the judged artefacts (run through gpurun from the repo root):
#   full GPU test suite, smoke(), the headline bench line (+ token CRC), the rocprofv3 kernel summary of the same command,
#   and the three separate PMC passes aggregated into profiles/-shaped JSON.  Everything lands under gpurun_out/session/.
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/session; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --write-crc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
cp profiles/bench_tokens_crc.json $O/
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -3 $O/kernel_stats.csv | cut -c1-250; rm -rf $O/prof
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $p | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $p --output-format csv -d $O/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --new-tokens 8 --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err; echo "pmc $tag rc=$?"
done
python tools/microbench/pmc_report.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES r3 $O > $O/pmc_report.txt 2>&1; tail -3 $O/pmc_report.txt
find $O -name "*counter_collection.csv" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES
