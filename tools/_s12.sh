cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s12; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bp -- python3 tools/beam_step_bench.py --clips 8 --beam 5 --new-tokens 32 > $O/bp.out 2> $O/bp.err < /dev/null
echo "rc=$?"
f=$(find $O/bp -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $O/beam_kernel_stats.csv; head -20 $f | cut -c1-220; fi
rm -rf $O/bp
