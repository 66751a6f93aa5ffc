cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s13; mkdir -p $O
for v in "--clips 6" "--clips 8 --option vocab_persistent=0" "--clips 8 --option graph=0"; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bp -- python3 tools/beam_step_bench.py $v --beam 5 --new-tokens 32 > $O/bp.out 2> $O/bp.err < /dev/null
  echo "variant [$v] rc=$?"; cat $O/bp.out | head -2
  f=$(find $O/bp -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f "$O/stats_$(echo $v | tr ' =' '__').csv"; fi
  rm -rf $O/bp
done
ls $O
