cd $GRAFT_REPO_ROOT
B=tools/microbench/bin/layer_bench3
for nt in 1 0; do
  timeout 120 $B 0 256 $nt
  for mode in 5 6 4 3 1; do timeout 120 $B $mode 256 $nt | tail -1; done
done
timeout 120 $B 3 64 1; timeout 120 $B 3 1024 1
