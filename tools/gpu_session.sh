set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O
for r in 30 60; do timeout 600 python tools/folder_bench.py large-v3 $r > $O/folder_$r.txt 2>&1; tail -1 $O/folder_$r.txt; done
