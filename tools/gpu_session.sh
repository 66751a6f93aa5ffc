set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
timeout 600 python tools/decode_variants.py --variants auto,slab_nt,slab_sc01,slab_sc1,slab_nt_sc01,auto,slab_nt,slab_sc01,slab_sc1,slab_nt_sc01 > $O/variants.txt 2>&1; cat $O/variants.txt
