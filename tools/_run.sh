cd $GRAFT_REPO_ROOT
python tools/bwd_bench.py sa2_l1 sa1_l2 2>&1 | grep "^sa"
MLP_FUSED_BWD_WGS=768 python tools/bwd_bench.py sa1_l2 2>&1 | grep "^sa"
timeout 600 python bench.py 2>&1 | tail -n 1 | cut -c1-260
