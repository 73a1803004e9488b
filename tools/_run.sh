cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
timeout 300 python tools/pair_bench.py 20 --sweep --json $O/pair.json > $O/pair.log 2>&1
