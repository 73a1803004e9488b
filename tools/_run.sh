set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
timeout 300 python tools/pair_bench.py 20 --sweep --json $O/pair.json > $O/pair.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1
