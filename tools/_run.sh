set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
timeout 600 python tools/stress_bench.py $O/stress.json > $O/stress.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1
