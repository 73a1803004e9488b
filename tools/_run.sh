set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > $O/bench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --workload semi > $O/bench_semi.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --workload sunrgbd > $O/bench_sun.log 2>&1
