set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2final
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 600 python bench.py --workload semi --no-cpu-baseline --no-kernels > $O/bench_semi.log 2>&1
timeout 600 python bench.py --workload sunrgbd --no-cpu-baseline --no-kernels > $O/bench_sun.log 2>&1
timeout 600 python tools/bwd_bench.py --json $O/bwd_fused.json > $O/bwd.log 2>&1
timeout 600 python tools/pair_bench.py 20 --sweep --json $O/pair.json > $O/pair.log 2>&1
timeout 600 python tools/stress_bench.py > $O/stress.log 2>&1
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o step -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > $O/prof_step.log 2>&1
rm -rf $O/prof_semi; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_semi -o step -- python bench.py --workload semi --steps 10 --warmup 4 --no-cpu-baseline --no-kernels > $O/prof_semi.log 2>&1
rm -rf $O/prof_pair; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pair -o pair -- python tools/pair_bench.py 20 > $O/prof_pair.log 2>&1
ls $O
