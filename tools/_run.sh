set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o step -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernels > $O/bench_prof.log 2>&1
python tools/prof_summary.py $O/prof/step_kernel_trace.csv fps_bucket_kernel 5 $O/step_summary.csv > /dev/null 2>&1
rm -rf $O/prof
