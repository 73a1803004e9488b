cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
rm -rf gpurun_out/r2h/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h/prof -o step -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2h/bench_prof.log 2>&1
