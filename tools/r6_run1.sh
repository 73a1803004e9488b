export TMPDIR=/tmp
mkdir -p gpurun_out/r6a
( python tools/semi_step_branches.py 12 0 > gpurun_out/r6a/branches_rank0.txt 2>&1 & 
  python tools/semi_step_branches.py 12 1 > gpurun_out/r6a/branches_rank1.txt 2>&1 ; wait )
echo "branches done"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r6a/gputests.txt; tail -5 gpurun_out/r6a/gputests.txt
timeout 300 python -m pytest tests/test_gpu_mlp.py -q -m gpu -s -k "float64 or chain_vs_sequential or virtual_first or pregathered" 2>&1 | grep -i "float64\|passed\|failed\|bound" | tail -40 > gpurun_out/r6a/mlp_bounds.txt
timeout 600 python bench.py > gpurun_out/r6a/bench_line.json 2> gpurun_out/r6a/bench.err; echo "bench rc=$?"
