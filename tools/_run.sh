cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -k "fused_backward or fused_chain" > gpurun_out/r2g/t1.log 2>&1; tail -n 3 gpurun_out/r2g/t1.log
python tools/bwd_bench.py sa2_l2 sa2_l3 2>&1 | grep "^sa" | cut -c1-230
timeout 600 python bench.py 2>&1 | tail -n 1 | cut -c1-260
