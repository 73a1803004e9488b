cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_train_step.py -q -m gpu -x -k "fused_backward or fused_chain or train" > gpurun_out/r2g/t1.log 2>&1; tail -n 3 gpurun_out/r2g/t1.log
timeout 600 python bench.py 2>&1 | tail -n 1 | cut -c1-260
