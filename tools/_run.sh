cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_train_step.py tests/test_layers.py -q -m gpu -x -k "pool or fused_chain or train or layers" > gpurun_out/r2g/t1.log 2>&1; tail -n 3 gpurun_out/r2g/t1.log
timeout 600 python bench.py 2>&1 | tail -n 1 | cut -c1-260
MLP_GEMM_EPILOGUE_POOL=0 timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -n 1 | cut -c1-260
