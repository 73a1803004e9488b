set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_layers.py tests/test_train_step.py tests/test_semi_step.py -m gpu -q > $O/pytest_mlp.log 2>&1; echo "rc=$?" >> $O/pytest_mlp.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > $O/bench.log 2>&1
MLP_GEMM_EPILOGUE_STATS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > $O/bench_nostats.log 2>&1
