set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_train_step.py tests/test_semi_step.py tests/test_layers.py -q -m gpu -x > gpurun_out/r2g/t2.log 2>&1; echo rc=$? >> gpurun_out/r2g/t2.log
timeout 600 python bench.py > gpurun_out/r2g/bench.log 2>&1
MLP_FUSED_BACKWARD=0 timeout 600 python bench.py > gpurun_out/r2g/bench_nofuse.log 2>&1
