set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "inverse" > gpurun_out/r2g/t1.log 2>&1; echo rc=$? >> gpurun_out/r2g/t1.log
timeout 900 python -m pytest tests/test_train_step.py tests/test_semi_step.py tests/test_layers.py tests/test_ddp.py -q -m gpu -x > gpurun_out/r2g/t2.log 2>&1; echo rc=$? >> gpurun_out/r2g/t2.log
timeout 600 python bench.py > gpurun_out/r2g/bench.log 2>&1
