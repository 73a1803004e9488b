cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_layers.py tests/test_train_step.py -q -m gpu -x -k "inverse or layers or train" > gpurun_out/r2g/t1.log 2>&1; tail -n 3 gpurun_out/r2g/t1.log
timeout 600 python bench.py 2>&1 | tail -n 1 > gpurun_out/r2g/bench.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r2g/bench.json'))
print(d['ms_per_step'], d['value'])
for k,v in d['kernels'].items():
    if 'grad' in k or 'inverse' in k: print(k, v)
P
