export TMPDIR=/tmp
O=${1:-r6d}
mkdir -p gpurun_out/$O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_layers.py tests/test_sampling_chain.py -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/$O/tests.txt
bash tools/r6_pair_ab.sh $O ""
GRID_PROBE_TAG=_layer timeout 300 python tools/micro/grid_probe.py gpurun_out/$O/centroid_clocks_layer.json --raw --layer > gpurun_out/$O/probe_layer.log 2>&1; echo probe rc=$?
python - <<PY
import json
c=json.load(open("gpurun_out/$O/centroid_clocks_layer.json"))
for k in c:
    for cl,v in c[k]["classes"].items(): print(k, cl, v)
PY
