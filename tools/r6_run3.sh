export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_layers.py -q -m gpu -x 2>&1 | tail -4
bash tools/r6_pair_ab.sh r6c ""
GRID_PROBE_TAG=_layer timeout 300 python tools/micro/grid_probe.py gpurun_out/r6c/centroid_clocks_layer.json --raw --layer > gpurun_out/r6c/probe_layer.log 2>&1; echo probe rc=$?
timeout 120 python tools/micro/grid_occupancy.py gpurun_out/r6c/occupancy_layer.json > gpurun_out/r6c/occ_layer.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r6c/occupancy_layer.json"))
for k in d:
    print(k, {q:d[k][q] for q in d[k] if q!="timeline_every_quarter_us"})
    print(d[k]["timeline_every_quarter_us"]["resident"])
c=json.load(open("gpurun_out/r6c/centroid_clocks_layer.json"))
for k in c:
    for cl,v in c[k]["classes"].items(): print(k, cl, v)
PY
