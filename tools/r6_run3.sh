export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_layers.py -q -m gpu -x 2>&1 | tail -2
PN2_LIB_SUFFIX=_p2 timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -2
bash tools/r6_pair_ab.sh r6c "" _p2 _p3 _p4
for v in "base:" "p2:-DGRID_CPW=2"; do
  tag=${v%%:*}; fl=${v#*:}
  GRID_PROBE_TAG=_$tag GRID_PROBE_FLAGS="$fl" timeout 300 python tools/micro/grid_probe.py --raw gpurun_out/r6c/centroid_clocks_$tag.json > gpurun_out/r6c/probe_$tag.log 2>&1; echo probe rc=$?
  timeout 120 python tools/micro/grid_occupancy.py gpurun_out/r6c/occupancy_$tag.json > gpurun_out/r6c/occ_$tag.log 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/r6c/occupancy_$tag.json"))["U"]
print("$tag", {k:d[k] for k in d if k!="timeline_every_quarter_us"})
print(d["timeline_every_quarter_us"]["resident"])
PY
done
