#!/bin/bash
# tools/ab_env.sh on the semi-supervised step:  gpurun -- 'bash tools/ab_env_semi.sh NAME 0 1'
name=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $name=$v python bench.py --workload semi --steps 20 --warmup 5 --no-kernels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name=$v', d['value'], d['ms_per_step'])"
done; done
