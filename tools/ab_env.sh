#!/bin/bash
# A/B of one build switch on ONE box (box-to-box spread of the step is 2-3 %): the bench twice per
# value, interleaved.   gpurun -- 'bash tools/ab_env.sh PN2_PREGATHER 0 1'
name=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $name=$v python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name=$v', d['value'], d['ms_per_step'])"
done; done
