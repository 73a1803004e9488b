#!/bin/bash
# A/B of query-kernel variants inside ONE gpurun call: bash tools/r6_pair_ab.sh OUT "" _c ...
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for sfx in "$@"; do
  for c in U R step; do
    PN2_LIB_SUFFIX=$sfx timeout 120 python tools/pair_bench.py 40 --only layer --cloud $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib[%s] cloud %s rep $rep: layer %.2f us' % ('$sfx', d['cloud'], d['layer_us']))"
  done
done
done | tee $O/ab.txt
