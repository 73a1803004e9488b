export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_layers.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do
for c in U R step; do
  for flag in "" "--no-plan"; do
  timeout 120 python tools/pair_bench.py 40 --only layer --cloud $c $flag 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flag] cloud %s rep $rep: layer %.2f us' % (d['cloud'], d['layer_us']))"
  done
done
done | tee gpurun_out/r6c/ab_plan.txt
P=/tmp/prof_r6c; rm -rf $P; mkdir -p $P
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P/psq -o pair -- python tools/pair_bench.py 3 --plain --only layer > gpurun_out/r6c/sq.log 2>&1
python tools/pmc_table.py gpurun_out/r6c/pair_sq_counters.csv --filter grid_ $P/psq/pair_counter_collection.csv > gpurun_out/r6c/pmc_table.txt 2>&1
cat gpurun_out/r6c/pair_sq_counters.csv | head -5
