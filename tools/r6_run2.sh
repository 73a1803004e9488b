export TMPDIR=/tmp
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_layers.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r6b/tests.txt
PN2_LIB_SUFFIX=_c timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_layers.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r6b/tests_c.txt
bash tools/r6_pair_ab.sh r6b "" _c
P=/tmp/prof_r6b; rm -rf $P; mkdir -p $P
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P/psq -o pair -- python tools/pair_bench.py 3 --plain --only layer > gpurun_out/r6b/sq.log 2>&1
python tools/pmc_table.py gpurun_out/r6b/pair_sq_counters.csv --filter grid_ $P/psq/pair_counter_collection.csv > gpurun_out/r6b/pmc_table.txt 2>&1
cat gpurun_out/r6b/pair_sq_counters.csv | head -5
