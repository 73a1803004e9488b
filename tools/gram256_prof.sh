#!/bin/bash
# per-kernel durations + SQ counters of the 256 x 128 Gram passes (run on the GPU box):
#     bash tools/gram256_prof.sh <outdir>
O=${1:-gpurun_out/gram256_prof}; mkdir -p $O
export TMPDIR=/tmp
P=/tmp/g256prof; rm -rf $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/s -o g -- python tools/gram256_bench.py > $O/stats.log 2>&1
cp $P/s/g_kernel_stats.csv $O/gram256_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $P/c -o g -- python tools/gram256_bench.py > $O/pmc.log 2>&1
python tools/pmc_kernel.py $P/c/g_counter_collection.csv pool_gram256 $O/gram256_pmc.json > $O/pmc_kernel.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $P/d -o g -- python tools/gram256_bench.py > $O/pmc2.log 2>&1
python tools/pmc_table.py $O/gram256_pmc2.csv --filter pool_gram256 $P/d/g_counter_collection.csv > /dev/null 2>&1
