#!/bin/bash
# per-kernel durations + SQ counters of the no-store pooled forward (run on the GPU box):
#     bash tools/pool_fwd256_prof.sh <outdir>
O=${1:-gpurun_out/pool_fwd256_prof}; mkdir -p $O
export TMPDIR=/tmp
P=/tmp/pf256prof; rm -rf $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/s -o g -- python tools/pool_fwd256_check.py > $O/stats.log 2>&1
cp $P/s/g_kernel_stats.csv $O/pool_fwd256_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $P/c -o g -- python tools/pool_fwd256_check.py > $O/pmc.log 2>&1
python tools/pmc_kernel.py $P/c/g_counter_collection.csv pool_fwd256 $O/pool_fwd256_pmc.json > $O/pmc_kernel.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $P/d -o g -- python tools/pool_fwd256_check.py > $O/pmc2.log 2>&1
python tools/pmc_table.py $O/pool_fwd256_pmc2.csv --filter pool_fwd256 $P/d/g_counter_collection.csv > /dev/null 2>&1
