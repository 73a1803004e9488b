#!/bin/bash
# SQ counters of the small-GEMM kernels (run on the GPU box):   bash tools/small_gemm_prof.sh <outdir>
O=${1:-gpurun_out/small_gemm_prof}; mkdir -p $O
export TMPDIR=/tmp
P=/tmp/sgprof; rm -rf $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/s -o g -- python tools/small_gemm_bench.py > $O/stats.log 2>&1
cp $P/s/g_kernel_stats.csv $O/small_gemm_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $P/c -o g -- python tools/small_gemm_bench.py > $O/pmc.log 2>&1
python tools/pmc_kernel.py $P/c/g_counter_collection.csv gemm_ $O/small_gemm_pmc.json > $O/pmc_kernel.txt 2>&1
