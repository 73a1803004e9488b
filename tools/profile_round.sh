#!/bin/bash
# Every rocprofv3 summary of a round, at HEAD (run on the GPU box from the repo root):
#     bash tools/profile_round.sh r4
# writes gpurun_out/<tag>_profiles/*, to be copied into profiles/.  Counter passes are separate
# runs (MI355X_MICROARCH.md); every pass is bounded by `timeout`.
TAG=${1:-r4}
export TMPDIR=/tmp
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_profiles; mkdir -p $O
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P
run() { name=$1; shift; timeout 240 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
# -- the step, timed steps only
run step rocprofv3 --kernel-trace --stats --output-format csv -d $P/step -o step -- python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline --no-workloads
python tools/step_breakdown.py $P/step/step_kernel_trace.csv 20 400 $O/${TAG}_train_step_timed_summary.csv > $O/step_breakdown.txt 2>&1
run semi rocprofv3 --kernel-trace --stats --output-format csv -d $P/semi -o step -- python bench.py --workload semi --steps 20 --warmup 5 --no-kernels --no-cpu-baseline
python tools/step_breakdown.py $P/semi/step_kernel_trace.csv 20 400 $O/${TAG}_semi_step_timed_summary.csv > $O/semi_breakdown.txt 2>&1
# -- the north-star pair: the `layer` form ALONE (one form per kernel name in every stats / counter file)
run pair_stats rocprofv3 --kernel-trace --stats --output-format csv -d $P/pair -o pair -- python tools/pair_bench.py 10 --plain --only layer
cp $P/pair/pair_kernel_stats.csv $O/${TAG}_pair_kernel_stats.csv
cp $P/pair/pair_kernel_stats.csv $O/${TAG}_pair_kernel_stats_U.csv
# -- the same on cloud R and on the timed step's own batch (bench.pair_cloud)
for c in R step; do
  run pair_stats_$c rocprofv3 --kernel-trace --stats --output-format csv -d $P/pair_$c -o pair -- python tools/pair_bench.py 10 --plain --only layer --cloud $c
  cp $P/pair_$c/pair_kernel_stats.csv $O/${TAG}_pair_kernel_stats_$c.csv
done
timeout 120 python tools/micro/grid_probe.py $O/${TAG}_pair_centroid_clocks.json --layer > /dev/null 2> $O/grid_probe.log; echo "grid_probe rc=$?"
run pair_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pf -o pair -- python tools/pair_bench.py 5 --plain --only layer
run pair_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pw -o pair -- python tools/pair_bench.py 5 --plain --only layer
python tools/pair_pmc.py $P/pf/pair_counter_collection.csv $P/pw/pair_counter_collection.csv $P/pair/pair_kernel_stats.csv $O/${TAG}_pair_pmc.json > $O/pair_pmc.txt 2>&1
run pair_sq rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P/psq -o pair -- python tools/pair_bench.py 3 --plain --only layer
run pair_mem rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $P/pmem -o pair -- python tools/pair_bench.py 3 --plain --only layer
python tools/pmc_table.py $O/${TAG}_pair_sq_counters.csv --filter grid_ $P/psq/pair_counter_collection.csv $P/pmem/pair_counter_collection.csv > /dev/null 2>&1
# -- VALU-bound operators
run ops_valu rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $P/ops -o ops -- python tools/op_bench.py 3
python tools/valu_util.py $P/ops/ops_counter_collection.csv $O/${TAG}_ops_valu_util.json > $O/ops_valu.txt 2>&1
# -- furthest point sampling: product timings, round clocks
timeout 120 python tools/micro/fps_time.py > $O/${TAG}_fps_time.json 2> $O/fps_time.log; echo "fps_time rc=$?"
# -- shared-MLP GEMMs
run gemm_util rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $P/gemm -o g -- python tools/gemm_bench.py
python tools/mfma_util.py $P/gemm/g_counter_collection.csv $O/${TAG}_gemm_mfma_util.csv > /dev/null 2>&1
run bwd_util rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $P/bwd -o g -- python tools/bwd_bench.py
python tools/mfma_util.py $P/bwd/g_counter_collection.csv $O/${TAG}_bwd_mfma_util.csv > /dev/null 2>&1
# -- SA1's chained forward and Gram backward (round 5)
timeout 120 python tools/chain_bench.py $O/${TAG}_chain_bench.json > $O/chain_bench.log 2>&1; echo "chain_bench rc=$?"
timeout 120 python tools/gram_bench.py $O/${TAG}_gram_bench.json > $O/gram_bench.log 2>&1; echo "gram_bench rc=$?"
run chain_pmc rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $P/chain -o p -- python tools/chain_bench.py
python tools/pmc_kernel.py $P/chain/p_counter_collection.csv chain_lin4 $O/${TAG}_chain_pmc.json > /dev/null 2>&1
run gram_pmc rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $P/gram -o p -- python tools/gram_bench.py
python tools/pmc_kernel.py $P/gram/p_counter_collection.csv "" $O/${TAG}_gram_pmc.json > /dev/null 2>&1
# -- SA2's pooled 128 -> 256 layer without its raw output (round 6): both passes, durations + SQ counters
timeout 300 bash tools/gram256_prof.sh $O/gram256 > /dev/null 2>&1; echo "gram256_prof rc=$?"
cp $O/gram256/gram256_kernel_stats.csv $O/${TAG}_gram256_kernel_stats.csv 2>/dev/null
cp $O/gram256/gram256_pmc.json $O/${TAG}_gram256_pmc.json 2>/dev/null
timeout 120 python tools/gram256_bench.py $O/${TAG}_gram256_bench.json > $O/gram256_bench.log 2>&1; echo "gram256_bench rc=$?"
# -- the same layer's forward as a persistent T-form kernel (round 6): durations + SQ / LDS counters, micro benchmarks
timeout 400 bash tools/pool_fwd256_prof.sh $O/pool_fwd256 > /dev/null 2>&1; echo "pool_fwd256_prof rc=$?"
cp $O/pool_fwd256/pool_fwd256_kernel_stats.csv $O/${TAG}_pool_fwd256_kernel_stats.csv 2>/dev/null
cp $O/pool_fwd256/pool_fwd256_pmc.json $O/${TAG}_pool_fwd256_pmc.json 2>/dev/null
cp $O/pool_fwd256/pool_fwd256_pmc2.csv $O/${TAG}_pool_fwd256_lds_counters.csv 2>/dev/null
grep "^b \|us per\|stored y3" $O/pool_fwd256/stats.log > $O/${TAG}_pool_fwd256_check.txt 2>/dev/null
timeout 200 python tools/micro/mfma_bf16_peak.py $O/${TAG}_mfma_bf16_peak.json > /dev/null 2>&1; echo "mfma_bf16_peak rc=$?"
timeout 100 python tools/micro/lds_read_rate.py > $O/${TAG}_lds_read_rate.json 2> /dev/null; echo "lds_read_rate rc=$?"
ls -la $O | head -60
