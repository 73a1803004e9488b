set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F -o pair -- python tools/pair_bench.py 5 --plain > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/W -o pair -- python tools/pair_bench.py 5 --plain > $O/w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S -o pair -- python tools/pair_bench.py 10 --plain > $O/s.log 2>&1
python tools/pair_pmc.py $O/F/pair_counter_collection.csv $O/W/pair_counter_collection.csv $O/S/pair_kernel_stats.csv $O/r2_pair_pmc.json > $O/pmc_json.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/SQ -o pair -- python tools/pair_bench.py 3 --plain --ablate > $O/sq.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
