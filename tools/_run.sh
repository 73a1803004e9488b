set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "ballquery or query_and_group or north_star or group_vs" > $O/pytest_pair.log 2>&1; echo "rc=$?" >> $O/pytest_pair.log
PN2_GRID_CPW=2 timeout 900 python -m pytest tests -m gpu -x -q -k "ballquery or query_and_group or north_star" > $O/pytest_pair_cpw2.log 2>&1; echo "rc=$?" >> $O/pytest_pair_cpw2.log
timeout 300 python tools/pair_bench.py 20 --sweep --json $O/pair.json > $O/pair.log 2>&1
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o pair -- python tools/pair_bench.py 10 --plain > $O/prof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc -o pair -- python tools/pair_bench.py 3 --plain --ablate > $O/pmc.log 2>&1
