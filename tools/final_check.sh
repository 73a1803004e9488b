#!/bin/bash
# End-of-round check on the GPU box: full bench line, stress configuration, the GPU test suite, smoke.
export TMPDIR=/tmp
mkdir -p gpurun_out/r6f
timeout 900 python bench.py > gpurun_out/r6f/bench_line.json 2> gpurun_out/r6f/bench.err; echo "bench rc=$?"
timeout 600 python tools/stress_bench.py gpurun_out/r6f/r6_stress_config5.json > gpurun_out/r6f/stress.log 2>&1; echo "stress rc=$?"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python tools/gram256_soak.py 100 2>&1 | tail -1
timeout 600 python tools/pool_fwd256_soak.py 120 2>&1 | tail -2
