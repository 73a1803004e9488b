#!/bin/bash
# One rocprofv3 kernel summary of the timed train steps (gpurun_out/stepprof_summary.csv) and two
# plain bench lines, in under a minute of box time:   gpurun -- 'bash tools/stepprof.sh'
export TMPDIR=/tmp
P=/tmp/sp; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/step -o step -- python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline --no-workloads > $P/log 2>&1
grep '"metric"' $P/log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('profiled', d['ms_per_step'])"
python tools/step_breakdown.py $P/step/step_kernel_trace.csv 20 400 gpurun_out/stepprof_summary.csv > gpurun_out/stepprof_breakdown.txt 2>&1
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-kernels --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_no_prefetch'])"; done
