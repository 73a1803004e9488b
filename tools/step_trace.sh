#!/bin/bash
# The kernel timeline of ONE train step (everything between two adam_update_kernel launches, both
# streams, profiler-serialised): start offset, duration, kernel -> gpurun_out/step_trace_one.txt
export TMPDIR=/tmp
P=/tmp/sp; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P/step -o step -- python bench.py --steps 6 --warmup 3 --no-kernels --no-cpu-baseline --no-workloads > $P/log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/sp/step/step_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last adam_update_kernel and the one before it: one step between them on the main stream
ad=[i for i,r in enumerate(rows) if 'adam_update' in r['Kernel_Name']]
lo,hi=ad[-2],ad[-1]
t0=int(rows[lo]['End_Timestamp'])
out=[]
for r in rows[lo+1:hi+1]:
    n=r['Kernel_Name']
    if n.startswith('void '): n=n[5:]
    n=n.replace('(anonymous namespace)::','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
    out.append("%8.1f %7.1f q%s %s"%((int(r['Start_Timestamp'])-t0)/1000,d,r.get('Queue_Id','?'),n[:90]))
open('gpurun_out/step_trace_one.txt','w').write("\n".join(out))
print(len(out))
PY
