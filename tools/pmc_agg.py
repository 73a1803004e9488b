import csv, sys
from collections import defaultdict
agg=defaultdict(lambda: defaultdict(lambda:[0,0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if len(sys.argv) > 2 and not any(w in k for w in sys.argv[2].split(',')): continue
    short=k.split('::')[-1][:48] if '::' in k else k[:48]
    a=agg[short][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,d in agg.items():
    print(k, {c:round(v[1]/v[0]) for c,v in d.items()})
