#!/bin/bash
# which kernels surround torch's elementwise launches in one step?  (run on the GPU box)
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/trace_adds" -o t -- python3 "$root/bench.py" --steps 1 --warmup 2 --no-cpu-baseline --no-roofline > "$root/gpurun_out/trace_adds.log" 2>&1
cd $root
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_adds/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last third = the timed step (approximately): use the final 1700 launches
rows=rows[-1700:]
def short(n): return n.replace('void ','').split('(')[0][:70]
from collections import Counter
c=Counter()
for i,r in enumerate(rows):
    n=r['Kernel_Name']
    if 'at::native' in n:
        key=(short(n), r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), short(rows[i-1]['Kernel_Name']), short(rows[i+1]['Kernel_Name']) if i+1<len(rows) else '')
        c[key]+=1
for k,v in c.most_common(40): print(v,k)
PY
