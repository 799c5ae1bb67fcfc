#!/usr/bin/env python
"""Per (kernel, grid size) launch counts and mean durations of the C8 kernels from a rocprofv3 --kernel-trace --output-format csv
run in /tmp/prof_kt (grid size tells the batch of a launch apart: 1024 workgroups = 256 images for c8_conv3x3_kernel)."""
import csv,sys,glob,collections
f=glob.glob('/tmp/prof_kt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
acc=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'c8' in n and ('conv3x3' in n or 'wgrad_kernel' in n or 'c8s2' in n):
        g=int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size',0))
        acc[(n[:46],g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,v in sorted(acc.items()):
    print('%-48s grid %8d calls %3d avg %.3f ms' % (k[0],k[1],len(v),sum(v)/len(v)))
