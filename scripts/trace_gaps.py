"""Gaps on the caller's queue in one frame of a rocprofv3 --kernel-trace run (csv): python scripts/trace_gaps.py <dir>"""
import csv,glob,re,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'gsr::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
frames=[]; cur=[]
for r in rows:
    if 'preprocess_kernel' in r['Kernel_Name'] and cur: frames.append(cur); cur=[]
    cur.append(r)
frames.append(cur)
fr=frames[len(frames)//2]
t0=int(fr[0]['Start_Timestamp']); q0=fr[0]['Queue_Id']; prev=None
for r in fr:
    if r['Queue_Id']!=q0: continue
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    m=re.search(r'::(\w+?)(?:<[^(]*>)?\(', r['Kernel_Name']); name=m.group(1) if m else '?'
    if prev is not None and s-prev>1500: print(f"gap {(s-prev)/1e3:5.1f} us before {name} (at {s/1e3:.1f})")
    prev=e
print("frame kernels span", (max(int(r['End_Timestamp']) for r in fr)-t0)/1e3)
