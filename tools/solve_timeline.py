import csv,glob,os,sys
g=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True),key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(g)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'][:28] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('k_build_u0')]
i0=idx[60]; i1=idx[61]
prev_end=None
for r in rows[i0:i1+1]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(st-prev_end)/1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:24]:26s} dur {(en-st)/1e3:7.2f} gap {gap:6.2f}")
    prev_end=en
print('solve period', (int(rows[i1]['Start_Timestamp'])-int(rows[i0]['Start_Timestamp']))/1e3)
