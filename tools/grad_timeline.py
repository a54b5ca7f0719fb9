"""One loss_and_grad call's kernels with durations and the gaps between them, from a rocprofv3 kernel trace of tools/prof_grad.py:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prof_grad.py 3 32 6 ; python tools/grad_timeline.py DIR"""
import csv, glob, os, sys
g = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(g)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last gradient of the run: from the last-but-one k_grad_reduce to the last one
red = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_grad_reduce')]
i0, i1 = red[-2] + 1, red[-1]
prev_end, tot_k = None, 0.0
for r in rows[i0:i1 + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    tot_k += (en - st) / 1e3
    print(f"{r['Kernel_Name'][:44]:46s} dur {(en - st) / 1e3:8.2f} us   gap before {gap:7.2f} us")
    prev_end = en
span = (int(rows[i1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
print(f"span {span:.1f} us, kernels {tot_k:.1f} us, gaps {span - tot_k:.1f} us")
