"""How far ahead of the GPU is the host?  For every launch of a kernel (default k_final_gene): the idle time in front of it on the
GPU and the time between the end of its hipLaunchKernel call and its start on the GPU.
python tools/host_ahead.py <dir with t_kernel_trace.csv + t_hip_api_trace.csv> [kernel-name-prefix]"""
import csv
import os
import sys

d = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "k_final_gene"
k = list(csv.DictReader(open(os.path.join(d, "t_kernel_trace.csv"))))
a = list(csv.DictReader(open(os.path.join(d, "t_hip_api_trace.csv"))))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
la = {r["Correlation_Id"]: r for r in a if "Launch" in r["Function"]}
prev, rows = None, []
for r in k:
    name = r["Kernel_Name"].replace("void ", "")
    if name.startswith(pre) and prev is not None and r["Correlation_Id"] in la:
        l = la[r["Correlation_Id"]]
        rows.append(((int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3,
                     (int(r["Start_Timestamp"]) - int(l["End_Timestamp"])) / 1e3,
                     (int(prev["End_Timestamp"]) - int(prev["Start_Timestamp"])) / 1e3, prev["Kernel_Name"].replace("void ", "")[:16]))
    prev = r
print(f"{len(rows)} launches of {pre}*: gap in front (us) | launch call returned this long before the GPU start (us) | predecessor (us)")
for i, (g, h, pd, pn) in enumerate(rows):
    if i < 60 or i % 10 == 0:
        print(f"{i:4d}  gap {g:7.2f}   host ahead {h:8.1f}   after {pn} {pd:.1f}")
