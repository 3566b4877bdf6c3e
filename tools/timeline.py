#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [iteration marker kernel] -- one steady-state iteration of a rocprofv3 --kernel-trace as start / end offsets (us): which launches
overlap, where the gaps are.  The marker (default k_update_merged) ends an iteration; the one printed is in the middle of the trace."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_update_merged"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows), key=lambda e: e[0])
idx = [i for i, e in enumerate(ev) if mark in e[2]]
if len(idx) < 4:
    sys.exit("marker not found often enough")
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = ev[a][1]
print(f"iteration = {(ev[b][1] - t0) / 1e3:.1f} us (end of {mark} to end of the next)")
for s, e, n, q in ev[a + 1:b + 1]:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f} us)  q{q}  {n}")
