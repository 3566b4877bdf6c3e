"""Mean idle time between consecutive kernels of each queue, from a rocprofv3 kernel trace (second half of the run)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
last, gaps = {}, collections.defaultdict(list)
for r in rows:
    q = r["Queue_Id"]
    if q in last:
        p = last[q]
        gaps[(p["Kernel_Name"][:20], r["Kernel_Name"][:20], q)].append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1000)
    last[q] = r
for k, v in gaps.items():
    if len(v) > 20:
        v.sort()
        print("  %-20s -> %-20s q%s n=%d mean %.2f median %.2f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v), v[len(v) // 2]))
