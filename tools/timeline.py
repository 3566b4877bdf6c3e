"""One steady-state iteration of a rocprofv3 kernel trace as a timeline: start / end of every kernel relative to the
iteration's forward sweep, per queue.  python tools/timeline.py <kernel_trace.csv> [iteration]
iteration < 0: counted from the end of the trace (bench.py ends with the ca_run fit, whose host reads every ELBO before it
queues the update half: 6 us in front of k_final_gene that ca_iterate does not have); > 0: from the start (the timed ca_iterate
regions follow the warm-up)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else -10
fw = [i for i, r in enumerate(rows) if r["Kernel_Name"].replace("void ", "").startswith(("k_fwd_cell", "k_fwd_bal"))]
i0, i1 = (fw[back - 1], fw[back]) if back < 0 else (fw[back], fw[back + 1])
t0 = int(rows[i0]["Start_Timestamp"])
print(f"iteration length {(int(rows[i1]['Start_Timestamp']) - t0) / 1000:.1f} us")
lo = i0
while lo > 0 and int(rows[lo - 1]["End_Timestamp"]) > t0 - 40000:
    lo -= 1
for r in rows[lo:i1 + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1000, (int(r["End_Timestamp"]) - t0) / 1000
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:34]
    print(f"  q{r['Queue_Id']:>2} {s:9.1f} -> {e:9.1f}  ({e - s:7.1f})  {name}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')} vgpr {r.get('VGPR_Count', '?')}")
