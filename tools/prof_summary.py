"""Summarise a rocprofv3 kernel_stats.csv (engine kernels only) -> stdout / profiles/*.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
hdr = "# " + (sys.argv[3] if len(sys.argv) > 3 else "") + "\nkernel,calls,avg_us,total_ms\n"
if out: out.write(hdr)
for r in rows:
    n = r["Name"]
    if "at::" in n or "rocclr" in n: continue
    n = n.split("(")[0].replace("void ", "")
    line = f"{n},{r['Calls']},{float(r['AverageNs'])/1e3:.1f},{float(r['TotalDurationNs'])/1e6:.2f}"
    print(line)
    if out: out.write(line + "\n")
