#!/usr/bin/env python3
"""tools/side_time.py N G C [K [P [S [steps]]]] [key=value ...] -- bench.py's side_config() for one shape (one JSON line): the time per iteration of a shape other
than the headline's on the driver's kind of clock, then the time per kernel class.  off=name,name / on=name,name: engine variants; tune.key=value: tuning keys.
   python tools/side_time.py 100000 5000 8 2 1"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

a = [int(x) for x in sys.argv[1:] if "=" not in x]
N, G, C = a[0], a[1], a[2]
K = a[3] if len(a) > 3 else 1
P = a[4] if len(a) > 4 else 0
S = a[5] if len(a) > 5 else 1
steps = a[6] if len(a) > 6 else 40
kw = {}
for x in sys.argv[1:]:
    if x.startswith("off="):
        kw["variant_off"] = tuple(x[4:].split(","))
    elif x.startswith("on="):
        kw["variant_on"] = tuple(x[3:].split(","))
    elif x.startswith("tune."):
        k, v = x[5:].split("=")
        kw.setdefault("tune", {})[k] = int(v)
r = bench.side_config("side", N=N, G=G, C=C, K=K, P=P, S=S, steps=steps, regions=3, kernel_classes=True, engine_kw=kw)
print(json.dumps({k: r[k] for k in ("workload", "us_per_iter", "frac_of_roof", "roof_is", "kernel_class_us_per_iter", "fwd_mfma", "bwd_mfma", "fwd_block_cells")}))
