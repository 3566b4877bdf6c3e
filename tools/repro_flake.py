"""Repeats one short fit on fresh engines and counts the distinct results (a deterministic engine gives ONE):  python tools/repro_flake.py [reps] [variant_off,...]"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
voff = tuple(v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if v)
shape = dict(N=40_100, G=1100, C=8, K=1)
case = make_case(seed=77, **shape)
rng = np.random.default_rng(3)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = case["Y"].shape[1]
epss = np.stack([eps_for(1, G, 300 + i) for i in range(10)])
seen = Counter()
per_iter = {}
for r in range(reps):
    eng = HipEngine(**case, variant_off=voff)
    try:
        eng.gamma_init(eps_for(1, G, 0))
        vals = []
        for k in range(5):                       # one iteration per call: where does a deviating run part ways?
            vals.append(eng.iterate(1, epss[2 * k:2 * k + 2]))
        last = eng.iterate(5, epss)
    finally:
        eng.close()
    seen[(tuple(vals), last)] += 1
for (vals, last), n in seen.items():
    print(n, "x", [f"{v:.6f}" for v in vals], f"then iterate(5): {last:.6f}")
print("distinct results:", len(seen), "variants off:", voff)
