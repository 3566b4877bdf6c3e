"""Under a co-tenant (tools/corun.py): gamma init, then ONE iteration of the fused loop -- what moved, and by how much, against what the first TF1-Adam step can do (0.1)."""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
voff = tuple(v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if v)
case = make_case(seed=77, N=40_100, G=1100, C=8, K=1)
if os.environ.get("OVF", "1") != "0":          # counts above 255: the overflow list next to the 1-byte matrix (as the test has them)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
    case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = 1100
epss = np.stack([eps_for(1, G, 300 + i) for i in range(10)])
seen = Counter()
for r in range(reps):
    eng = HipEngine(**case, variant_off=voff)
    try:
        eng.gamma_init(eps_for(1, G, 0))
        s0 = eng.get_state()
        last = eng.iterate(1, epss[:2])
        s1 = eng.get_state()
    finally:
        eng.close()
    seen[last] += 1
    if seen[last] <= 1:
        print(f"run {r}: ELBO {last}")
        for n in s0:
            d = np.abs(np.asarray(s1[n], float) - np.asarray(s0[n], float))
            if d.size:
                bad = np.abs(d - 0.1) > 1e-6
                where = np.argwhere(bad)
                print(f"   {n:16s} moved by 0.1 in {int((~bad).sum())} of {d.size}; otherwise: {int(bad.sum())} entries, |move| min {d[bad].min() if bad.any() else 0:.3e} max {d[bad].max() if bad.any() else 0:.3e}"
                      + (f"; first rows {sorted(set(int(w[0]) for w in where[:2000]))[:12]} ... last {int(where[-1][0])}" if bad.any() else ""))
print(dict(seen), "variants off", voff)
