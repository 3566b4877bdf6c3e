"""Hunting the rare deviation of tests/test_gpu_parity.py::test_riding_dispatch_order...[shape1] (seen on some boxes only, profiles/r04_flake.txt): many fresh
engines, k = 1 .. 5 iterations each; the first state that is not bitwise the majority's is saved with a normal one for an offline diff.
   python tools/repro_flake3.py [engines per k] [out dir]"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r4", "flake")
case = make_case(seed=77, N=40_100, G=1100, C=8, K=1)
rng = np.random.default_rng(3)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = case["Y"].shape[1]
epss = np.stack([eps_for(1, G, 300 + i) for i in range(10)])
pats = [None, "255:1", -3, -64, "1:200"]
for k in (1, 2, 5):
    seen, states = Counter(), {}
    for r in range(reps):
        pat = pats[r % len(pats)]
        eng = HipEngine(**case, tune=({} if pat is None else {"ride_pattern": pat}))
        try:
            eng.gamma_init(eps_for(1, G, 0))
            last = eng.iterate(k, epss[:2 * k])
            st = eng.get_state()
        finally:
            eng.close()
        seen[last] += 1
        states.setdefault(last, (pat, st))
    print(f"k = {k}: {dict(seen)}")
    if len(seen) > 1:
        os.makedirs(out, exist_ok=True)
        major = seen.most_common(1)[0][0]
        for v, (pat, st) in states.items():
            np.savez_compressed(os.path.join(out, f"k{k}_{'normal' if v == major else 'deviant'}_{abs(hash(v)) % 10**6}.npz"), last=v, pat=str(pat), **st)
        a = states[major][1]
        for v, (pat, st) in states.items():
            if v == major:
                continue
            print(f"  deviant (pattern {pat}) {v} vs normal {major}")
            for n in a:
                d = np.abs(np.asarray(st[n], float) - np.asarray(a[n], float))
                if d.size and d.max() > 0:
                    print(f"    {n:16s} differs in {int((d > 0).sum())} of {d.size} entries, max {d.max():.3e} (|normal| max {np.abs(a[n]).max():.3e}), first at {tuple(int(x) for x in np.argwhere(d > 0)[0])}")
        break
