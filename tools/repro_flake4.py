"""Which call gives a different result when ANOTHER PROCESS shares the GPU (tools/corun.py)?  Fresh engines; hashes of what each call leaves.
   python tools/repro_flake4.py [reps]"""
import hashlib
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
voff = tuple(v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if v)
case = make_case(seed=77, N=40_100, G=1100, C=8, K=1)
if os.environ.get("OVF", "1") != "0":          # counts above 255: the overflow list next to the 1-byte matrix (as the test has them)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
    case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = 1100
hs = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]  # noqa: E731
out = {k: Counter() for k in ("gamma_init logits", "elbo (plain pass)", "elbo again", "gradients W", "gradients psi", "gradients logits", "step: W", "step: psi")}
for r in range(reps):
    eng = HipEngine(**case, variant_off=voff)
    try:
        eng.gamma_init(eps_for(1, G, 0))
        out["gamma_init logits"][hs(eng.get("gamma_logits"))] += 1
        out["elbo (plain pass)"][eng.elbo(eps_for(1, G, 1))] += 1
        out["elbo again"][eng.elbo(eps_for(1, G, 1))] += 1
        g, _ = eng.gradients(eps_for(1, G, 2))
        out["gradients W"][hs(g["W"])] += 1
        out["gradients psi"][hs(g["psi"])] += 1
        out["gradients logits"][hs(g["gamma_logits"])] += 1
        eng.step(eps_for(1, G, 2))
        out["step: W"][hs(eng.get("W"))] += 1
        out["step: psi"][hs(eng.get("psi"))] += 1
    finally:
        eng.close()
for k, c in out.items():
    print(f"{k:22s} {len(c)} distinct: {dict(c.most_common(4))}")
