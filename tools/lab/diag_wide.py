import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests._cases import make_case, eps_for
from clonealign_amd.engine import HipEngine
case = make_case(seed=35, N=1301, G=700, C=8, K=1)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
case["psi0"] = case["psi0"] * scale
G = 700
eps = np.stack([eps_for(1, G, 700 + i) for i in range(25)])
one = HipEngine(**case, variant_on=("series",))
one.gamma_init(eps[0])
for i in range(8):
    try:
        a = one.iterate(1, eps[2 * i:2 * i + 2])
    except Exception as ex:
        print("iteration", i, "error:", str(ex)[:160]); break
    inf = one.info(); st = one.get_state()
    W = st["W"]; psi = st["psi"]
    print(i, "elbo", a, "series", inf["series_passes"], "fallbacks", inf["series_fallbacks"], "max|psi|", float(np.abs(psi).max()), "W range", float(W.max() - W.min()), "product", float(np.abs(psi).max() * (W.max() - W.min())))
