"""mc_samples = 2: gradients of one train pass with the two-sample backward sweep against the sweep per sample, variable by variable (bitwise)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case, perturbed_state  # noqa: E402
case = make_case(seed=71, N=700, G=1100, C=5, K=1, S=2)
G = case["Y"].shape[1]
res = []
for off in ((), ("s2_fuse",)):
    eng = HipEngine(**case, variant_off=off)
    st = perturbed_state({n: np.asarray(v).shape for n, v in eng.get_state().items()}, amp=0.2)
    for n, v in st.items():
        eng.set(n, v)
    g, e = eng.gradients(eps_for(2, G, 5))
    res.append((g, e))
    eng.close()
print("elbo", res[0][1], res[1][1], res[0][1] == res[1][1])
for n in res[0][0]:
    a, b = np.asarray(res[0][0][n]), np.asarray(res[1][0][n])
    d = np.abs(a - b)
    print(f"{n:14s} equal {np.array_equal(a, b)}  differing {int((d > 0).sum())} of {a.size}  max |diff| {d.max() if d.size else 0:.3e}  max |value| {np.abs(b).max() if b.size else 0:.3e}")
