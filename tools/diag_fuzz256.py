"""Diagnosis of fuzz case 256 of `tools/fuzz_parity.py 300 401` (N=451, G=481, C=16, K=1, P=1): per-variable gradients of the engine against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from oracle.fused_numpy import FusedModel  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402


def the_case(want=256, seed=401, nv=18):
    rng = np.random.default_rng(seed)
    for it in range(want + 1):
        N = int(rng.integers(1, 900)); G = int(rng.integers(1, 700))
        C = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 19))
        K = int(rng.choice([0, 1, 1, 1, 2])); P = int(rng.choice([0, 0, 0, 1])) if K > 0 else 0
        S = 1 if rng.random() < 0.8 else 2
        rng.integers(0, nv)
        kw = dict(N=N, G=G, C=C, K=K, S=S)
        if P:
            kw["P"] = P
        cseed = int(rng.integers(0, 10**6))
        frac = rng.random() < 0.25
        Lf = rng.random((G, C)) * 0.9 if frac else None
        ovf = rng.random() < 0.4
        if ovf:
            sz = max(1, (N * G) // 3000)
            idx = rng.integers(0, N * G, size=sz); add = rng.integers(200, 2000, size=sz)
        rng.integers(1, 6)
        if it == want:
            case = make_case(seed=cseed, **kw)
            if frac:
                case["L"] = case["L"] + Lf
            if ovf:
                case["Y"].reshape(-1)[idx] += add
            print("case", kw, "fractional L", frac, "overflow", ovf)
            return case


case = the_case()
G, S = case["Y"].shape[1], case["S"]
for voff in ((), ("bwd_mfma",), ("fwd_mfma",)):
    eng, ora = HipEngine(**case, variant_off=voff), FusedModel(**case, dtype="float32")
    try:
        e0, e1 = eps_for(S, G, 1), eps_for(S, G, 2)
        eng.gamma_init(e0); ora.gamma_init(e0)
        for step in range(2):
            ge, ea = eng.gradients(e1)
            go, eo = ora.gradients(e1)
            print("variants off", voff, "step", step, "elbo rel", abs(ea - eo) / abs(eo), "info", {k: eng.info()[k] for k in ("fwd_mfma", "bwd_mfma", "fwd_cell", "fused_sweep")})
            for n in ora.VAR_NAMES:
                a, b = np.asarray(ge[n], float), np.asarray(go[n], float)
                if a.size:
                    d = np.abs(a - b)
                    i = np.unravel_index(d.argmax(), d.shape)
                    print(f"   call-by-call d/d{n:15s} max abs diff {d.max():.3e} of {np.abs(b).max():.3e} at {i}: engine {a[i]:.6e} oracle {b[i]:.6e}")
            eng.step(e1); ora.step(e1)
        # the loop's kernels: one iteration of ca_iterate from here, state against the oracle's
        eps = np.stack([eps_for(S, G, 30 + i) for i in range(2)])
        le = eng.iterate(1, eps)
        ora.step(eps[0]); lo = ora.elbo(eps[1])
        print("   ca_iterate(1): elbo rel", abs(le - lo) / abs(lo))
        se = eng.get_state()
        for n in ora.VAR_NAMES:
            a, b = np.asarray(se[n], float), np.asarray(getattr(ora, n), float)
            if a.size:
                d = np.abs(a - b)
                i = np.unravel_index(d.argmax(), d.shape)
                print(f"   loop state {n:15s} max abs diff {d.max():.3e} of {np.abs(b).max():.3e} at {i}: engine {a[i]:.6e} oracle {b[i]:.6e}")
    finally:
        eng.close()

# ---- the fuzz's own sequence: ca_run with EpsStream(3, S, G); after which iteration does a coordinate part ways, and what was its gradient?
from clonealign_amd.rng import EpsStream  # noqa: E402
print("\n=== the sweep's sequence (ca_run, EpsStream(3)) ===")
ora = FusedModel(**case, dtype="float32")
es = EpsStream(3, S, G)
ora.gamma_init(es.next()); ora.elbo(es.next())
for it in range(1, 5):
    e_tr = es.next()
    go, _ = ora.gradients(e_tr)
    before = {n: np.array(getattr(ora, n), float) for n in ora.VAR_NAMES}
    ora.step(e_tr); ora.elbo(es.next())
    eng = HipEngine(**case)
    try:
        eng.run(EpsStream(3, S, G), it, 1e-12)
        se = eng.get_state()
    finally:
        eng.close()
    for n in ("W", "beta", "psi", "loc", "ls"):
        a, b = np.asarray(se[n], float), np.asarray(getattr(ora, n), float)
        d = np.abs(a - b)
        bad = np.argwhere(d > 0.02)
        if len(bad):
            i = tuple(bad[0])
            print(f"after iteration {it}: {n}{i} engine {a[i]:.5f} oracle {b[i]:.5f} ({len(bad)} coordinates off by > 0.02); oracle's gradient there at this step {np.asarray(go[n])[i]:.3e} "
                  f"(largest |gradient| of {n}: {np.abs(np.asarray(go[n])).max():.3e}); value before the step {before[n][i]:.5f}")
