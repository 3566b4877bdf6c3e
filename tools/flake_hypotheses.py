"""Offline (CPU, numpy oracle): which wrong computation gives the alternative result of profiles/r04_flake.txt?  The engine's loop on the test's problem
(40 100 x 1100 x 8, gamma init + 5 iterations) with ONE thing made stale by one parameter state at iteration k: the count-matrix products
Y.W / Y^T psi (what the riding stream delivers), for k = 1 .. 5, and variants.  Prints the last monitor ELBO of every hypothesis next to the two observed values.
   python tools/flake_hypotheses.py"""
import inspect
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fused_numpy as fn  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

NORMAL, DEVIANT = -132399942.66614686, -131740535.4721053

src_e = inspect.getsource(fn.FusedModel._elbo_from).replace("YW = self.Y @ V[:, :self.K]", "YW = self._yw(V)")
src_g = inspect.getsource(fn.FusedModel.gradients).replace("YtPsi = self.Y.T @ self.psi.astype(np.float64) if K > 0 else np.zeros((G, 0))", "YtPsi = self._ytpsi()")
ns = dict(fn.__dict__)
exec("class Patched(FusedModel):\n" + src_e + "\n" + src_g, ns)
Patched = ns["Patched"]


class Model(Patched):
    stale_W = None      # when set: the W the row products are taken with
    stale_psi = None    # when set: the psi the column products are taken with

    def _yw(self, V):
        W = V[:, :self.K] if self.stale_W is None else self.stale_W
        return self.Y @ W

    def _ytpsi(self):
        p = self.psi.astype(np.float64) if self.stale_psi is None else self.stale_psi
        return self.Y.T @ p


case = make_case(seed=77, N=40_100, G=1100, C=8, K=1)
rng = np.random.default_rng(3)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = 1100
epss = np.stack([eps_for(1, G, 300 + i) for i in range(10)])


def run(hyp):
    """hyp = (k, what): at parameter state k (after train pass k; 0 = after gamma init) the products named by `what` are the previous state's."""
    m = Model(**case, dtype="float32")
    m.gamma_init(eps_for(1, G, 0))
    prev = (m.W.astype(np.float64).copy(), m.psi.astype(np.float64).copy())
    last = None
    for i in range(5):
        k_state = i          # parameters in force during train pass i + 1 are state i
        stale = hyp is not None and hyp[0] == k_state
        m.stale_W = prev[0] if stale and "W" in hyp[1] else None
        m.stale_psi = prev[1] if stale and "psi" in hyp[1] else None
        before = (m.W.astype(np.float64).copy(), m.psi.astype(np.float64).copy())
        m.step(epss[2 * i])                      # train pass i + 1 on state i
        prev = before
        k_state = i + 1
        stale = hyp is not None and hyp[0] == k_state
        m.stale_W = prev[0] if stale and "W" in hyp[1] else None
        m.stale_psi = None
        last = m.elbo(epss[2 * i + 1])           # monitor pass on state i + 1
    return last


print(f"observed: normal {NORMAL:.4f}   deviant {DEVIANT:.4f}   (difference {DEVIANT - NORMAL:.1f})")
base = run(None)
print(f"oracle, nothing stale: {base:.4f}  (engine normal - oracle {NORMAL - base:.2f})")
for k in range(0, 6):
    for what in (("W", "psi"), ("W",), ("psi",)):
        v = run((k, what))
        print(f"stale {'+'.join(what):6s} products at state {k}: {v:.4f}   minus oracle normal {v - base:12.1f}   (observed deviation {DEVIANT - NORMAL:.1f})")
