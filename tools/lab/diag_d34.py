import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests._cases import make_case, perturbed_state, eps_for
from clonealign_amd.engine import HipEngine
from oracle.fused_numpy import FusedModel
shape = dict(N=2100, G=600, C=3, K=3)
case = make_case(seed=67, **shape)
rng = np.random.default_rng(9)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 4000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
res = {}
for tag, kw in (("default", {}), ("valu", dict(variant_off=("fwd_mfma", "bwd_mfma"))), ("fwd_valu", dict(variant_off=("fwd_mfma",))), ("bwd_valu", dict(variant_off=("bwd_mfma",)))):
    eng, ora = HipEngine(**case, **kw), FusedModel(**case, dtype="float32")
    st = perturbed_state({n: getattr(ora, n).shape for n in ora.VAR_NAMES}, amp=0.2)
    for n, v in st.items():
        setattr(ora, n, v.astype(ora.pdt)); eng.set(n, v)
    G = ora.G
    n_iter = 5
    epss = np.stack([eps_for(1, G, 100 + i) for i in range(2 * n_iter)])
    tr = []
    for it in range(1, n_iter + 1):
        pass
    last = eng.iterate(n_iter, epss)
    es = []
    for i in range(n_iter):
        ora.step(epss[2 * i]); es.append(ora.elbo(epss[2 * i + 1]))
    p = eng.get_state()
    rel = {n: float(np.abs(p[n] - np.asarray(getattr(ora, n), float)).max() / max(np.abs(np.asarray(getattr(ora, n), float)).max(), 1e-30)) for n in ora.VAR_NAMES if np.asarray(getattr(ora, n)).size}
    i = eng.info()
    print(tag, "fwd_mfma", i["fwd_mfma"], "bwd_mfma", i["bwd_mfma"], "rel elbo", abs(last - es[-1]) / abs(es[-1]), {k: f"{v:.1e}" for k, v in rel.items()})
    eng.close()
# per-iteration ELBO drift of the default path
eng, ora = HipEngine(**case), FusedModel(**case, dtype="float32")
st = perturbed_state({n: getattr(ora, n).shape for n in ora.VAR_NAMES}, amp=0.2)
for n, v in st.items():
    setattr(ora, n, v.astype(ora.pdt)); eng.set(n, v)
epss = np.stack([eps_for(1, ora.G, 100 + i) for i in range(10)])
for i in range(5):
    l = eng.iterate(1, epss[2 * i:2 * i + 2]); ora.step(epss[2 * i]); e = ora.elbo(epss[2 * i + 1])
    print(i, l, e, abs(l - e) / abs(e))
