import sys, numpy as np
sys.path.insert(0, '/root/repo')
from clonealign_amd.engine import HipEngine
from oracle.fused_numpy import FusedModel
from tests._cases import eps_for, make_case, perturbed_state
case = make_case(seed=61, N=40100, G=700, C=12, K=1)
ora = FusedModel(**case, dtype="float32")
st = perturbed_state({n: getattr(ora, n).shape for n in ora.VAR_NAMES}, amp=0.2)
engs = {"default": HipEngine(**case), "plain": HipEngine(**case, variant_off=("fused",))}
for n, v in st.items():
    setattr(ora, n, v.astype(ora.pdt))
    for e in engs.values(): e.set(n, v)
G = 700
epss = np.stack([eps_for(1, G, 100 + i) for i in range(10)])
for e in engs.values(): e.iterate(5, epss)
for i in range(5):
    ora.step(epss[2*i]); ora.elbo(epss[2*i+1])
go = np.asarray(ora.gamma_logits, dtype=np.float64)
for k, e in engs.items():
    g = e.get("gamma_logits")
    d = np.abs(g - go)
    i = np.unravel_index(d.argmax(), d.shape)
    print(k, "info", e.info()["fwd_mfma"], e.info()["bwd_mfma"], "max abs diff", d.max(), "at", i, "engine", g[i], "oracle", go[i], "n > 1e-4:", int((d > 1e-4).sum()), "n > 1e-5:", int((d > 1e-5).sum()), "max|logit|", np.abs(go).max())
d2 = np.abs(engs["default"].get("gamma_logits") - engs["plain"].get("gamma_logits"))
print("default vs plain: max", d2.max(), "n>1e-4", int((d2 > 1e-4).sum()))
