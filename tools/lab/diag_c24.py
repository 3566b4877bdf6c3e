import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests._cases import make_case, perturbed_state, eps_for
from clonealign_amd.engine import HipEngine
from oracle.fused_numpy import FusedModel
shape = dict(N=40_100, G=700, C=24, K=1)
case = make_case(seed=71, **shape)
rng = np.random.default_rng(9)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 4000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
ora = FusedModel(**case, dtype="float32")
st = perturbed_state({n: getattr(ora, n).shape for n in ora.VAR_NAMES}, amp=0.2)
for n, v in st.items():
    setattr(ora, n, v.astype(ora.pdt))
G = ora.G
n_iter = 5
epss = np.stack([eps_for(1, G, 100 + i) for i in range(2 * n_iter)])
g_steps = []
for i in range(n_iter):
    g_steps.append(np.asarray(ora.gradients(epss[2 * i])[0]["gamma_logits"], float))
    ora.step(epss[2 * i]); e = ora.elbo(epss[2 * i + 1])
g_steps = np.stack(g_steps)
print("typical |g| median", np.median(np.abs(g_steps[0])), "p1", np.percentile(np.abs(g_steps[0]), 1))
ref = np.asarray(ora.psi, float)
refg = np.asarray(ora.gamma_logits, float)
st_g = st["gamma_logits"]
for tag, kw in (("default", {}), ("valu", dict(variant_off=("fwd_mfma", "bwd_mfma"))), ("fwd_valu", dict(variant_off=("fwd_mfma",))), ("bwd_valu", dict(variant_off=("bwd_mfma",)))):
    eng = HipEngine(**case, **kw)
    for n, v in st.items():
        eng.set(n, v)
    last = eng.iterate(n_iter, epss)
    p = eng.get_state()
    d = np.abs(p["psi"] - ref) / np.abs(ref).max()
    i = eng.info()
    print(tag, "fwd_mfma", i["fwd_mfma"], "bwd_mfma", i["bwd_mfma"], "rel elbo", abs(last - e) / abs(e), "psi max", d.max(), "n>1e-4", int((d > 1e-4).sum()), "n>3e-5", int((d > 3e-5).sum()), "argmax", int(d.argmax()))
    dg = np.abs(p["gamma_logits"] - refg) / np.abs(refg).max()
    bad = np.argwhere(dg > 1e-4)
    print("   gamma_logits max", dg.max(), "n>1e-4", len(bad), "max|ref|", np.abs(refg).max())
    gm = np.abs(g_steps[:, dg > 1e-4]).min(0)
    print("   min over steps of |oracle g| at the deviating coords: max", gm.max(), "median", np.median(gm), "; count of ALL coords with min|g| below that max:", int((np.abs(g_steps).min(0) <= gm.max()).sum()))
    for (a, b) in bad[:3]:
        print("     cell", a, "clone", b, "start", st_g[a, b], "engine", p["gamma_logits"][a, b], "oracle", refg[a, b], "row max logit", refg[a].max())
    eng.close()
