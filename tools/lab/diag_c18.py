import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests._cases import make_case
from clonealign_amd.engine import HipEngine
from clonealign_amd.inference import run_vi_loop
from clonealign_amd.rng import EpsStream
from oracle.fused_numpy import FusedModel
case = make_case(seed=91, N=9000, G=5000, C=18, K=1)
G = 5000
ora = FusedModel(**case, dtype="float32")
to = np.asarray(run_vi_loop(ora, EpsStream(3, 1, G), 4, 1e-12))
for tag, kw in (("default", {}), ("valu", dict(variant_off=("fwd_mfma", "bwd_mfma"))), ("fwd_valu", dict(variant_off=("fwd_mfma",))), ("bwd_valu", dict(variant_off=("bwd_mfma",)))):
    eng = HipEngine(**case, **kw)
    tr = np.asarray(eng.run(EpsStream(3, 1, G), 4, 1e-12))
    st = eng.get_state()
    i = eng.info()
    print(tag, "fwd_mfma", i["fwd_mfma"], "bwd_mfma", i["bwd_mfma"], "trace", np.abs(tr - to).max() / np.abs(to).max())
    for n in ("W", "loc", "ls", "psi"):
        b = np.asarray(getattr(ora, n), float); a = np.asarray(st[n], float)
        d = np.abs(a - b) / np.abs(b).max()
        w = np.argsort(d.ravel())[::-1][:4]
        print("   ", n, "max", d.max(), "n>1e-4", int((d > 1e-4).sum()), "median", np.median(d), "worst idx", w.tolist(), "engine", a.ravel()[w[:2]], "oracle", b.ravel()[w[:2]])
    eng.close()
print("colsum of worst genes", case["Y"].sum(0)[[0]], "min colsum", case["Y"].sum(0).min())
