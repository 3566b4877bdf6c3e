import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from clonealign_amd.engine import HipEngine
from clonealign_amd.inference import run_vi_loop
from clonealign_amd.rng import EpsStream
from oracle.fused_numpy import FusedModel
from tests._cases import make_case
# replay the fuzz tool's random stream up to case 321 of seed 37 --r6 (same draws as tools/fuzz_parity.py)
rng = np.random.default_rng(37)
NV = 21
for it in range(322):
    N = int(rng.integers(1, 900)); G = int(rng.integers(1, 700))
    C = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 19))
    K = int(rng.choice([0, 1, 1, 1, 2])); P = int(rng.choice([0, 0, 0, 1])) if K > 0 else 0
    S = 1 if rng.random() < 0.8 else 2
    C = int(rng.integers(1, 37)); K = int(rng.choice([0, 1, 1, 2, 3, 4])); P = int(rng.choice([0, 0, 1, 2, 3])) if K > 0 else 0
    if K + P > 5: P = 5 - K
    S = int(rng.choice([1, 1, 2, 3, 4])); N = int(rng.integers(1, 2500))
    vi = int(rng.integers(0, NV))
    kw = dict(N=N, G=G, C=C, K=K, S=S)
    if P: kw["P"] = P
    seed = int(rng.integers(0, 10**6))
    if it < 321:
        frac = rng.random() < 0.25
        if frac: rng.random((G, C))
        if rng.random() < 0.4:
            sz = max(1, N * G // 3000); rng.integers(0, N * G, size=sz); rng.integers(200, 2000, size=sz)
        rng.integers(1, 6)
        continue
    case = make_case(seed=seed, **kw)
    if rng.random() < 0.25: case["L"] = case["L"] + rng.random(case["L"].shape) * 0.9
    if rng.random() < 0.4:
        idx = rng.integers(0, case["Y"].size, size=max(1, case["Y"].size // 3000)); case["Y"].reshape(-1)[idx] += rng.integers(200, 2000, size=idx.size)
    n_iter = int(rng.integers(1, 6))
print(kw, "n_iter", n_iter, "integer L", bool(np.all(case["L"] == np.round(case["L"]))))
ora = FusedModel(**case, dtype="float32")
to = np.asarray(run_vi_loop(ora, EpsStream(3, S, G), n_iter, 1e-12))
for tag, voff in (("as in the fuzz case", ("y_mfma1", "y_ride")), ("default", ()), ("fwd on the vector unit", ("fwd_mfma",)), ("way back on the vector unit", ("bwd_mfma",)), ("both", ("fwd_mfma", "bwd_mfma"))):
    eng = HipEngine(**case, variant_off=voff)
    tr = np.asarray(eng.run(EpsStream(3, S, G), n_iter, 1e-12))
    W = eng.get_state()["W"]
    i = eng.info()
    print(f"{tag:28s} fwd_mfma {i['fwd_mfma']} bwd_mfma {i['bwd_mfma']} trace {np.abs(tr - to).max() / np.abs(to).max():.2e}  W[64] engine {W.ravel()[64]:+.4f} oracle {np.asarray(ora.W).ravel()[64]:+.4f}  max |dW| {np.abs(W - np.asarray(ora.W, float)).max():.2e}")
    eng.close()
