import sys, numpy as np
sys.path.insert(0, '/root/repo')
from clonealign_amd.engine import HipEngine
from clonealign_amd.inference import run_vi_loop
from clonealign_amd.rng import EpsStream
from oracle.fused_numpy import FusedModel
from tests._cases import make_case
for seed in range(6):
    case = make_case(seed=seed, N=506, G=1, C=8, K=1, S=2)
    ora = FusedModel(**case, dtype="float32")
    to = np.asarray(run_vi_loop(ora, EpsStream(3, 2, 1), 3, 1e-12))
    out = []
    for kw in ({}, dict(variant_off=("y_mfma1",)), dict(variant_off=("y_mfma1", "y_ride"))):
        e = HipEngine(**case, **kw)
        tr = np.asarray(e.run(EpsStream(3, 2, 1), 3, 1e-12))
        out.append(float(np.abs(tr - to).max() / np.abs(to).max()))
        e.close()
    print(seed, ["%.1e" % v for v in out], to[:2])
