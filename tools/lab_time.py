"""us per iteration of ca_iterate for a (possibly lab, possibly wrong-result) library:  CLONEALIGN_HIP_LIB=/tmp/x.so python tools/lab_time.py [cells genes clones]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd import engine as E  # noqa: E402
import synth_data as synth  # noqa: E402
from tests._cases import eps_for  # noqa: E402

N, G, Cn = (int(a) for a in (sys.argv[1:4] + ["12500", "5000", "8"][len(sys.argv) - 1:]))
Yd, aux = synth.make_problem_torch(N, G, Cn, seed=20243, device="cuda:0")
psi0 = np.random.default_rng(1).normal(size=(N, 1))
loc0 = np.zeros(G) + 0.5
eng = E.HipEngine(None, aux["L"], psi0, loc0, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G))
eps = np.stack([eps_for(1, G, 10 + i) for i in range(600)])
best = 1e9
for rep in range(4):
    try:
        eng.iterate(50, eps[:100], want_elbo=False)
    except Exception:
        pass
    eng.synchronize()
    t0 = time.perf_counter()
    try:
        eng.iterate(300, eps, want_elbo=False)
    except Exception as ex:
        print("error", str(ex)[:80])
    eng.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
print("%.1f us per iteration" % best)
