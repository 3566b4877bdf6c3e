import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from clonealign_amd.engine import HipEngine, HipGroupEngine
from clonealign_amd.rng import EpsStream
from tests._cases import make_case
case = make_case(seed=31, N=1301, G=700, C=8, K=1)
one = HipEngine(**case); tr1 = one.run(EpsStream(77, 1, 700), 8, 1e-12); one.close()
t0 = time.time()
try:
    grp = HipGroupEngine(**case, devices=[0, 0], transport="p2p", variant_on=("p2p_same_device",), comm_timeout_ms=3000)
    print("created", grp.group_info(), time.time() - t0)
    trg = grp.run(EpsStream(77, 1, 700), 8, 1e-12)
    print("run ok", np.abs(trg - tr1).max() / np.abs(tr1).max(), time.time() - t0)
    fin = grp.final_elbo(EpsStream(78, 1, 700), 4); print("final ok")
    grp.close()
except Exception as e:
    print("FAILED:", repr(e)[:300], time.time() - t0)
