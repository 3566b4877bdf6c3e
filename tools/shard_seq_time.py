"""us per iteration of ca_iterate for ONE shard run as a rank of a sharded fit: world = 1 with the peer-to-peer transport committed (the rank's
all-reduce publishes to itself, waits for its own flag, sums one inbox), so the SEQUENCE of launches a rank of a sharded run makes -- and what
riding work in the all-reduce saves -- can be timed on one device.  The collective's own cost across devices is NOT in this number.
   python tools/shard_seq_time.py [cells genes clones] [--variant-off a,b]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd import engine as E  # noqa: E402
import synth_data as synth  # noqa: E402
from tests._cases import eps_for  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
voff = ()
for i, a in enumerate(sys.argv):
    if a == "--variant-off":
        voff = tuple(v for v in sys.argv[i + 1].split(",") if v)
        args = [x for x in args if x != sys.argv[i + 1]]
N, G, Cn = (int(a) for a in (args[:3] + ["12500", "5000", "8"][len(args):]))
Yd, aux = synth.make_problem_torch(N, G, Cn, seed=20243, device="cuda:0")
psi0 = np.random.default_rng(1).normal(size=(N, 1))
loc0 = np.zeros(G) + 0.5
eng = E.HipEngine(None, aux["L"], psi0, loc0, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), variant_off=voff)
eng._p2p_setup(lambda payload: [payload], 1)          # a world of one: its own handle, its own flag
info = eng.info()
assert info["transport_name"] == "p2p", info
eps = np.stack([eps_for(1, G, 10 + i) for i in range(600)])
best = 1e9
for rep in range(4):
    eng.iterate(50, eps[:100], want_elbo=False)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.iterate(300, eps, want_elbo=False)
    eng.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
print("%d x %d x %d as a rank of a sharded fit (p2p, world 1), variants off %s: %.1f us per iteration" % (N, G, Cn, list(voff), best))
eng.close()
