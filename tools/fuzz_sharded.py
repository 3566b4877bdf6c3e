"""Randomised sweep of the cell-sharded loop (not part of the test suite): W shards of random sizes as W handles on one GPU,
joined by the host all-reduce hook, whole loops (ca_run + final ELBOs) against ONE handle that holds all cells.

    python tools/fuzz_sharded.py [n_cases] [seed]
"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.rng import EpsStream  # noqa: E402
from clonealign_amd.sharding import cell_range  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402


class HostAllreduce:
    def __init__(self, world):
        self.world, self.bar, self.slots, self.calls = world, threading.Barrier(world), [None] * world, 0

    def make(self, rank):
        def fn(buf):
            self.slots[rank] = buf.copy()
            self.bar.wait()
            tot = self.slots[0].copy()
            for r in range(1, self.world):
                tot += self.slots[r]
            self.bar.wait()
            buf[:] = tot
            if rank == 0:
                self.calls += 1
        return fn


_a = [a for a in sys.argv[1:] if not a.startswith("--")]
WIDE = "--wide" in sys.argv   # round 5: shards ABOVE the side-stream threshold (4e7 counts each), up to 20 clones, mc_samples up to 3
n_cases = int(_a[0]) if len(_a) > 0 else 40
rng = np.random.default_rng(int(_a[1]) if len(_a) > 1 else 3)
fails = 0
for it in range(n_cases):
    W = int(rng.choice([2, 2, 3]))
    N = int(rng.integers(W * 2, 1200))
    G = int(rng.integers(2, 600))
    C = int(rng.integers(1, 9))
    K = int(rng.choice([0, 1, 1, 2]))
    P = int(rng.choice([0, 0, 1])) if K > 0 else 0
    S = 1 if rng.random() < 0.85 else 2
    if WIDE:
        G = int(rng.integers(1500, 5200))
        N = W * int(rng.integers(int(4.2e7 / G), int(7e7 / G)))
        C = int(rng.choice([3, 8, 8, 12, 16, 18, 20]))
        S = int(rng.choice([1, 1, 2, 3]))
    kw = dict(N=N, G=G, C=C, K=K, S=S)
    if P:
        kw["P"] = P
    case = make_case(seed=int(rng.integers(0, 10**6)), **kw)
    if rng.random() < 0.3:
        idx = rng.integers(0, case["Y"].size, size=max(1, case["Y"].size // 3000))
        case["Y"].reshape(-1)[idx] += rng.integers(200, 2000, size=idx.size)
    n_iter = int(rng.integers(1, 6))
    eps_f = np.stack([eps_for(S, G, 50 + i) for i in range(3)])

    def drive(eng):
        tr = np.asarray(eng.run(EpsStream(3, S, G), n_iter, 1e-12))
        fe = eng.final_elbo(eps_f, 3)
        return tr, fe, eng.get_state()

    try:
        ref = HipEngine(**case)
        tr0, fe0, st0 = drive(ref)
        ref.close()
        ar, out, errs = HostAllreduce(W), [None] * W, []

        def worker(rank):
            try:
                lo, hi = cell_range(N, rank, W)
                shard = dict(case)
                for k in ("Y", "psi0", "X", "extra_loglik"):
                    if shard.get(k) is not None:
                        shard[k] = shard[k][lo:hi]
                eng = HipEngine(**shard, rank=rank, world=W, host_allreduce=ar.make(rank))
                out[rank] = drive(eng) + ((lo, hi),)
                eng.close()
            except Exception as exc:   # noqa: BLE001
                errs.append(repr(exc))
                ar.bar.abort()

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if errs:
            raise RuntimeError("; ".join(errs))
        why = []
        for r in range(W):
            tr, fe, st, (lo, hi) = out[r]
            if tr.shape != tr0.shape or np.abs(tr - tr0).max() > 1e-5 * np.abs(tr0).max():
                why.append("rank %d trace %.2e" % (r, float(np.abs(tr - tr0).max() / np.abs(tr0).max())))
            if np.abs(fe - fe0).max() > 1e-5 * np.abs(fe0).max():
                why.append("rank %d final elbo" % r)
            for n in ("W", "v", "beta", "alpha_unconstr", "loc", "ls"):
                if not np.array_equal(st[n], out[0][2][n]):
                    why.append("rank %d replica of %s differs" % (r, n))
        if why:
            fails += 1
            print("FAIL", kw, "W", W, "iters", n_iter, "|", "; ".join(why), flush=True)
        elif WIDE:
            print("ok  ", kw, "W", W, "iters", n_iter, flush=True)
    except Exception as exc:   # noqa: BLE001
        fails += 1
        print("ERROR", kw, "W", W, repr(exc))
print(f"{n_cases - fails} of {n_cases} sharded cases agree with the single-handle fit")
sys.exit(1 if fails else 0)
