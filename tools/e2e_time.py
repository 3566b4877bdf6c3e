"""User-visible wall-clock of the Python mirror: clonealign() on a synthetic int32 count matrix, with a cProfile
summary of where the host time goes.   python tools/e2e_time.py [cells genes clones iters]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clonealign_amd.api import clonealign  # noqa: E402

N, G, C, iters = (int(a) for a in (sys.argv[1:5] + ["50000", "3000", "6", "100"][len(sys.argv) - 1:]))
rng = np.random.default_rng(5)
L = rng.choice([1, 2, 3, 4], size=(G, C), p=[0.29, 0.38, 0.23, 0.10]).astype(np.float64)
L[L.min(1) == L.max(1), 0] += 1
z = rng.integers(0, C, N)
base = rng.lognormal(-1.5, 1.2, G)
Y = rng.poisson(base[None, :] * L[:, z].T * 0.6).astype(np.int32)
Y[:, Y.sum(0) == 0] = 1
clonealign(Y[:2000], L, max_iter=3, verbose=False, seed=1)      # warm the library / first-use costs
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fit = clonealign(Y, L, max_iter=iters, verbose=False, seed=1)
pr.disable()
dt = time.perf_counter() - t0
acc = float(np.mean(np.array([ord(c[-1]) - 97 if c != "unassigned" else -1 for c in fit["clone"]]) == z))
print(f"clonealign() {N} x {G} x {C}, {iters} iterations: {dt:.3f} s; {len(fit['convergence_info']['elbo']) - 1} iterations run; "
      f"labels equal to the simulated clone: {acc:.3f}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

if os.environ.get("E2E_RESTARTS", "1") != "0":
    from clonealign_amd.api import run_clonealign  # noqa: E402
    t0 = time.perf_counter()
    best = run_clonealign(Y, L, initial_shrinks=(0, 5, 10), n_repeats=3, print_elbos=False, seed=2, max_iter=iters, verbose=False)
    t_shared = time.perf_counter() - t0
    ss = np.random.SeedSequence(2)
    seeds = [int(s.generate_state(1)[0]) for s in ss.spawn(9)]
    t0 = time.perf_counter()
    fits = [clonealign(Y, L, seed=s, max_iter=iters, verbose=False) for s in seeds]
    t_sep = time.perf_counter() - t0
    same = np.array_equal(best["multirun_info"]["elbos"], np.array([f["convergence_info"]["final_elbo"] for f in fits]))
    print(f"run_clonealign(), 9 restarts on one resident engine: {t_shared:.3f} s; nine separate clonealign() calls: {t_sep:.3f} s; "
          f"identical ELBOs: {same}")
