"""Wall-clock of the whole fit (ca_run of 200 iterations from the initial values + 20 final ELBOs), with the eps draws handed over
by the caller (as R does) and with the built-in Philox stream (generated inside the call).
python tools/fit_time.py [cells genes clones]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.hostprep import safe_inverse_softplus  # noqa: E402

N, G, C = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (100_000, 5_000, 8)
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
col = (Yd.to(torch.float64) / (Yd.sum(1, keepdim=True).to(torch.float64) / G)).sum(0).cpu().numpy() if N * G <= 6e7 else None
if col is None:
    col = torch.zeros(G, dtype=torch.float64, device="cuda:0")
    rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
    for b0 in range(0, N, 8192):
        col += (Yd[b0:b0 + 8192].to(torch.float64) / rm[b0:b0 + 8192]).sum(0)
    col = col.cpu().numpy()
loc0 = safe_inverse_softplus(np.maximum(col / N, 1e-6))
psi0 = np.random.default_rng(1).normal(size=(N, 1))
eps = np.random.default_rng(2).normal(size=(2 + 2 * 400, 1, G)).astype(np.float32)
for label, src in (("caller's eps", eps), ("built-in stream", None), ("caller's eps", eps), ("built-in stream", None)):
    eng = HipEngine(None, aux["L"], psi0, loc0, 1, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G),
                    variant_on=tuple(v for v in __import__("os").environ.get("FIT_VARIANT_ON", "").split(",") if v))
    ts, tl = [], []
    for rep in range(6):
        eng.reinit(psi0, loc0)
        eng.synchronize()
        t0 = time.perf_counter()
        tr = eng.run(src, 200, 1e-6)
        t1 = time.perf_counter()
        fin = eng.final_elbo(None if src is None else eps[:20], 20)
        t2 = time.perf_counter()
        ts.append(t2 - t0); tl.append(t1 - t0)
    print(f"{N}x{G}x{C} {label:22s} ca_run {np.median(tl) * 1e3:7.2f} ms ({np.median(tl) / (len(tr) - 1) * 1e6:6.1f} us/iter, {len(tr) - 1} iterations)  "
          f"+ 20 final ELBOs {np.median(ts) * 1e3:7.2f} ms   last ELBO {tr[-1]:.6f}")
    if src is not None:   # steady state of the loop: (400 iterations - 200 iterations) / 200, the start-up (gamma init, first passes) cancels
        t = {}
        for n_it in (200, 400, 200, 400):
            eng.reinit(psi0, loc0)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run(src, n_it, 0.0)
            t[n_it] = min(t.get(n_it, 1e9), time.perf_counter() - t0)
        print(f"{N}x{G}x{C} ca_run steady state {(t[400] - t[200]) / 200 * 1e6:6.1f} us per iteration")
    eng.close()
