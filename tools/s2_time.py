"""Iteration time with mc_samples = 2 (and any clone count) at the benchmark size: the fused matrix-core path against the plain passes.
    python tools/s2_time.py [cells genes clones S]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine  # noqa: E402
N, G, C, S = (int(a) for a in (sys.argv[1:5] + ["100000", "5000", "8", "2"][len(sys.argv) - 1:]))
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
psi0 = np.random.default_rng(1).normal(size=(N, 1)); loc0 = np.zeros(G) + 0.5
rng = np.random.default_rng(2)
for name, kw in (("matrix-core sweeps", {}), ("... a sweep per pass (variant_off s2_fuse)", dict(variant_off=("s2_fuse",))), ("plain passes (variant_off fused)", dict(variant_off=("fused",)))):
    eng = HipEngine(None, aux["L"], psi0, loc0, 1, S, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), **kw)
    eps = rng.normal(size=(2 * 50, S, G)).astype(np.float32)
    eng.gamma_init(eps[0]); eng.iterate(50, eps); eng.synchronize()
    t0 = time.perf_counter(); eng.iterate(50, eps); eng.iterate(50, eps); eng.synchronize(); dt = (time.perf_counter() - t0) / 100
    i = eng.info()
    print(f"{N} x {G} x {C}, S = {S}: {name:44s} {1 / dt:8.1f} it/s  {dt * 1e3:.4f} ms   fused {i['fused_sweep']} fwd_mfma {i['fwd_mfma']} bwd_mfma {i['bwd_mfma']}")
    eng.close()
