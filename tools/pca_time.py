import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from tests.test_gpu_scale import _synth
from clonealign_amd.engine import HipEngine
N, G, C = 100_000, 5_000, 8
Yd, L, psi0, loc0 = _synth(N, G, C)
eng = HipEngine(None, L, np.zeros((N, 1)), loc0, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G))
for it in (10, 40):
    t = time.perf_counter(); p = eng.pca_init(None, n_iter=it, seed=1); dt = time.perf_counter() - t
    print(f"device PCA init 100k x 5k, K=1, {it} iterations: {dt*1e3:.0f} ms; sd {p.std(ddof=1):.6f}")
    if it == 10: p10 = p.copy()
print("10 vs 40 iterations max diff", np.abs(p10 - p).max())
