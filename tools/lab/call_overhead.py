"""us per iteration of back-to-back ca_iterate(n) calls: host draws against the engine's built-in stream (eps = NULL), n = 20 and 200."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import synth_data as synth
import bench
from clonealign_amd.engine import HipEngine
from clonealign_amd.hostprep import safe_inverse_softplus
N, G, C = 100000, 5000, 8
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
col = sum((Yd[b:b + 8192].to(torch.float64) / rm[b:b + 8192]).sum(0) for b in range(0, N, 8192))
loc0 = safe_inverse_softplus(np.maximum(col.cpu().numpy() / N, 1e-6))
rng = np.random.default_rng(1)
psi0 = rng.normal(size=(N, 1))
eng = HipEngine(None, aux["L"], psi0, loc0, 1, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), profile=0)
eng.gamma_init(rng.normal(size=(1, G)).astype(np.float32))
for n in (20, 200):
    eps = bench.draws_for_calls(rng, n, 1, G)
    for tag, e in (("host draws", eps), ("built-in stream", None)):
        for _ in range(3):
            eng.iterate(n, e)
        eng.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.iterate(n, e); eng.synchronize()
            ts.append((time.perf_counter() - t0) / n * 1e6)
        print(f"ca_iterate({n}) {tag:16s}: median {np.median(ts):.1f} us/iter  min {min(ts):.1f}")
for tag, mask in (("no events", 0), ("events on every 8th stream launch", (1 << 2) | (7 << 8)), ("events on every launch of every class", 0x1F)):
    eng.set_profile(mask)
    eps = bench.draws_for_calls(rng, 20, 1, G)
    for _ in range(3):
        eng.iterate(20, eps)
    eng.synchronize()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.iterate(20, eps); eng.synchronize(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e6)
    eng.kernel_times(reset=True)
    print(f"ca_iterate(20), {tag}: median {np.median(ts):.1f} us/iter  min {min(ts):.1f}")
eng.close()
