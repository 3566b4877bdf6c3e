"""Short series-form run for rocprofv3 --kernel-trace --stats:  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_series -- python3 tools/series_prof.py N G C"""
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.hostprep import safe_inverse_softplus  # noqa: E402

N, G, C = (int(a) for a in sys.argv[1:4])
von = () if (len(sys.argv) > 4 and sys.argv[4] == "sweeps") else ("series",)
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
col = (Yd.to(torch.float64) / rm).sum(0) if N * G < 6e7 else sum((Yd[b:b + 8192].to(torch.float64) / rm[b:b + 8192]).sum(0) for b in range(0, N, 8192))
loc0 = safe_inverse_softplus(np.maximum(col.cpu().numpy() / N, 1e-6))
rng = np.random.default_rng(1)
psi0 = rng.normal(size=(N, 1))
torch.cuda.synchronize()
steps = 60
eps = rng.normal(size=(2 * steps + 1, 1, G)).astype(np.float32)
eps[-1] = eps[0]
eng = HipEngine(None, aux["L"], psi0, loc0, 1, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), variant_on=von, profile=0)
eng.gamma_init(eps[0])
for _ in range(4):
    eng.iterate(steps, eps)
eng.synchronize()
eng.close()
