"""Per-kernel-class time of one iteration against the cell count (the staircase of tile quantisation):
python tools/stair_time.py genes clones N1 N2 ... [--tune name=value,...]   -- us per iteration un-profiled, then fwd / bwd / other from HIP events around every launch"""
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
tune = {}
voff = ()
for a in sys.argv[1:]:
    if a.startswith("--ab="):          # A/B: every size with and without these variants, alternating in one process
        voff = tuple(a[5:].split(","))
    if a.startswith("--tune="):
        tune = {k: (v if ":" in v else int(v)) for k, v in (kv.split("=") for kv in a[7:].split(",") if kv)}
G, Cn = int(args[0]), int(args[1])
eps = np.stack([eps_for(1, G, 10 + i) for i in range(400)])
for N in (int(a) for a in args[2:]):
    Yd, aux = synth.make_problem_torch(N, G, Cn, seed=20243, device="cuda:0")
    import torch
    torch.cuda.synchronize()   # (the engine reads the matrix on its OWN stream: the generator's kernels must be done)
    psi0 = np.random.default_rng(1).normal(size=(N, 1))
    engs = [("tree", E.HipEngine(None, aux["L"], psi0, np.zeros(G) + 0.5, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), tune=tune))]
    if voff:
        engs.append(("off:" + ",".join(voff), E.HipEngine(None, aux["L"], psi0, np.zeros(G) + 0.5, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32,
                                                         shape=(N, G), tune=tune, variant_off=voff)))
    best = {nm: 1e9 for nm, _ in engs}
    for rep in range(4):
        for nm, eng in engs:
            eng.iterate(50, eps[:100], want_elbo=False); eng.synchronize()
            t0 = time.perf_counter()
            eng.iterate(200, eps, want_elbo=False); eng.synchronize()
            best[nm] = min(best[nm], (time.perf_counter() - t0) / 200 * 1e6)
    for nm, eng in engs:
        info = eng.info()
        eng.set_profile(0x1F)
        eng.iterate(20, eps[:40], want_elbo=False); eng.kernel_times(reset=True)
        eng.iterate(100, eps[:200], want_elbo=False)
        kt = eng.kernel_times(reset=True)
        per = {k: (v[0] / max(v[1], 1) * 1e3) for k, v in kt.items()}
        print(f"N={N:6d} tiles={-(-N // 16):5d} ({-(-N // 16) / info['n_cu']:.2f}/CU) {nm:12s} block {info['fwd_block_cells']:3d} cells, balanced q {info.get('fwd_balanced', 0)}: "
              f"{best[nm]:6.1f} us/iter   events: fwd {per['fwd']:5.1f}  bwd {per['bwd']:5.1f}  other {per['other']:5.1f} us per launch", flush=True)
        eng.close()
    del Yd
