"""us per ca_iterate iteration with the contraction in its series form (CA_VARX_SERIES) against the matrix-core sweeps, per kernel class:
    python tools/series_time.py [cells genes clones] ..."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402,F401  (first: its HIP runtime)
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.hostprep import safe_inverse_softplus  # noqa: E402

shapes = [(100_000, 5_000, 8), (10_000, 2_000, 4), (50_000, 3_000, 6), (12_500, 5_000, 8), (25_000, 5_000, 8)]
if len(sys.argv) > 3:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for N, G, C in shapes:
    Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
    rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
    col = torch.zeros(G, dtype=torch.float64, device="cuda:0")
    for b0 in range(0, N, 8192):
        col += (Yd[b0:b0 + 8192].to(torch.float64) / rm[b0:b0 + 8192]).sum(0)
    loc0 = safe_inverse_softplus(np.maximum(col.cpu().numpy() / N, 1e-6))
    rng = np.random.default_rng(1)
    psi0 = rng.normal(size=(N, 1))
    torch.cuda.synchronize()
    steps = 100
    eps = rng.normal(size=(2 * steps + 1, 1, G)).astype(np.float32)
    eps[-1] = eps[0]
    res = {}
    import os
    # SERIES_SWEEP=1: the lab knobs (ca_options.reserved): cell blocks per CU; side 1 = the count-matrix stream on the side stream, 3 = the stream's finisher as a launch of its own,
    # 4 = the moment launches on the side stream, 5 = k_poly_xmax as a launch of its own
    knobs = [("series", {})] + [(f"series b{b} side{sd}", {"series_blocks": b, "series_side": sd}) for b, sd in ((2, 5), (2, 4), (2, 3), (2, 0))] if os.environ.get("SERIES_SWEEP") else [("series", {})]
    for name, von, tn in [("sweeps", (), {})] + [(n_, ("series",), t_) for n_, t_ in knobs]:
        voff = ("series",) if name == "sweeps" else ()
        eng = HipEngine(None, aux["L"], psi0, loc0, 1, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), variant_on=von, variant_off=voff, profile=0, tune=tn)
        eng.gamma_init(eps[0])
        eng.iterate(steps, eps); eng.iterate(steps, eps)
        eng.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            last = eng.iterate(steps, eps)
            eng.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e6)
        eng.set_profile(0x1F)
        eng.iterate(20, eps[:41])
        kt = eng.kernel_times(reset=True)
        res[name] = (min(ts), last, {k: round(v[0] / 20 * 1e3, 1) for k, v in kt.items()}, {k: v[1] // 20 for k, v in kt.items()})
        eng.close()
    s, w = res["series"], res["sweeps"]
    print(f"{N} x {G} x {C}: sweeps {w[0]:.1f} us/iter, series {s[0]:.1f} us/iter ({w[0] / s[0]:.2f}x); last ELBO rel diff {abs(s[1] - w[1]) / abs(w[1]):.2e}")
    for nm, r in res.items():
        print(f"   {nm:18s} {r[0]:7.1f} us/iter; per class us {r[2]}", flush=True)
    del Yd
    torch.cuda.empty_cache()
