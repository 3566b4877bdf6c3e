"""Whole-loop parity at sizes the suite does not reach, for the secondary paths (mc_samples, covariates, K = 0 / 2, many clones, storage widths):
    python tools/big_parity.py N G C K P S [iters] [y_storage] [overflow_fraction]
ca_run on a fresh engine against oracle/c/clonealign_oracle.c (float64 arithmetic, float32 variables, all host threads) on ALL cells, same eps stream."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.inference import run_vi_loop  # noqa: E402
from clonealign_amd.rng import EpsStream  # noqa: E402
from oracle.c_port import CPortModel  # noqa: E402

N, G, C, K, P, S = (int(a) for a in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
ystore = sys.argv[8] if len(sys.argv) > 8 else "auto"
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
ovf = float(sys.argv[9]) if len(sys.argv) > 9 else 0.0   # fraction of the entries inflated past 255 (the 1-byte storage's overflow list at scale)
if ovf > 0:
    import torch
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(11)
    m = torch.rand(Yd.shape, device="cuda:0", generator=gen) < ovf
    Yd += m.to(torch.int32) * torch.randint(300, 3000, Yd.shape, device="cuda:0", generator=gen, dtype=torch.int32)
    del m
    torch.cuda.synchronize()
Y = Yd.cpu().numpy().astype(np.float64)
rng = np.random.default_rng(5)
psi0 = rng.normal(size=(N, max(K, 1)))[:, :K]
X = rng.normal(size=(N, P)) if P else None
from clonealign_amd.hostprep import mu_guess, safe_inverse_softplus  # noqa: E402
loc0 = safe_inverse_softplus(np.maximum(mu_guess(Y, True), 1e-6))
kw = dict(y_storage=ystore) if ystore != "auto" else {}
eng = HipEngine(Y, aux["L"], psi0, loc0, K, S=S, X=X, **kw)
info = eng.info()
t0 = time.time()
tr = np.asarray(eng.run(EpsStream(77, S, G), iters, 1e-12))
eng.synchronize()
t1 = time.time()
ora = CPortModel(Y, aux["L"], psi0, loc0, K, S=S, X=X, dtype="float32")
to = np.asarray(run_vi_loop(ora, EpsStream(77, S, G), iters, 1e-12))
se, so = eng.get_state(), ora.get_state()
perr = max(float(np.abs(se[n] - so[n]).max(initial=0) / max(np.abs(so[n]).max(initial=0), 1e-30)) for n in so)
pe, po = eng.get("clone_probs"), ora.get_params()["clone_probs"]
flips = int((np.where(pe.max(1) >= 0.95, pe.argmax(1), -1) != np.where(po.max(1) >= 0.95, po.argmax(1), -1)).sum())
print(f"N={N} G={G} C={C} K={K} P={P} S={S} overflow entries {info.get('n_overflow', '?')} storage {info['y_storage_name']} fwd_cell {info['fwd_cell']} bwd_mfma {info['bwd_mfma']} async_y {info.get('async_y')} y_ride {info.get('y_ride')}: "
      f"ELBO rel {np.abs(tr - to).max() / np.abs(to).max():.2e}, parameters {perr:.2e}, label flips {flips} of {N}; oracle {time.time() - t1:.0f} s")
