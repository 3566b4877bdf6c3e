"""Randomised parity sweep at shard-to-bench sizes (not part of the test suite): ca_run + final ELBOs through the fused loop
against the C/OpenMP float64 oracle driven call by call.  Ragged cell counts from 20k to 90k exercise the two block sizes of
the forward sweep, several row groups of the Y stream and the resident-round decomposition of the backward sweep.

    python tools/fuzz_large.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.inference import run_vi_loop  # noqa: E402
from clonealign_amd.rng import EpsStream  # noqa: E402
from oracle.c_port import CPortModel  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

_a = [a for a in sys.argv[1:] if not a.startswith("--")]
WIDE = "--wide" in sys.argv
n_cases = int(_a[0]) if len(_a) > 0 else 8
rng = np.random.default_rng(int(_a[1]) if len(_a) > 1 else 2)
fails = 0
for it in range(n_cases):
    N = int(rng.integers(20_000, 90_000))
    G = int(rng.integers(200, 1500))
    C = int(rng.integers(2, 9))
    K = int(rng.choice([1, 1, 2]))
    P = int(rng.choice([0, 0, 1]))
    S = 1
    env = {}
    if WIDE:   # round 5: every path at sizes ABOVE the side-stream threshold (4e7 counts): many clones, K = 0, covariates, mc_samples up to 3, storage widths, variants
        G = int(rng.integers(600, 5200))
        N = int(rng.integers(max(8000, int(4.2e7 / G)), max(8001, int(1.6e8 / G))))
        C = int(rng.choice([2, 3, 5, 8, 8, 11, 16, 17, 20]))
        K = int(rng.choice([0, 1, 1, 1, 2]))
        P = int(rng.choice([0, 0, 1, 2])) if K > 0 else 0
        S = int(rng.choice([1, 1, 1, 2, 3]))
        r = rng.random()
        if r < 0.15:
            env["y_storage"] = "u16"
        elif r < 0.25:
            env["y_storage"] = "f32"
        voff = [(), (), (), ("y_ride",), ("y_mfma1",), ("update_merge",), ("run_gate",), ("fwd_cell",), ("bwd_mfma",), ("tail_fuse",), ("pre",), ("fused",)][int(rng.integers(0, 12))]
        if voff:
            env["variant_off"] = voff
    kw = dict(N=N, G=G, C=C, K=K)
    if P:
        kw["P"] = P
    if S > 1:
        kw["S"] = S
    case = make_case(seed=int(rng.integers(0, 10**6)), **kw)
    if rng.random() < 0.5:
        idx = rng.integers(0, case["Y"].size, size=case["Y"].size // 20000)
        case["Y"].reshape(-1)[idx] += rng.integers(200, 2000, size=idx.size)
    eng = ora = None
    try:
        eng = HipEngine(**case, **env)
        ora = CPortModel(case["Y"], case["L"], case["psi0"], case["loc0"], K, S=S, X=case["X"], dtype="float32")
        n_iter = int(rng.integers(2, 5))
        tr = np.asarray(eng.run(EpsStream(4, S, G), n_iter, 1e-12))
        to = np.asarray(run_vi_loop(ora, EpsStream(4, S, G), n_iter, 1e-12))
        eps = np.stack([eps_for(S, G, 60 + i) for i in range(3)])
        fe, fo = eng.final_elbo(eps, 3), np.array([ora.elbo(e) for e in eps])
        info = eng.info()
        d1, d2 = float(np.abs(tr - to).max() / np.abs(to).max()), float(np.abs(fe - fo).max() / np.abs(fo).max())
        ok = tr.shape == to.shape and d1 <= 1e-5 and d2 <= 1e-5
        print("ok  " if ok else "FAIL", kw, env, "iters", n_iter, "storage", info["y_storage_name"], "cell kernel", info["fwd_cell"], "trace %.1e final %.1e" % (d1, d2), flush=True)
        fails += 0 if ok else 1
    except Exception as exc:   # noqa: BLE001
        fails += 1
        print("ERROR", kw, env, repr(exc))
    finally:
        if eng is not None:
            eng.close()
        if ora is not None:
            ora.close()
print(f"{n_cases - fails} of {n_cases} cases agree with the C oracle")
sys.exit(1 if fails else 0)
