"""Randomised parity sweep (not part of the test suite): whole loops through the C ABI against the float64-accumulating
oracle on random shapes, storage widths, covariates, clone counts and environment variants.

    python tools/fuzz_parity.py [n_cases] [seed]

Prints one line per failure and a summary; exit code 1 if anything disagreed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from clonealign_amd.inference import run_vi_loop  # noqa: E402
from clonealign_amd.rng import EpsStream  # noqa: E402
from oracle.fused_numpy import FusedModel  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
VARIANTS = [{}, {}, {}, {}, {"CA_FWD_CELL": "0"}, {"CA_FWD_MFMA": "0"}, {"CA_BWD_MFMA": "0"}, {"CA_ASYNC_Y": "0"}, {"CA_PRE": "0"},
            {"CA_TAIL_FUSE": "0"}, {"CA_FC_TL": "4", "CA_FC_NBIG": "2"}, {"CA_PAIR_ELBO": "0"},
            {"CA_Y_MFMA1": "0"}, {"CA_Y_MFMA1": "0", "CA_RIDE_SEQ_ON": "1"}, {"CA_Y_MFMA1": "0", "CA_Y_RIDE": "0"}, {"CA_Y_RIDE": "0"},
            {"CA_UPDATE_MERGE": "0"}, {"CA_UPDATE_MERGE": "0", "CA_Y_MFMA1": "0"}, {"CA_S2_FUSE": "0"}]   # round 4: the two-launch update (the default is the merged launch)   # round 3: the vector stream and its riding forms
fails = 0
for it in range(n_cases):
    N = int(rng.integers(1, 900))
    G = int(rng.integers(1, 700))
    C = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 19))     # (9..16: the sixteen-column matrix-core form; 17, 18: plain passes)
    K = int(rng.choice([0, 1, 1, 1, 2]))
    P = int(rng.choice([0, 0, 0, 1])) if K > 0 else 0
    S = 1 if rng.random() < 0.8 else 2
    env = dict(VARIANTS[int(rng.integers(0, len(VARIANTS)))])
    kw = dict(N=N, G=G, C=C, K=K, S=S)
    if P:
        kw["P"] = P
    case = make_case(seed=int(rng.integers(0, 10**6)), **kw)
    if rng.random() < 0.25:         # copy numbers that are not integers (two-part split in the backward sweep)
        case["L"] = case["L"] + rng.random(case["L"].shape) * 0.9
    if rng.random() < 0.4:          # counts above 255: overflow list next to 1-byte storage
        idx = rng.integers(0, case["Y"].size, size=max(1, case["Y"].size // 3000))
        case["Y"].reshape(-1)[idx] += rng.integers(200, 2000, size=idx.size)
    # replay of one case: FUZZ_ONLY=<index> runs only that case (the random stream is consumed as in the full sweep); FUZZ_ENV="A=1,B=0" adds switches
    only = os.environ.get("FUZZ_ONLY")
    if only is not None and int(only) != it:
        rng.integers(1, 6)
        continue
    if only is not None:
        env.update(kv.split("=") for kv in os.environ.get("FUZZ_ENV", "").split(",") if kv)
        print("replaying case", it, kw, env)
    env = dict(env, CLONEALIGN_DEBUG_ENV="1")      # the library reads CA_* from the environment only in this debug mode
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    eng = None
    try:
        eng, ora = HipEngine(**case), FusedModel(**case, dtype="float32")
        n_iter = int(rng.integers(1, 6))
        tr = np.asarray(eng.run(EpsStream(3, S, G), n_iter, 1e-12))
        to = np.asarray(run_vi_loop(ora, EpsStream(3, S, G), n_iter, 1e-12))
        ok = tr.shape == to.shape and np.all(np.isfinite(tr)) and np.abs(tr - to).max() <= 1e-4 * np.abs(to).max()
        eps = np.stack([eps_for(S, G, 50 + i) for i in range(5)])
        fe = eng.final_elbo(eps, 5)
        fo = np.array([ora.elbo(e) for e in eps])
        why = [] if ok else ["trace"]
        if np.abs(fe - fo).max() > 1e-4 * np.abs(fo).max():
            why.append("final elbo %.2e" % float(np.abs(fe - fo).max() / np.abs(fo).max()))
        pe, po = eng.get_state(), {n: getattr(ora, n) for n in ora.VAR_NAMES}
        for n in ora.VAR_NAMES:
            a, b = np.asarray(pe[n], float), np.asarray(po[n], float)
            # loose on purpose: Adam normalises every gradient, so float32-level differences in a small gradient become 1e-3-level
            # differences of its variable after a few steps (seen: W 2.8e-4 of 0.2); the ELBO trace is the tight check
            if a.size and np.abs(a - b).max() > 5e-3 * max(np.abs(b).max(), 1e-2):
                why.append("%s %.2e of %.2e" % (n, float(np.abs(a - b).max()), float(np.abs(b).max())))
        if only is not None:
            print("trace engine", tr.tolist(), "oracle", to.tolist(), "rel", float(np.abs(tr - to).max() / np.abs(to).max()))
        if why:
            fails += 1
            print("FAIL case", it, kw, env, "iters", n_iter, "trace diff", float(np.abs(tr - to).max() / np.abs(to).max()), "|", "; ".join(why))
    except Exception as exc:   # noqa: BLE001
        fails += 1
        print("ERROR", kw, env, repr(exc))
    finally:
        if eng is not None:
            eng.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
print(f"{n_cases - fails} of {n_cases} cases agree with the oracle")
sys.exit(1 if fails else 0)
