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

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_cases = int(_args[0]) if len(_args) > 0 else 60
rng = np.random.default_rng(int(_args[1]) if len(_args) > 1 else 7)
# engine variants as ca_options (the release library reads no switches from the environment since round 5): (variant_off, variant_on, tune)
VARIANTS = [((), (), {})] * 4 + [(("fwd_cell",), (), {}), (("fwd_mfma",), (), {}), (("bwd_mfma",), (), {}), (("async_y",), (), {}), (("pre",), (), {}),
            (("tail_fuse",), (), {}), ((), (), {"fc_tl": 4, "fc_nbig": 2}), (("pair_elbo",), (), {}),
            (("y_mfma1",), (), {}), (("y_mfma1", "y_ride"), (), {}), (("y_ride",), (), {}),
            (("update_merge",), (), {}), (("update_merge", "y_mfma1"), (), {}), (("s2_fuse",), (), {}), (("run_gate",), (), {}), (("fwd_bal",), (), {}), (("bwd_tl3",), (), {})]
BAL = "--bal" in sys.argv   # shapes of the balanced eight-wave forward sweep (4096+ cells, 3072+ genes, K = 1, up to eight clones) against the C oracle
fails = 0
for it in range(n_cases):
    N = int(rng.integers(1, 900))
    G = int(rng.integers(1, 700))
    C = int(rng.integers(1, 9)) if rng.random() < 0.7 else int(rng.integers(9, 19))     # (9..16: the sixteen-column matrix-core form; 17, 18: plain passes)
    K = int(rng.choice([0, 1, 1, 1, 2]))
    P = int(rng.choice([0, 0, 0, 1])) if K > 0 else 0
    S = 1 if rng.random() < 0.8 else 2
    if "--r6" in sys.argv:   # round 6's argument space: D = K + P up to 5, up to 36 clones, up to four MC samples (matrix-core plain passes, D = 3 / 4 sweeps)
        C = int(rng.integers(1, 37))
        K = int(rng.choice([0, 1, 1, 2, 3, 4]))
        P = int(rng.choice([0, 0, 1, 2, 3])) if K > 0 else 0
        if K + P > 5:
            P = 5 - K
        S = int(rng.choice([1, 1, 2, 3, 4]))
        N = int(rng.integers(1, 2500))
        G = max(G, 2)   # (one gene: the likelihood does not depend on the parameters, every gradient is the sign of rounding noise -- profiles/r06_fuzz.txt)
    if BAL:
        N, G, C, K, P, S = int(rng.integers(4096, 28672)), int(rng.integers(3072, 3400)), int(rng.integers(2, 9)), 1, 0, 1   # (up to six whole tiles per CU: the balanced range)
    voff, von, tune = VARIANTS[int(rng.integers(0, len(VARIANTS)))] if not BAL else ((), (), {})
    env = {"variant_off": voff, "variant_on": von, "tune": tune}
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
        print("replaying case", it, kw, env)
    eng = ora = None
    try:
        eng = HipEngine(**case, **env)
        if BAL:
            from oracle.c_port import CPortModel
            assert eng.info()["fwd_balanced"] >= 1, eng.info()
            ora = CPortModel(case["Y"], case["L"], case["psi0"], case["loc0"], 1, dtype="float32")
        else:
            ora = FusedModel(**case, dtype="float32")
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
        pe, po = eng.get_state(), (ora.get_state() if BAL else {n: getattr(ora, n) for n in ora.VAR_NAMES})
        for n in po:
            a, b = np.asarray(pe[n], float), np.asarray(po[n], float)
            # loose on purpose: Adam normalises every gradient, so float32-level differences in a small gradient become 1e-3-level
            # differences of its variable after a few steps (seen: W 2.8e-4 of 0.2); the ELBO trace is the tight check
            if a.size and np.abs(a - b).max() > 5e-3 * max(np.abs(b).max(), 1e-2):
                why.append("%s %.2e of %.2e" % (n, float(np.abs(a - b).max()), float(np.abs(b).max())))
        if only is not None:
            print("trace engine", tr.tolist(), "oracle", to.tolist(), "rel", float(np.abs(tr - to).max() / np.abs(to).max()))
            # where a variable disagrees, and how small the oracle's gradient was there at each step (Adam turns the SIGN of a gradient of rounding-noise
            # size into a step of lr): a fresh oracle replays the loop and records |g| at the worst element against the variable's median |g|
            bad = [n for n in po if np.asarray(po[n]).size and np.abs(np.asarray(pe[n], float) - np.asarray(po[n], float)).max() > 5e-3 * max(np.abs(np.asarray(po[n], float)).max(), 1e-2)]
            if bad and not BAL:
                o2 = FusedModel(**case, dtype="float32")
                es = EpsStream(3, S, G)
                o2.gamma_init(es.next()); o2.elbo(es.next())     # (run_vi_loop's order of draws)
                for step in range(n_iter):
                    e = es.next()
                    g, _ = o2.gradients(e)
                    for n in bad:
                        d = np.abs(np.asarray(pe[n], float) - np.asarray(po[n], float)).reshape(-1)
                        j = int(np.argmax(d))
                        gv = np.abs(np.asarray(g[n], float)).reshape(-1)
                        print(f"  step {step}: {n}[{j}] final diff {d[j]:.3e}; oracle |g| there {gv[j]:.3e}, median |g| of {n} {np.median(gv):.3e}, elements with |g| < 1e-6 median: {int((gv < 1e-6 * np.median(gv)).sum())}")
                    o2.step(e); o2.elbo(es.next())
        if why:
            fails += 1
            print("FAIL case", it, kw, env, "iters", n_iter, "trace diff", float(np.abs(tr - to).max() / np.abs(to).max()), "|", "; ".join(why))
    except Exception as exc:   # noqa: BLE001
        fails += 1
        print("ERROR", kw, env, repr(exc))
    finally:
        if eng is not None:
            eng.close()
        if BAL and ora is not None:
            ora.close()
print(f"{n_cases - fails} of {n_cases} cases agree with the oracle")
sys.exit(1 if fails else 0)
