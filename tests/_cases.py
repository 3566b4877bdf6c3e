"""Seeded synthetic problems shared by the oracle and GPU parity tests."""
import numpy as np


def make_case(N, G, C, K, P=0, S=1, extra=False, seed=0, scale=0.5):
    rng = np.random.default_rng(seed)
    L = rng.integers(1, 5, size=(G, C)).astype(np.float64)
    mu = rng.lognormal(0, 1, G)
    z = rng.integers(0, C, N)
    Y = rng.poisson(mu[None, :] * L[:, z].T * scale).astype(np.float64)
    Y[:, 0] += 1                      # no empty cells
    Y[0, :] += 1                      # no empty genes
    psi0 = rng.normal(size=(N, K))
    loc0 = rng.normal(size=G) + 1.0
    X = rng.normal(size=(N, P)) if P else None
    ex = rng.normal(size=(N, C)) if extra else None
    return dict(Y=Y, L=L, psi0=psi0, loc0=loc0, K=K, S=S, X=X, extra_loglik=ex)


def perturbed_state(model_shapes, seed=1, amp=0.3):
    """Generic (non-initial) variable values so every gradient path is exercised."""
    rng = np.random.default_rng(seed)
    st = {}
    for n, sh in model_shapes.items():
        v = rng.normal(size=sh) * amp
        if n == "loc":
            v = v + 1.0
        st[n] = v.astype(np.float32).astype(np.float64)   # exactly representable in the engine's float32
    return st


def eps_for(S, G, seed):
    return np.random.default_rng(seed).normal(size=(S, G)).astype(np.float32)


def label_flips(p_test, p_ref, threshold=0.95, margin=1e-3):
    """clone_assignment of R/inference-tflow.R:22-29 on two [N,C] posterior matrices: label = argmax if max >= 0.95 else
    "unassigned" (-1 here).  Returns (cells whose label differs, those of them whose reference maximum is NOT within
    ``margin`` of the threshold).  A label can only flip where the maximum crosses 0.95 (above it the runner-up is <= 0.05),
    so the second number must be 0 and the first is the count the tests print and bound."""
    lt = np.where(p_test.max(1) >= threshold, p_test.argmax(1), -1)
    lr = np.where(p_ref.max(1) >= threshold, p_ref.argmax(1), -1)
    diff = lt != lr
    near = np.abs(p_ref.max(1) - threshold) <= margin
    return int(diff.sum()), int((diff & ~near).sum())


def record_labels(tag, p_test, p_ref, threshold=0.95):
    """label_flips plus a kept observation: one JSON line per call in gpurun_out/labels.jsonl (copied to profiles/ per round) with
    the number of cells, the flips, the smallest distance of any reference cell's max-gamma from the threshold, and for every
    flipped cell its index and max-gamma on both sides -- the evidence behind each test's bound (north_star: labels exactly)."""
    import json
    import os
    flips, far = label_flips(p_test, p_ref, threshold)
    mt, mr = p_test.max(1), p_ref.max(1)
    lt = np.where(mt >= threshold, p_test.argmax(1), -1)
    lr = np.where(mr >= threshold, p_ref.argmax(1), -1)
    idx = np.nonzero(lt != lr)[0]
    row = dict(tag=tag, cells=int(p_ref.shape[0]), clones=int(p_ref.shape[1]), flips=flips, flips_outside_margin=far,
               unassigned_ref=int((lr < 0).sum()), min_margin_ref=float(np.abs(mr - threshold).min()),
               cells_within_1em4_of_threshold=int((np.abs(mr - threshold) <= 1e-4).sum()),
               max_abs_gamma_diff=float(np.abs(p_test - p_ref).max()),
               flipped=[dict(cell=int(i), max_gamma_engine=float(mt[i]), max_gamma_ref=float(mr[i])) for i in idx[:20]])
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "labels.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    print(f"labels {tag}: {flips} of {row['cells']} differ ({far} outside the 1e-3 margin); closest reference cell to 0.95: {row['min_margin_ref']:.2e}")
    return flips, far


_CHILD_KEYS = ("[rank", "bench.py", "dist_check", "clonealign", "EngineError", "CA_ERR", "Error:", "SystemExit", "refusing", "timed out", "assert")


def child_report(r, tail=1500):
    """What a failed child launch (torchrun / bench.py / tools/*.py) said, for an assertion message: the lines the RANKS wrote
    (torchrun's own elastic traceback fills the last kilobytes of stderr and says nothing about the cause), then the tails."""
    err, out = r.stderr or "", r.stdout or ""
    said = [l for l in (err + "\n" + out).splitlines()
            if any(k in l for k in _CHILD_KEYS) and "torch/distributed" not in l and "elastic" not in l]
    return ("\n--- rank messages ---\n" + "\n".join(said[-60:]) + f"\n--- stdout tail ---\n{out[-tail:]}\n--- stderr tail ---\n{err[-tail:]}")
