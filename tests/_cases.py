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
