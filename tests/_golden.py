"""Replay helpers for the committed golden vectors (tests/golden/*.npz)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, f"golden_{name}.npz"), allow_pickle=False))


def example():
    d = np.load(os.path.join(GOLDEN, "example_sce.npz"))
    return d["Y"].astype(np.float64), d["L"].astype(np.float64), [str(c) for c in d["clones"]], \
        [str(g) for g in d["genes"]], [str(c) for c in d["cells"]]


def case_of(name, g):
    if name == "cfg1":
        Y, L, *_ = example()
        return dict(Y=Y, L=L, psi0=g["psi0"], loc0=g["loc0"], K=1, S=1)
    return dict(Y=g["in_Y"], L=g["in_L"], psi0=g["in_psi0"], loc0=g["in_loc0"], K=int(g["in_K"]), S=int(g["in_S"]),
                X=g.get("in_X"), extra_loglik=g.get("in_extra_loglik"))


def replay(model, g, n_iter):
    """Drive any engine through the golden eps stream exactly as the goldens were made."""
    eps = g["eps"]
    model.gamma_init(eps[0])
    trace = [model.elbo(eps[1])]
    for i in range(1, n_iter + 1):
        model.step(eps[2 * i])
        trace.append(model.elbo(eps[2 * i + 1]))
    n_final = len(g["final_elbos"])
    final = [model.elbo(eps[2 + 2 * n_iter + j]) for j in range(n_final)]
    return np.array(trace), np.array(final)
