#!/usr/bin/env python3
"""Generate golden input/output vectors with the LITERAL float64 oracle (autodiff restatement of
R/inference-tflow.R:240-346).  The reference itself cannot run here (no R/TensorFlow), so these
are self-generated goldens: they pin fused-oracle and HIP-engine results against the op-by-op
restatement under an explicit eps stream.  Re-run only when the model restatement changes.

  cfg1      : example_sce (200 x 100 x 3), K=1, 200 iterations + 20 final ELBOs  (BASELINE configs[0])
  tiny_k0   : 40 x 25 x 3, K=0, 12 iterations
  tiny_full : 36 x 20 x 4, K=2, P=1, S=2, allele-style extra term, 12 iterations
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from clonealign_amd import hostprep  # noqa: E402
from clonealign_amd.api import clone_assignment  # noqa: E402
from clonealign_amd.rng import EpsStream  # noqa: E402
from oracle.literal_torch import LiteralModel  # noqa: E402
from tests._cases import make_case  # noqa: E402


def run(case, n_iter, n_final, seed):
    m = LiteralModel(**case, dtype="float64")
    es = EpsStream(seed, m.S, m.G)
    eps = es.block(2 + 2 * n_iter + n_final)
    m.gamma_init(eps[0])
    trace = [m.elbo(eps[1])]
    for i in range(1, n_iter + 1):
        m.step(eps[2 * i])
        trace.append(m.elbo(eps[2 * i + 1]))
    final = [m.elbo(eps[2 + 2 * n_iter + j]) for j in range(n_final)]
    out = dict(eps=eps, elbo_trace=np.array(trace), final_elbos=np.array(final))
    for k, v in m.get_params().items():
        out["param_" + k] = v
    for k, v in m.get_state().items():
        out["state_" + k] = v
    return out


def main():
    d = np.load(os.path.join(HERE, "example_sce.npz"))
    Y, L = d["Y"].astype(np.float64), d["L"].astype(np.float64)
    rng = np.random.default_rng(2024)
    noise = rng.normal(0, 0.05, size=(1, 200)).T
    psi0 = hostprep.pca_init(Y, 1, noise)
    loc0 = hostprep.safe_inverse_softplus(hostprep.mu_guess(Y, True))
    case = dict(Y=Y, L=hostprep.saturate(L, 6), psi0=psi0, loc0=loc0, K=1, S=1)
    out = run(case, 200, 20, seed=77001)
    out.update(psi0=psi0, loc0=loc0)
    out["clone"] = clone_assignment(out["param_clone_probs"], list(d["clones"])).astype(str)
    np.savez_compressed(os.path.join(HERE, "golden_cfg1.npz"), **out)
    print("cfg1", out["elbo_trace"][[0, 1, 10, 200]], out["final_elbos"].mean(),
          dict(zip(*np.unique(out["clone"], return_counts=True))))
    for name, kw, seed in (("tiny_k0", dict(N=40, G=25, C=3, K=0), 77002),
                           ("tiny_full", dict(N=36, G=20, C=4, K=2, P=1, S=2, extra=True), 77003)):
        case = make_case(seed=seed, **kw)
        out = run(case, 12, 3, seed=seed)
        out.update({"in_" + k: v for k, v in case.items() if isinstance(v, np.ndarray)})
        out["in_K"], out["in_S"] = case["K"], case["S"]
        np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), **out)
        print(name, out["elbo_trace"][[0, 1, 12]])


if __name__ == "__main__":
    main()
