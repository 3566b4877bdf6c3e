"""The oracle checked against itself three ways and against the committed goldens (CPU only).

literal (autodiff, materialising, follows R/inference-tflow.R:240-346 op by op)
   <-> fused (hand gradients, what the HIP engine mirrors) <-> central finite differences.
"""
import numpy as np
import pytest

from oracle.fused_numpy import FusedModel
from oracle.literal_torch import LiteralModel
from tests import _golden
from tests._cases import eps_for, make_case, perturbed_state

CASES = {
    "k1": dict(N=30, G=20, C=3, K=1),
    "k0": dict(N=30, G=20, C=3, K=0),
    "k2p1s2x": dict(N=25, G=17, C=4, K=2, P=1, S=2, extra=True),
    "k0p1s2": dict(N=25, G=17, C=4, K=0, P=1, S=2),
    "c9": dict(N=20, G=15, C=9, K=1),
}


def _pair(name):
    import torch
    case = make_case(seed=3, **CASES[name])
    a, b = LiteralModel(**case), FusedModel(**case)
    st = perturbed_state({n: getattr(b, n).shape for n in b.VAR_NAMES})
    for n, v in st.items():
        setattr(b, n, v.copy())
        setattr(a, n, torch.tensor(v))
    return a, b


@pytest.mark.parametrize("name", list(CASES))
def test_fused_matches_literal(name):
    a, b = _pair(name)
    eps = eps_for(b.S, b.G, 5)
    np.testing.assert_allclose(b.elbo_terms(eps), a.elbo_terms(eps), rtol=1e-11)
    ga, ea = a.gradients(eps)
    gb, eb = b.gradients(eps)
    assert abs(ea - eb) < 1e-9 * abs(ea)
    for n in b.VAR_NAMES:
        np.testing.assert_allclose(gb[n], ga[n].numpy(), rtol=1e-9, atol=1e-9 * max(1.0, np.abs(gb[n]).max(initial=0)))
    a.gamma_init(eps)
    b.gamma_init(eps)
    np.testing.assert_allclose(b.gamma_logits, a.gamma_logits.numpy(), rtol=1e-10, atol=1e-9)
    for i in range(4):
        e = eps_for(b.S, b.G, 50 + i)
        a.step(e)
        b.step(e)
    sa, sb = a.get_state(), b.get_state()
    for n in sa:
        np.testing.assert_allclose(sb[n], sa[n], rtol=1e-6, atol=1e-7)


def test_hand_gradients_match_finite_differences():
    _, b = _pair("k2p1s2x")
    eps = eps_for(b.S, b.G, 9)
    g, _ = b.gradients(eps)
    rng = np.random.default_rng(0)
    for n in b.VAR_NAMES:
        v = getattr(b, n)
        if v.size == 0:
            continue
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in v.shape)
            old = v[idx]
            h = 1e-5
            v[idx] = old + h
            ep = b.elbo(eps)
            v[idx] = old - h
            em = b.elbo(eps)
            v[idx] = old
            fd = (ep - em) / (2 * h)
            assert abs(fd - g[n][idx]) < 1e-5 * max(1.0, abs(fd)), (n, idx, fd, g[n][idx])


@pytest.mark.parametrize("name,n_iter", [("cfg1", 200), ("tiny_k0", 12), ("tiny_full", 12)])
def test_fused_oracle_replays_goldens(name, n_iter):
    g = _golden.load(name)
    m = FusedModel(**_golden.case_of(name, g))
    trace, final = _golden.replay(m, g, n_iter)
    np.testing.assert_allclose(trace, g["elbo_trace"], rtol=1e-7)
    np.testing.assert_allclose(final, g["final_elbos"], rtol=1e-7)
    p = m.get_params()
    for k, v in p.items():
        np.testing.assert_allclose(v, g["param_" + k], rtol=1e-5, atol=1e-8)


def test_float32_variables_stay_within_1e4_of_float64_on_cfg1():
    """What holding the variables in float32 (TensorFlow's default dtype, and the engine's) costs."""
    g = _golden.load("cfg1")
    m = FusedModel(**_golden.case_of("cfg1", g), dtype="float32")
    trace, final = _golden.replay(m, g, 200)
    assert np.abs(trace - g["elbo_trace"]).max() <= 1e-4 * np.abs(g["elbo_trace"]).max()
    p = m.get_params()
    for k in ("mu", "alpha", "psi", "W", "chi", "clone_probs"):
        assert np.abs(p[k] - g["param_" + k]).max() <= 1e-4 * np.abs(g["param_" + k]).max(), k


def test_literal_float32_graph_runs():
    case = make_case(seed=4, **CASES["k1"])
    a = LiteralModel(**case, dtype="float32")
    e = eps_for(1, a.G, 1)
    a.gamma_init(e)
    a.step(e)
    assert np.isfinite(a.elbo(e))
