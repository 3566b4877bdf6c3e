"""The C/OpenMP oracle port (CPU baseline + large-size checker) against the numpy oracle and the goldens."""
import numpy as np
import pytest

from oracle.c_port import CPortModel
from oracle.fused_numpy import FusedModel
from tests import _golden
from tests._cases import eps_for, make_case, perturbed_state


@pytest.mark.parametrize("kw", [dict(N=40, G=30, C=3, K=1), dict(N=33, G=21, C=4, K=0),
                                dict(N=30, G=18, C=4, K=2, P=1, S=2, extra=True), dict(N=30, G=18, C=9, K=0, P=1)])
def test_c_port_matches_numpy_oracle(kw):
    case = make_case(seed=8, **kw)
    a, b = CPortModel(**case), FusedModel(**case)
    st = perturbed_state({n: getattr(b, n).shape for n in b.VAR_NAMES})
    for n, v in st.items():
        setattr(b, n, v.copy())
        a.set(n, v)
    eps = eps_for(b.S, b.G, 2)
    np.testing.assert_allclose(a.elbo_terms(eps), b.elbo_terms(eps), rtol=1e-11)
    ga, _ = a.gradients(eps)
    gb, _ = b.gradients(eps)
    for n in b.VAR_NAMES:
        np.testing.assert_allclose(ga[n], gb[n], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(gb[n]).max(initial=0)))
    a.gamma_init(eps)
    b.gamma_init(eps)
    np.testing.assert_allclose(a.get("gamma_logits"), b.gamma_logits, rtol=1e-10, atol=1e-9)
    a.close()


def test_c_port_replays_cfg1_golden():
    g = _golden.load("cfg1")
    m = CPortModel(**_golden.case_of("cfg1", g))
    trace, final = _golden.replay(m, g, 200)
    np.testing.assert_allclose(trace, g["elbo_trace"], rtol=1e-7)
    np.testing.assert_allclose(final, g["final_elbos"], rtol=1e-7)
    for k, v in m.get_params().items():
        np.testing.assert_allclose(v, g["param_" + k], rtol=1e-5, atol=1e-8)
    m.close()


@pytest.mark.parametrize("shape", [(600, 300, 5, 1, 0), (500, 200, 3, 2, 1), (400, 150, 4, 0, 1)])
def test_simd_baseline_port_follows_the_float64_port(shape):
    """The float32 SIMD port bench.py times as the CPU baseline (oracle/c/clonealign_simd.c) against the scalar float64 port:
    ELBO terms before and after gamma init, then seven Adam iterations (float32 accumulation over the genes: 1e-5 on the ELBO;
    the q(z) logits are O(1e3) numbers updated from differences of such numbers: 5e-3 of their range).  Loaded AFTER the float64
    port on purpose: the SIMD library must not depend on flush-to-zero being set by whoever was loaded first."""
    import synth_data as synth
    from oracle.c_port import CSimdModel
    N, G, C, K, P = shape
    d = synth.make_problem(N, G, C, seed=3, median_s=500)
    Y, L = d["Y"].astype(np.float64), d["L"]
    psi0, loc0 = synth.cheap_init(Y, K=max(K, 1))
    X = np.random.default_rng(0).normal(size=(N, P)) if P else None
    a = CPortModel(Y, L, psi0[:, :K], loc0, K, 1, X=X, dtype="float32")
    b = CSimdModel(Y, L, psi0[:, :K], loc0, K, 1, X=X)
    e = lambda i: np.random.default_rng(100 + i).normal(size=G).astype(np.float32)  # noqa: E731
    np.testing.assert_allclose(b.elbo_terms(e(50)), a.elbo_terms(e(50)), rtol=2e-6)
    a.gamma_init(e(0)); b.gamma_init(e(0))
    np.testing.assert_allclose(b.elbo_terms(e(51)), a.elbo_terms(e(51)), rtol=2e-6, atol=1e-3)
    for i in range(1, 8):
        a.step(e(2 * i)); b.step(e(2 * i))
        ea, eb = a.elbo(e(2 * i + 1)), b.elbo(e(2 * i + 1))
        assert np.isfinite(eb) and abs(ea - eb) <= 1e-5 * abs(ea), (i, ea, eb)
    sa, sb = a.get_state(), b.get_state()
    for n in a.VAR_NAMES:
        if sa[n].size:
            err = np.abs(sa[n] - sb[n]).max() / max(np.abs(sa[n]).max(), 1e-30)
            assert err < (5e-3 if n in ("gamma_logits", "alpha_unconstr") else 1e-4), (n, err)
    a.close(); b.close()
