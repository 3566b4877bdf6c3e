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
