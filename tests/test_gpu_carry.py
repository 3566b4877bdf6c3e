"""ca_iterate, ABI 6: a call given one draw more than it consumes makes the forward half of the NEXT call's first train pass in its last sweep;
back-to-back calls of n iterations then run n sweeps each instead of n + 1 (the driver's 20-step command paid 21 sweeps).
The fit must not depend on how the iterations are cut into calls."""
import numpy as np
import pytest

from tests._cases import eps_for, make_case

pytestmark = pytest.mark.gpu

CASES = {"k1": dict(N=1301, G=700, C=8, K=1), "k1p1": dict(N=777, G=333, C=5, K=1, P=1), "c12": dict(N=900, G=260, C=12, K=1),
         "k2": dict(N=515, G=97, C=2, K=2), "s2": dict(N=640, G=300, C=6, K=1, S=2)}


def _fwd_launches(eng):
    return eng.kernel_times()["fwd"][1]


@pytest.mark.parametrize("name", list(CASES))
def test_calls_that_carry_the_next_forward_half_equal_one_long_call(name):
    from clonealign_amd.engine import HipEngine
    case = make_case(seed=19, **CASES[name])
    G, S = case["Y"].shape[1], case["S"]
    n, calls = 4, 3
    eps = np.stack([eps_for(S, G, 100 + i) for i in range(2 * n * calls + 1)])
    one = HipEngine(**case, profile=True)
    one.gamma_init(eps_for(S, G, 0))
    e_one = one.iterate(n * calls, eps[:2 * n * calls])
    st_one, l_one = one.get_state(), _fwd_launches(one)
    one.close()
    # the same iterations as three calls, each handed the next call's first draw as well
    cut = HipEngine(**case, profile=True)
    cut.gamma_init(eps_for(S, G, 0))
    for c in range(calls):
        e_cut = cut.iterate(n, eps[2 * n * c: 2 * n * (c + 1) + 1])
    st_cut, l_cut = cut.get_state(), _fwd_launches(cut)
    # ... and as three calls that carry nothing (2 n draws each): the old behaviour, one duplicate sweep per call
    plain = HipEngine(**case, profile=True)
    plain.gamma_init(eps_for(S, G, 0))
    for c in range(calls):
        e_plain = plain.iterate(n, eps[2 * n * c: 2 * n * (c + 1)])
    st_plain, l_plain = plain.get_state(), _fwd_launches(plain)
    fused = cut.info()["fused_sweep"] and S == 1
    cut.close(); plain.close()
    if fused:
        # the same sweeps pair the same draws as in the long call: bit for bit, and no sweep more than the long call makes (+ the last call's look ahead)
        assert e_cut == e_one
        for k, v in st_one.items():
            assert np.array_equal(st_cut[k], v), k
        assert l_cut <= l_one + 1 and l_plain >= l_cut + (calls - 1), (l_one, l_cut, l_plain)
    assert abs(e_plain - e_one) <= 1e-6 * abs(e_one) and abs(e_cut - e_one) <= 1e-6 * abs(e_one)
    for k, v in st_one.items():
        assert np.abs(st_plain[k] - v).max(initial=0) <= 2e-5 * max(np.abs(v).max(initial=0), 1e-30), k


def test_a_carried_half_is_dropped_when_the_next_call_brings_another_draw_or_something_else_ran():
    from clonealign_amd.engine import HipEngine
    case = make_case(seed=3, N=1301, G=700, C=8, K=1)
    G = 700
    eps = np.stack([eps_for(1, G, 300 + i) for i in range(40)])
    ref = HipEngine(**case)
    ref.gamma_init(eps_for(1, G, 0))
    ref.iterate(3, eps[:6])
    ref.iterate(3, eps[20:26])                   # the reference: plain calls
    want = ref.get_state()
    ref.close()
    a = HipEngine(**case)
    a.gamma_init(eps_for(1, G, 0))
    a.iterate(3, eps[:7])                        # carries a half made with draw 6 ...
    a.iterate(3, eps[20:26])                     # ... but the next call starts with draw 20: dropped, not used
    got = a.get_state()
    a.close()
    for k, v in want.items():
        assert np.abs(got[k] - v).max(initial=0) <= 2e-5 * max(np.abs(v).max(initial=0), 1e-30), k
    b = HipEngine(**case)
    b.gamma_init(eps_for(1, G, 0))
    b.iterate(3, eps[:7])
    v0 = b.elbo(eps[30])                         # another pass in between: the carried half is gone, the next call is a plain one
    b.iterate(3, np.concatenate([eps[6:7], eps[21:26]]))
    c = HipEngine(**case)
    c.gamma_init(eps_for(1, G, 0))
    c.iterate(3, eps[:6])
    assert abs(c.elbo(eps[30]) - v0) <= 1e-6 * abs(v0)
    c.iterate(3, np.concatenate([eps[6:7], eps[21:26]]))
    sb, sc = b.get_state(), c.get_state()
    b.close(); c.close()
    for k, v in sc.items():
        assert np.abs(sb[k] - v).max(initial=0) <= 2e-5 * max(np.abs(v).max(initial=0), 1e-30), k


def test_built_in_stream_looks_one_draw_ahead_and_consumes_what_it_did_before():
    from clonealign_amd.engine import HipEngine, eps_draw
    case = make_case(seed=5, N=900, G=256, C=4, K=1)
    G = 256
    a = HipEngine(**case, seed=77)
    a.gamma_init(None)                            # draw 0
    a.iterate(4, None)                            # draws 1..8 (+ a look at 9)
    a.iterate(4, None)                            # draws 9..16
    sa = a.get_state()
    a.close()
    eps = np.stack([eps_draw(77, d, G).reshape(1, G) for d in range(18)])
    b = HipEngine(**case, seed=77)
    b.gamma_init(eps[0])
    b.iterate(8, eps[1:17])
    sb = b.get_state()
    b.close()
    for k, v in sb.items():
        assert np.array_equal(sa[k], v), k
