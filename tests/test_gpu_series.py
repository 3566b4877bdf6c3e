"""The SERIES form of the loop's contraction (CA_VARX_SERIES, clonealign_amd/csrc/ca_poly.hip): for one exponent dimension, Z_nc = sum_g M_gc exp(x_n v_g) is
evaluated from moments over gene bins instead of a cells x genes sweep, and the way back has the same form.  Same fit as the matrix-core sweeps and as the
float64 oracle -- closer to the oracle, in fact: the expansion runs in float64 with a remainder below 1e-11."""
import numpy as np
import pytest

from tests._cases import eps_for, label_flips, make_case

pytestmark = pytest.mark.gpu

CASES = {"c8": dict(N=1301, G=700, C=8, K=1), "c3": dict(N=515, G=97, C=3, K=1), "c5": dict(N=2600, G=1300, C=5, K=1), "c4_ragged": dict(N=777, G=33, C=4, K=1)}


@pytest.mark.parametrize("name", list(CASES))
def test_series_loop_matches_the_oracle_and_the_sweeps(name):
    from clonealign_amd.engine import HipEngine
    from clonealign_amd.inference import run_vi_loop
    from clonealign_amd.rng import EpsStream
    from oracle.fused_numpy import FusedModel
    case = make_case(seed=29, **CASES[name])
    G = case["Y"].shape[1]
    n_iter = 8
    ser = HipEngine(**case, variant_on=("series",))
    swp = HipEngine(**case)
    ora = FusedModel(**case, dtype="float32")
    try:
        assert ser.info()["fwd_series"] == 1 and swp.info()["fwd_series"] == 0 and swp.info()["series_passes"] == 0
        ts = np.asarray(ser.run(EpsStream(5, 1, G), n_iter, 1e-12))
        tw = np.asarray(swp.run(EpsStream(5, 1, G), n_iter, 1e-12))
        to = np.asarray(run_vi_loop(ora, EpsStream(5, 1, G), n_iter, 1e-12))
        assert np.abs(ts - to).max() <= 1e-5 * np.abs(to).max(), np.abs(ts - to).max() / np.abs(to).max()
        assert np.abs(ts - tw).max() <= 1e-5 * np.abs(tw).max()
        ps, po = ser.get_state(), ora.get_state()
        for n in ("W", "psi", "loc", "ls", "gamma_logits"):
            assert np.abs(ps[n] - po[n]).max() <= 1e-4 * np.abs(po[n]).max(), (n, np.abs(ps[n] - po[n]).max() / np.abs(po[n]).max())
        assert label_flips(ser.get("clone_probs"), ora.get_params()["clone_probs"])[0] == 0
        # ca_iterate on top (carried halves, the loop without the host's look), then a call-by-call pass: the paths mix freely
        eps = np.stack([eps_for(1, G, 300 + i) for i in range(9)])
        a = ser.iterate(2, eps[:5]); a = ser.iterate(2, eps[4:9])
        for i in range(4):
            ora.step(eps[2 * i]); e = ora.elbo(eps[2 * i + 1])
        assert abs(a - e) <= 1e-5 * abs(e)
        ser.step(eps[0]); ora.step(eps[0])
        assert abs(ser.elbo(eps[1]) - ora.elbo(eps[1])) <= 1e-5 * abs(e)
    finally:
        ser.close(); swp.close()


def test_series_gradients_match_the_oracle():
    from clonealign_amd.engine import HipEngine
    from oracle.fused_numpy import FusedModel
    case = make_case(seed=31, N=900, G=260, C=6, K=1)
    G = 260
    ser, ora = HipEngine(**case, variant_on=("series",)), FusedModel(**case, dtype="float32")
    try:
        # the gradients the LOOP applies: after three iterations through the series path, compare the variables' Adam first moments indirectly --
        # run both one more train pass from identical states and compare every variable
        eps = np.stack([eps_for(1, G, 10 + i) for i in range(7)])
        ser.gamma_init(eps_for(1, G, 0)); ora.gamma_init(eps_for(1, G, 0))
        ser.iterate(3, eps)
        for i in range(3):
            ora.step(eps[2 * i]); ora.elbo(eps[2 * i + 1])
        st, so = ser.get_state(), ora.get_state()
        for n in so:
            d = np.abs(st[n] - so[n]).max(initial=0) / max(np.abs(so[n]).max(initial=0), 1e-30)
            assert d <= (5e-4 if n == "alpha_unconstr" else 1e-4), (n, d)
    finally:
        ser.close()


def test_series_handles_a_wide_exponent_range_with_more_bins_and_hands_over_to_the_sweeps_beyond():
    """psi and W pushed apart (|x v| up to ~30): more bins, same accuracy.  Beyond what 32 bins cover -- max|psi| (max W - min W) > 128, counting what the
    Adam steps since the last look can add -- the pass is given to the matrix-core sweeps BEFORE it is queued (the host looks ahead at the ranges every
    pass leaves in mapped memory; ca_info counts both kinds): never a truncated series, never an error, and the same decision on every run."""
    from clonealign_amd.engine import HipEngine
    from oracle.fused_numpy import FusedModel
    case = make_case(seed=33, N=600, G=200, C=4, K=1)
    G = 200
    rng = np.random.default_rng(2)
    W = rng.normal(0, 2.5, size=(G, 1)).astype(np.float32).astype(np.float64)
    ser, ora = HipEngine(**case, variant_on=("series",)), FusedModel(**case, dtype="float32")
    try:
        ser.gamma_init(eps_for(1, G, 0)); ora.gamma_init(eps_for(1, G, 0))
        ser.set("W", W); ora.W = W.astype(np.float32)
        eps = np.stack([eps_for(1, G, 50 + i) for i in range(5)])
        a = ser.iterate(2, eps)
        for i in range(2):
            ora.step(eps[2 * i]); e = ora.elbo(eps[2 * i + 1])
        assert abs(a - e) <= 1e-5 * abs(e), (a, e)
        i0 = ser.info()
        assert i0["series_passes"] >= 2 and i0["series_fallbacks"] == 0, i0
        Wb = (W * 14.0).astype(np.float32).astype(np.float64)      # max|psi| (max W - min W) of several hundred: past what 32 bins cover
        ser.set("W", Wb); ora.W = Wb.astype(np.float32)
        b = ser.iterate(2, eps)
        for i in range(2):
            ora.step(eps[2 * i]); e = ora.elbo(eps[2 * i + 1])
        i1 = ser.info()
        assert i1["series_passes"] == i0["series_passes"] and i1["series_fallbacks"] >= 2, (i0, i1)
        assert np.isfinite(b) == np.isfinite(e) and (not np.isfinite(e) or abs(b - e) <= 1e-5 * abs(e)), (b, e)
    finally:
        ser.close()


def test_series_decisions_are_the_same_on_every_run():
    """The look ahead decides from the ranges of the pass CA_POLY_LAG passes back, exactly -- not from whatever happens to have arrived: two runs of a fit
    that crosses the limit (W growing by hand between calls) make the same decisions and give the same bits."""
    from clonealign_amd.engine import HipEngine
    case = make_case(seed=35, N=900, G=300, C=5, K=1)
    G = 300
    eps = np.stack([eps_for(1, G, 70 + i) for i in range(13)])
    outs = []
    for rep in range(2):
        eng = HipEngine(**case, variant_on=("series",))
        try:
            eng.gamma_init(eps_for(1, G, 0))
            eng.iterate(6, eps)
            W = eng.get("W")
            eng.set("W", W * 60.0 + 3.0 * np.sign(W))
            eng.iterate(6, eps)
            i = eng.info()
            outs.append((eng.get_state(), i["series_passes"], i["series_fallbacks"]))
        finally:
            eng.close()
    assert outs[0][1:] == outs[1][1:] and outs[0][1] >= 6, (outs[0][1:], outs[1][1:])
    for n, v in outs[0][0].items():
        assert np.array_equal(v, outs[1][0][n], equal_nan=True), n
