"""The restated third-party arithmetic of the oracle, checked against INDEPENDENT library implementations (CPU only).

The reference's ELBO (R/inference-tflow.R:260-336) is written with TensorFlow-Probability distribution objects; neither R nor
TensorFlow can run here, and the reference's tests pin no number (tests/testthat/test_clonealign.R:17-37,61-64 check structure
and seed determinism).  oracle/literal_torch.py restates every ``tfd$...$log_prob`` as a closed formula.  This file rebuilds the
same ELBO a SECOND time from ``torch.distributions`` objects that mirror the reference's ``tfd$`` calls one to one, and a third
time from ``scipy.stats``, and asserts agreement with ``LiteralModel.elbo_terms`` to 1e-12 on the three golden cases -- so a
formula restated wrongly from memory (a sign, a normaliser, a Jacobian) cannot survive.  It does not make the oracle "pinned to
TensorFlow": two independent libraries agreeing with the published definitions is the closest thing this container allows.

The TF1 Adam rule (tf.train.AdamOptimizer, R/inference-tflow.R:345-346: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t),
theta -= lr_t m / (sqrt(v) + eps) -- epsilon OUTSIDE the bias correction) is pinned by a table worked out in 60-digit decimal
arithmetic (script in the docstring of ``ADAM_TABLE``); its third row has gradients of 1e-6, where sqrt(v) is comparable with
eps = 1e-8 and the torch/Keras form of the rule (eps added to the bias-corrected sqrt(v_hat)) gives a different step.
"""
import numpy as np
import pytest
import scipy.special
import scipy.stats
import torch
import torch.distributions as td

from oracle.fused_numpy import FusedModel
from oracle.literal_torch import LiteralModel
from tests import _golden

F64 = torch.float64


def _t(a):
    return torch.tensor(np.asarray(a, dtype=np.float64), dtype=F64)


def _load_state(model, g):
    """Put the golden run's FINAL variables into the model (generic values: every term of the ELBO is away from its initial zero)."""
    for n in LiteralModel.VAR_NAMES:
        v = g["state_" + n]
        setattr(model, n, _t(v) if isinstance(model, LiteralModel) else np.array(v, dtype=np.float64))


def elbo_terms_from_distribution_objects(case, state, eps):
    """R/inference-tflow.R:240-336 with one torch.distributions object per tfd$ call (line numbers on the right)."""
    Y, L = _t(case["Y"]), _t(case["L"])
    N, G = Y.shape
    C = L.shape[1]
    K, S = int(case["K"]), int(case["S"])
    X = None if case.get("X") is None else _t(case["X"]).reshape(N, -1)
    P = 0 if X is None else X.shape[1]
    W, v, psi, beta = _t(state["W"]), _t(state["v"]), _t(state["psi"]), _t(state["beta"])
    alpha_unconstr, loc, ls, gamma_logits = _t(state["alpha_unconstr"]), _t(state["loc"]), _t(state["ls"]), _t(state["gamma_logits"])
    eps = _t(eps).reshape(S, G)
    s = Y.sum(1)                                                                                   # :210
    chi = torch.exp(v)                                                                             # :241
    log_alpha = torch.log_softmax(alpha_unconstr, 0)                                               # :255
    base = td.Normal(loc, torch.exp(ls))                                                           # :261-263
    qmu = td.TransformedDistribution(base, [td.transforms.SoftplusTransform()])                    # :260-266
    mu_samples = td.transforms.SoftplusTransform()(loc + torch.exp(ls) * eps)                      # :269 (sample = bijector(loc + scale eps))
    gamma = torch.softmax(gamma_logits, 1)                                                         # :273
    if P == 0 and K > 0:                                                                           # :279-285
        rfe = torch.exp(psi @ W.T)
    elif P > 0 and K > 0:
        rfe = torch.exp(psi @ W.T + X @ beta.T)
    else:
        rfe = torch.ones(N, G, dtype=F64)
    mu_scg = torch.einsum("sg,gc->scg", mu_samples, L)                                             # :288
    mu_sgcn = torch.einsum("scg,ng->sgcn", mu_scg, rfe)                                            # :289
    mu_scng = (mu_sgcn / mu_sgcn.sum(1, keepdim=True)).permute(0, 2, 3, 1)                         # :290-292
    # (torch's Multinomial takes ONE integer total_count and validates counts against it; its log_prob, like TFP's, uses the
    #  counts' own sum, so validate_args=False gives the per-cell totals of :294)
    y_pdf = td.Multinomial(total_count=1, probs=mu_scng, validate_args=False)                      # :294
    p_y_on_c = y_pdf.log_prob(Y)                                                                   # :296  [S,C,N]
    p_y_scipy = scipy.stats.multinomial.logpmf(Y.numpy().astype(np.int64)[None, None], n=s.numpy().astype(np.int64)[None, None],
                                               p=mu_scng.numpy())
    np.testing.assert_allclose(p_y_on_c.numpy(), p_y_scipy, rtol=1e-12, atol=1e-9)
    if case.get("extra_loglik") is not None:                                                       # :302-304
        p_y_on_c = p_y_on_c + _t(case["extra_loglik"]).T
    EE_p_y = (gamma * p_y_on_c.mean(0).T).sum()                                                    # :306-308
    one, zero = torch.ones(1, dtype=F64), torch.zeros(1, dtype=F64)
    E_log_p_p = ((log_alpha * gamma).sum()                                                         # :322
                 + td.Normal(zero, one).log_prob(torch.log(mu_samples)).sum() / float(S)           # :323
                 + td.Dirichlet(torch.full((C,), 1.0 / C, dtype=F64), validate_args=False)
                   .log_prob(torch.exp(log_alpha) + 1e-3).sum())                                   # :324
    if K > 0:
        W_lp = td.Normal(zero, torch.sqrt(one / chi)).log_prob(W).sum()                            # :312-313
        chi_lp = td.Gamma(torch.tensor(2.0, dtype=F64), one).log_prob(chi).sum()                   # :315-316
        psi_lp = td.Normal(zero, one).log_prob(psi).sum()                                          # :318-319
        E_log_p_p = E_log_p_p + W_lp + chi_lp + psi_lp                                             # :326-328
        sc = dict(
            W=scipy.stats.norm.logpdf(W.numpy(), 0.0, np.sqrt(1.0 / chi.numpy())).sum(),
            chi=scipy.stats.gamma.logpdf(chi.numpy(), a=2.0, scale=1.0).sum(),
            psi=scipy.stats.norm.logpdf(psi.numpy()).sum())
        np.testing.assert_allclose([float(W_lp), float(chi_lp), float(psi_lp)], [sc["W"], sc["chi"], sc["psi"]], rtol=1e-12, atol=1e-12)
    # :332 qmu$log_prob; second opinion from the change of variables written with scipy (x = softplus^-1(mu))
    qlp = qmu.log_prob(mu_samples)
    x = (loc + torch.exp(ls) * eps).numpy()
    q_scipy = scipy.stats.norm.logpdf(x, loc.numpy(), np.exp(ls.numpy())) - np.log(scipy.special.expit(x))   # d softplus/dx = sigmoid(x)
    np.testing.assert_allclose(qlp.numpy(), q_scipy, rtol=1e-12, atol=1e-12)
    # :333 sum gamma log gamma = minus the entropy of Categorical(logits)
    ent = -td.Categorical(logits=gamma_logits).entropy().sum()
    E_log_q = qlp.mean(0).sum() + ent
    return float(EE_p_y), float(E_log_p_p), float(E_log_q)


@pytest.mark.parametrize("name", ["cfg1", "tiny_k0", "tiny_full"])
@pytest.mark.parametrize("where", ["initial", "final"])
def test_restated_log_probs_equal_distribution_objects(name, where):
    g = _golden.load(name)
    case = _golden.case_of(name, g)
    lit, fus = LiteralModel(**case), FusedModel(**case)
    if where == "final":
        _load_state(lit, g)
        _load_state(fus, g)
    else:                                          # initial values + the gamma initialisation of :338-342,368-369
        lit.gamma_init(g["eps"][0])
        fus.gamma_init(g["eps"][0])
    state = lit.get_state()
    for j in (1, 5):
        eps = g["eps"][j]
        want = elbo_terms_from_distribution_objects(case, state, eps)
        np.testing.assert_allclose(lit.elbo_terms(eps), want, rtol=1e-12)
        np.testing.assert_allclose(fus.elbo_terms(eps), want, rtol=1e-11)     # (fused: blocked summation order)
        assert abs(lit.elbo(eps) - (want[0] + want[1] - want[2])) <= 1e-12 * abs(lit.elbo(eps))     # :336


def test_gamma_init_is_the_posterior_under_a_flat_prior():
    """:338-342 -- logits = sum_s loglik - logsumexp_c: softmax of them is the normalised likelihood (no log_alpha term)."""
    g = _golden.load("tiny_full")
    case = _golden.case_of("tiny_full", g)
    lit = LiteralModel(**case)
    p, *_ = lit._p_y_on_c(g["eps"][0])
    lit.gamma_init(g["eps"][0])
    want = scipy.special.softmax(p.sum(0).numpy().T, axis=1)
    np.testing.assert_allclose(torch.softmax(lit.gamma_logits, 1).numpy(), want, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(scipy.special.logsumexp(lit.gamma_logits.numpy(), axis=1), 0.0, atol=1e-9)


# TF1 Adam after 1, 2, 3 steps, lr = 0.1, beta1 = 0.9, beta2 = 0.999, eps = 1e-8 (the AdamOptimizer defaults the reference
# leaves in place, :345).  Worked in decimal arithmetic, 60 digits:
#   from decimal import Decimal as D, getcontext; getcontext().prec = 60
#   m = v = 0; b1p, b2p = b1, b2
#   for g in grads: lr_t = lr*(1-b2p).sqrt()/(1-b1p); m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g
#                   theta -= lr_t*m/(v.sqrt()+eps); b1p *= b1; b2p *= b2
# rows: (gradients of the MINIMISED function, theta_0, theta after each step)
ADAM_TABLE = [
    ((1.0, -0.5, 0.25), 0.0, (-0.0999999683772333983130, -0.1266336648071116469157, -0.1606765702232370190034)),
    ((-2000.0, -1000.0, 3000.0), 1.5, (1.59999999998418861170165, 1.69321796385214122743113, 1.68502028406775423816147)),
    ((1e-6, 2e-6, -1e-6), -0.25, (-0.3259746926647957852000, -0.4105333055170806220771, -0.4477368730735299128478)),
]


@pytest.mark.parametrize("model_cls", [LiteralModel, FusedModel])
@pytest.mark.parametrize("row", range(len(ADAM_TABLE)))
def test_adam_rule_is_tf1_epsilon_hat_form(model_cls, row):
    grads, theta0, want = ADAM_TABLE[row]
    g = _golden.load("tiny_k0")
    m = model_cls(**_golden.case_of("tiny_k0", g))
    lit = model_cls is LiteralModel
    m.loc = _t(np.full(m.G, theta0)) if lit else np.full(m.G, theta0)
    seq = iter(grads)

    def prescribed(_eps):   # the oracle MAXIMISES the ELBO: step() negates what gradients() returns (minimize(-elbo), :346)
        gv = -next(seq)
        out = {n: (torch.zeros_like(getattr(m, n)) if lit else np.zeros_like(getattr(m, n))) for n in m.VAR_NAMES}
        out["loc"] = out["loc"] + gv
        return out, 0.0
    m.gradients = prescribed
    for k in range(3):
        m.step(None)
        got = np.asarray(m.loc if not lit else m.loc.numpy())
        np.testing.assert_allclose(got, want[k], rtol=2e-15, atol=0)
    # and the rule is NOT the torch/Keras one on the row that can tell them apart
    if row == 2:
        p = torch.nn.Parameter(torch.tensor([theta0], dtype=F64))
        opt = torch.optim.Adam([p], lr=0.1, betas=(0.9, 0.999), eps=1e-8)
        for gv in grads:
            p.grad = torch.tensor([gv], dtype=F64)
            opt.step()
        assert abs(float(p.detach()) - want[2]) > 1e-2


def _zero_copy_number_case():
    """tiny_full with one gene that no cell expresses and whose copy number is 0 in clone 1: y = 0 AND L = 0 at every (cell, that gene,
    clone 1) -- the `0 * log 0` of tfd$Multinomial$log_prob (R/inference-tflow.R:294-296), which TFP-0.9's `counts * log(probs)` turns
    into NaN and later TFP versions (`multiply_no_nan`) into 0.  The oracles and the engine take the second reading (SURVEY.md section 7.4,
    DESIGN.md section 3): xlogy, 0 * log 0 := 0."""
    g = _golden.load("tiny_full")
    case = _golden.case_of("tiny_full", g)
    Y, L = np.array(case["Y"], dtype=np.float64), np.array(case["L"], dtype=np.float64)
    g0 = 3
    Y[:, g0] = 0.0
    L[g0, 1] = 0.0
    case = dict(case, Y=Y, L=L)
    return g, case, g0


def test_zero_copy_number_where_nothing_is_counted_is_xlogy_zero_not_nan():
    """VERDICT r4 #9a: the stated semantics of `L_gc = 0`, asserted as a VALUE (the NaN-initial-ELBO error for y > 0 is a GPU test already):
    with y = 0 wherever L = 0 every ELBO term is finite and equals what torch.distributions.Multinomial and scipy.stats.multinomial give
    for a probability vector with an exact zero at an uncounted category -- both libraries define that term as 0 -- to 1e-12; and the
    zero entry changes the fit only through Z: the same case with L = 1e-300 there gives the same ELBO to 1e-12."""
    g, case, g0 = _zero_copy_number_case()
    lit, fus = LiteralModel(**case), FusedModel(**case)
    lit.gamma_init(g["eps"][0]); fus.gamma_init(g["eps"][0])
    state = lit.get_state()
    for j in (1, 4):
        eps = g["eps"][j]
        want = elbo_terms_from_distribution_objects(case, state, eps)     # (asserts torch == scipy on the multinomial term inside)
        assert np.all(np.isfinite(want))
        np.testing.assert_allclose(lit.elbo_terms(eps), want, rtol=1e-12)
        np.testing.assert_allclose(fus.elbo_terms(eps), want, rtol=1e-11)
    tiny = dict(case, L=np.where(case["L"] == 0.0, 1e-300, case["L"]))
    lit2 = LiteralModel(**tiny)
    lit2.gamma_init(g["eps"][0])
    np.testing.assert_allclose(lit2.elbo(g["eps"][1]), lit.elbo(g["eps"][1]), rtol=1e-12)
    # gradients stay finite too (the backward pass multiplies by L, never divides by it)
    gr, _ = fus.gradients(g["eps"][2])
    assert all(np.all(np.isfinite(v)) for v in gr.values())
    # ... and ONE count on that gene makes the clone impossible for that cell: log-likelihood -inf, posterior weight exactly 0 after the
    # gamma initialisation (:338-342), ELBO NaN from 0 * (-inf) -- the reference's "Initial elbo is NA" (:374-376)
    Yb = case["Y"].copy(); Yb[2, g0] = 1.0
    bad = FusedModel(**dict(case, Y=Yb))
    bad.gamma_init(g["eps"][0])
    gam = np.exp(bad.gamma_logits - scipy.special.logsumexp(bad.gamma_logits, axis=1, keepdims=True))
    assert gam[2, 1] == 0.0 and np.isnan(bad.elbo(g["eps"][1]))
