"""ORACLE (test infrastructure, NOT product code) -- literal restatement of the reference model.

Follows ``/root/reference/R/inference-tflow.R:240-346`` op by op, materialising the
same ``[S,G,C,N]`` tensors the TensorFlow graph builds, and obtains the gradients by
reverse-mode autodiff (torch autograd standing in for ``tf.gradients``), so that it is
an independent check of the hand-derived gradients in ``oracle/fused_numpy.py`` and in
the HIP kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything under ``oracle/``.

PARITY STATUS: the reference path (R + reticulate + TensorFlow 2.1.0 + TFP 0.9) can be
neither built nor imported in this environment, and the reference's own tests
(tests/testthat/test_clonealign.R) pin only structure and seed determinism.  The third
party arithmetic is restated from the published definitions:
  * tfd.Multinomial.log_prob   = lgamma(n+1) - sum lgamma(k+1) + sum k*log p      (:294-296)
  * tfd.Normal.log_prob        = -0.5*((x-m)/s)^2 - log s - 0.5*log(2*pi)           (:312,318,323)
  * tfd.Gamma(2,1).log_prob    = log x - x                                          (:315-316)
  * tfd.Dirichlet(a).log_prob  = sum (a-1) log x - [sum lgamma(a) - lgamma(sum a)]  (:324)
  * TransformedDistribution(Normal, Softplus).log_prob(y) = Normal.log_prob(x) + softplus(-x),
    x = softplus^-1(y)                                                               (:260-266,332)
  * tf.train.AdamOptimizer (TF1 form, epsilon outside the bias correction)          (:345-346)
Each of these formulas is checked to 1e-12 against torch.distributions objects mirroring the tfd$ calls one to one and
against scipy.stats, and the Adam rule against a decimal-arithmetic table: tests/test_oracle_pin.py.
The only numeric known-answer the reference ships is the rendered vignette run
(docs/introduction_to_clonealign.html:816-819,908): see tests/test_host_api.py::test_vignette_known_answer_soft.
Otherwise: **parity unpinned** against the TensorFlow path itself.
"""
import math

import numpy as np
import torch

LOG2PI = math.log(2.0 * math.pi)


def normal_log_prob(x, loc, scale):
    return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * LOG2PI


class LiteralModel:
    """One fit's TF graph + session state (variables and Adam slots)."""

    VAR_NAMES = ("W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits")

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None,
                 learning_rate=0.1, dtype="float64"):
        dt = torch.float64 if dtype == "float64" else torch.float32
        self.dt = dt
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=dt)  # noqa: E731
        self.Y, self.L = t(Y), t(L)
        self.N, self.G = self.Y.shape
        self.C = self.L.shape[1]
        self.K, self.S = int(K), int(S)
        self.X = None if X is None else t(X).reshape(self.N, -1)
        self.P = 0 if self.X is None else self.X.shape[1]
        self.extra = None if extra_loglik is None else t(extra_loglik)
        self.s = self.Y.sum(1)                                  # :210 s_init
        z = lambda *sh: torch.zeros(*sh, dtype=dt)               # noqa: E731
        # variables, in the order of :240-272
        self.W = z(self.G, self.K)                               # :240
        self.v = z(self.K)                                       # :241 chi = exp(v)
        self.psi = t(psi0).reshape(self.N, self.K).clone()       # :242
        self.beta = z(self.G, self.P)                            # :245
        self.alpha_unconstr = z(self.C)                          # :254
        self.loc = t(loc0).clone()                               # :262
        self.ls = z(self.G)                                      # :263 log(sdinit)=0
        self.gamma_logits = z(self.N, self.C)                    # :272
        self.lr = learning_rate
        self.b1, self.b2, self.adam_eps = 0.9, 0.999, 1e-8
        self.b1p = torch.tensor(self.b1, dtype=dt)
        self.b2p = torch.tensor(self.b2, dtype=dt)
        self.m = {n: torch.zeros_like(getattr(self, n)) for n in self.VAR_NAMES}
        self.vv = {n: torch.zeros_like(getattr(self, n)) for n in self.VAR_NAMES}

    # ------------------------------------------------------------------ graph
    def _p_y_on_c(self, eps):
        """[S,C,N] multinomial log-probs, :268-304."""
        eps = torch.as_tensor(np.asarray(eps, dtype=np.float64), dtype=self.dt).reshape(self.S, self.G)
        x = self.loc + torch.exp(self.ls) * eps                  # Normal sample
        mu = torch.nn.functional.softplus(x)                     # :269 [S,G]
        if self.P == 0 and self.K > 0:                           # :279-285
            rfe = torch.exp(self.psi @ self.W.T)
        elif self.P > 0 and self.K > 0:
            rfe = torch.exp(self.psi @ self.W.T + self.X @ self.beta.T)
        else:
            rfe = torch.ones(self.N, self.G, dtype=self.dt)
        mu_scg = torch.einsum("sg,gc->scg", mu, self.L)          # :288
        mu_sgcn = torch.einsum("scg,ng->sgcn", mu_scg, rfe)      # :289
        norm = 1.0 / mu_sgcn.sum(1)                              # :290 [S,C,N]
        mu_sgcn_norm = torch.einsum("sgcn,scn->sgcn", mu_sgcn, norm)   # :291
        probs = mu_sgcn_norm.permute(0, 2, 3, 1)                 # :292 [S,C,N,G]
        logp = torch.log(probs)
        lp = (torch.lgamma(self.s + 1.0) - torch.lgamma(self.Y + 1.0).sum(1)
              + torch.xlogy(self.Y, probs).sum(-1))              # :294-296
        del logp
        if self.extra is not None:                               # :302-304
            lp = lp + self.extra.T
        return lp, mu, x, eps

    def _elbo_terms(self, eps):
        p_y_on_c, mu, x, eps = self._p_y_on_c(eps)
        gamma = torch.softmax(self.gamma_logits, 1)              # :273
        log_alpha = torch.log_softmax(self.alpha_unconstr, 0)    # :255
        E_p_y_on_c = p_y_on_c.mean(0)                            # :306
        EE_p_y = (gamma * E_p_y_on_c.T).sum()                    # :308
        one = torch.ones(1, dtype=self.dt)
        zero = torch.zeros(1, dtype=self.dt)
        C = self.C
        conc = torch.full((C,), 1.0 / C, dtype=self.dt)
        xa = torch.exp(log_alpha) + 1e-3
        dirichlet = ((conc - 1.0) * torch.log(xa)).sum() - (torch.lgamma(conc).sum() - torch.lgamma(conc.sum()))
        E_log_p_p = ((log_alpha * gamma).sum()
                     + normal_log_prob(torch.log(mu), zero, one).sum() / float(self.S)
                     + dirichlet)                                # :322-324
        if self.K > 0:                                           # :311-320,326-328
            chi = torch.exp(self.v)
            W_lp = normal_log_prob(self.W, zero, torch.sqrt(one / chi)).sum()
            chi_lp = (torch.log(chi) - chi).sum()                # Gamma(2, 1)
            psi_lp = normal_log_prob(self.psi, zero, one).sum()
            E_log_p_p = E_log_p_p + W_lp + chi_lp + psi_lp
        # qmu.log_prob(mu_samples): Normal(loc, scale).log_prob(x) + softplus(-x)
        qlp = normal_log_prob(x, self.loc, torch.exp(self.ls)) + torch.nn.functional.softplus(-x)
        log_gamma = torch.log_softmax(self.gamma_logits, 1)
        ent = torch.where(gamma == 0, torch.zeros_like(gamma), gamma * log_gamma).sum()
        E_log_q = qlp.mean(0).sum() + ent                        # :332-333
        return EE_p_y, E_log_p_p, E_log_q

    # ---------------------------------------------------------------- session
    def elbo(self, eps):
        with torch.no_grad():
            a, b, c = self._elbo_terms(eps)
            return float(a + b - c)                              # :336

    def elbo_terms(self, eps):
        with torch.no_grad():
            return tuple(float(t) for t in self._elbo_terms(eps))

    def gamma_init(self, eps):
        """:338-342,368-369 -- note: SUM over MC samples, no log_alpha term."""
        with torch.no_grad():
            p, _, _, _ = self._p_y_on_c(eps)
            gi = p.sum(0)
            gi = gi - torch.logsumexp(gi, 0)
            self.gamma_logits = gi.T.contiguous().clone()

    def gradients(self, eps):
        """d(elbo)/d(var) for every variable (autodiff)."""
        vs = [getattr(self, n) for n in self.VAR_NAMES]
        for t in vs:
            t.requires_grad_(True)
        a, b, c = self._elbo_terms(eps)
        elbo = a + b - c
        used = [t for t in vs if t.numel() > 0]
        gr = torch.autograd.grad(elbo, used, allow_unused=True)
        it = iter(gr)
        out = {}
        for n, t in zip(self.VAR_NAMES, vs):
            t.requires_grad_(False)
            if t.numel() > 0:
                g = next(it)
                out[n] = torch.zeros_like(t) if g is None else g.detach()
            else:
                out[n] = torch.zeros_like(t)
        return out, float(elbo.detach())

    def step(self, eps):
        """One ``sess$run(train)``: minimize(-elbo) with TF1 Adam (:345-346,401)."""
        g, _ = self.gradients(eps)
        lr_t = self.lr * torch.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for n in self.VAR_NAMES:
            grad = -g[n]
            self.m[n] = self.b1 * self.m[n] + (1.0 - self.b1) * grad
            self.vv[n] = self.b2 * self.vv[n] + (1.0 - self.b2) * grad * grad
            setattr(self, n, (getattr(self, n) - lr_t * self.m[n] / (torch.sqrt(self.vv[n]) + self.adam_eps)).detach())
        self.b1p = self.b1p * self.b1
        self.b2p = self.b2p * self.b2

    def get_params(self):
        """:424-434 fetch."""
        with torch.no_grad():
            out = {
                "mu": torch.nn.functional.softplus(self.loc).numpy().astype(np.float64),
                "clone_probs": torch.softmax(self.gamma_logits, 1).numpy().astype(np.float64),
                "s": self.s.numpy().astype(np.float64),
                "alpha": torch.exp(torch.log_softmax(self.alpha_unconstr, 0)).numpy().astype(np.float64),
            }
            if self.P > 0:
                out["beta"] = self.beta.numpy().astype(np.float64)
            if self.K > 0:
                out["psi"] = self.psi.numpy().astype(np.float64)
                out["W"] = self.W.numpy().astype(np.float64)
                out["chi"] = torch.exp(self.v).numpy().astype(np.float64)
            return out

    def get_state(self):
        return {n: getattr(self, n).numpy().astype(np.float64).copy() for n in self.VAR_NAMES}

    def close(self):
        pass
