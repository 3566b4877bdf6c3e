"""ORACLE (test infrastructure, NOT product code) -- fused float64 restatement with hand gradients.

Same model as ``oracle/literal_torch.py`` (reference ``R/inference-tflow.R:240-346``), but
evaluated the way the HIP engine evaluates it: constants hoisted (``A = Y.log L``,
``c_n``, ``colsum``), ``E = exp(psi W^T + X beta^T)`` generated blockwise, the
cells x genes x clones contraction ``Z = E.M`` done as a matmul, and every gradient written
out by hand.  It is cross-checked against the literal/autodiff oracle in
``tests/test_oracle.py``; the HIP kernels are then checked against this one (it scales to
the 10k x 2k x 4 configuration in seconds, the literal one does not).

Variables can be held in float32 (``dtype="float32"``, what TensorFlow's float32 variables
do) while every pass is evaluated in float64.

PARITY STATUS: see the header of ``oracle/literal_torch.py`` -- **parity unpinned** against
the TensorFlow path itself; pinned only by the vignette known-answer and by
literal <-> fused <-> finite-difference agreement.
"""
import math

import numpy as np
from scipy.special import gammaln, logsumexp, xlogy

LOG2PI = math.log(2.0 * math.pi)


def softplus(x):
    return np.logaddexp(0.0, x)


def sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


class FusedModel:
    VAR_NAMES = ("W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits")

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None,
                 learning_rate=0.1, dtype="float64", block=4096):
        self.pdt = np.float64 if dtype == "float64" else np.float32
        Y = np.asarray(Y, dtype=np.float64)
        self.Y = Y
        self.L = np.asarray(L, dtype=np.float64)
        self.N, self.G = Y.shape
        self.C = self.L.shape[1]
        self.K, self.S = int(K), int(S)
        self.X = None if X is None else np.asarray(X, dtype=np.float64).reshape(self.N, -1)
        self.P = 0 if self.X is None else self.X.shape[1]
        self.extra = None if extra_loglik is None else np.asarray(extra_loglik, dtype=np.float64)
        self.block = block
        # constants of the fit
        self.s = Y.sum(1)
        self.cn = gammaln(self.s + 1.0) - gammaln(Y + 1.0).sum(1)
        self.colsum = Y.sum(0)
        with np.errstate(divide="ignore"):
            self.A = xlogy(Y[:, :, None], self.L[None, :, :]).sum(1) if self.N * self.G * self.C < 5e7 \
                else self._A_blocked()
        # K = 0 silently disables covariates too (R/inference-tflow.R:279-285)
        self.D = (self.K + self.P) if self.K > 0 else 0
        self.YtX = (Y.T @ self.X) if (self.P > 0 and self.K > 0) else np.zeros((self.G, 0))
        p = self.pdt
        self.W = np.zeros((self.G, self.K), p)
        self.v = np.zeros(self.K, p)
        self.psi = np.asarray(psi0, dtype=np.float64).reshape(self.N, self.K).astype(p)
        self.beta = np.zeros((self.G, self.P), p)
        self.alpha_unconstr = np.zeros(self.C, p)
        self.loc = np.asarray(loc0, dtype=np.float64).astype(p)
        self.ls = np.zeros(self.G, p)
        self.gamma_logits = np.zeros((self.N, self.C), p)
        # cell-sharded evaluation: `allreduce(vec)` must return the sum of `vec` over all
        # shards.  It is applied at exactly the points where the HIP engine all-reduces.
        self.allreduce = None
        self.lr, self.b1, self.b2, self.adam_eps = learning_rate, 0.9, 0.999, 1e-8
        self.b1p, self.b2p = p(self.b1), p(self.b2)
        self.m = {n: np.zeros_like(getattr(self, n)) for n in self.VAR_NAMES}
        self.vv = {n: np.zeros_like(getattr(self, n)) for n in self.VAR_NAMES}

    def set_allreduce(self, fn):
        """Switch to sharded mode: this model holds a shard of the cells; per-gene count totals become global."""
        self.allreduce = fn
        self.colsum = fn(self.colsum.copy())
        if self.YtX.size:
            self.YtX = fn(self.YtX.reshape(-1).copy()).reshape(self.YtX.shape)

    def _ar(self, v):
        return v if self.allreduce is None else self.allreduce(np.ascontiguousarray(v, dtype=np.float64))

    def _A_blocked(self):
        with np.errstate(divide="ignore"):
            logL = np.log(self.L)
        A = np.empty((self.N, self.C))
        for c in range(self.C):
            A[:, c] = xlogy(self.Y, np.exp(logL[:, c])[None, :]).sum(1) if not np.all(np.isfinite(logL[:, c])) \
                else self.Y @ logL[:, c]
        return A

    # cell factors F = [psi | X], gene loadings V = [W | beta]
    def _FV(self):
        if self.D == 0:
            return np.zeros((self.N, 0)), np.zeros((self.G, 0))
        F = self.psi.astype(np.float64)
        V = self.W.astype(np.float64)
        if self.P > 0:
            F = np.concatenate([F, self.X], 1)
            V = np.concatenate([V, self.beta.astype(np.float64)], 1)
        return F, V

    def _forward(self, eps, need_grad):
        S, G, C, N = self.S, self.G, self.C, self.N
        eps = np.asarray(eps, dtype=np.float64).reshape(S, G)
        loc, ls = self.loc.astype(np.float64), self.ls.astype(np.float64)
        x = loc + np.exp(ls) * eps                     # [S,G]
        mu = softplus(x)
        F, V = self._FV()
        logZ = np.empty((S, N, C))
        for b0 in range(0, N, self.block):
            sl = slice(b0, min(N, b0 + self.block))
            E = np.exp(F[sl] @ V.T) if self.D > 0 else np.ones((sl.stop - sl.start, G))
            for s_ in range(S):
                logZ[s_, sl] = np.log(E @ (mu[s_][:, None] * self.L))
        cache = dict(x=x, mu=mu, eps=eps, F=F, V=V, logZ=logZ)
        return cache

    def _elbo_from(self, c):
        S, C = self.S, self.C
        x, mu, eps, F, V, logZ = c["x"], c["mu"], c["eps"], c["F"], c["V"], c["logZ"]
        gl = self.gamma_logits.astype(np.float64)
        log_gamma = gl - logsumexp(gl, 1, keepdims=True)
        gamma = np.exp(log_gamma)
        au = self.alpha_unconstr.astype(np.float64)
        log_alpha = au - logsumexp(au)
        # c-dependent part of the per cell/clone log-lik, averaged over samples
        llp = self.A - self.s[:, None] * logZ.mean(0)
        if self.extra is not None:
            llp = llp + self.extra
        logmu = np.log(mu)
        # ---- per-cell summands: local to a shard, all-reduced as one 3-vector
        T = 0.0
        psi_prior = 0.0
        if self.D > 0:
            YW = self.Y @ V[:, :self.K]
            T = float((F[:, :self.K] * YW).sum())
            c["YW"] = YW
            psi = self.psi.astype(np.float64)
            psi_prior = float((-0.5 * psi ** 2 - 0.5 * LOG2PI).sum())
        ent = float(np.where(gamma == 0, 0.0, gamma * log_gamma).sum())
        loc3 = self._ar(np.array([float(self.cn.sum()) + T + float((gamma * llp).sum()),
                                  float((gamma * log_alpha[None, :]).sum()) + psi_prior, ent]))
        # ---- global summands (replicated parameters; identical on every shard)
        EE_p_y = loc3[0] + float((self.colsum[None, :] * logmu).sum()) / S
        if self.D > 0 and self.P > 0:
            EE_p_y += float((V[:, self.K:] * self.YtX).sum())
        xa = np.exp(log_alpha) + 1e-3
        dirichlet = float(((1.0 / C - 1.0) * np.log(xa)).sum()) - (C * gammaln(1.0 / C) - gammaln(1.0))
        E_log_p_p = loc3[1] + float((-0.5 * logmu ** 2 - 0.5 * LOG2PI).sum()) / S + dirichlet
        if self.K > 0:
            W = self.W.astype(np.float64)
            v = self.v.astype(np.float64)
            chi = np.exp(v)
            E_log_p_p += float((-0.5 * W ** 2 * chi[None, :] + 0.5 * v[None, :] - 0.5 * LOG2PI).sum())
            E_log_p_p += float((v - chi).sum())
        ls = self.ls.astype(np.float64)
        qlp = -0.5 * eps ** 2 - ls[None, :] - 0.5 * LOG2PI + softplus(-x)
        E_log_q = float(qlp.mean(0).sum()) + loc3[2]
        c.update(gamma=gamma, log_gamma=log_gamma, log_alpha=log_alpha, llp=llp, logmu=logmu)
        return EE_p_y, E_log_p_p, E_log_q

    def elbo_terms(self, eps):
        return self._elbo_from(self._forward(eps, False))

    def elbo(self, eps):
        a, b, c = self.elbo_terms(eps)
        return a + b - c

    def gamma_init(self, eps):
        c = self._forward(eps, False)
        ll = self.S * self.A - self.s[:, None] * c["logZ"].sum(0)   # SUM over samples (:338)
        if self.extra is not None:
            ll = ll + self.S * self.extra
        self.gamma_logits = (ll - logsumexp(ll, 1, keepdims=True)).astype(self.pdt)

    def gradients(self, eps):
        S, G, C, N, K, P = self.S, self.G, self.C, self.N, self.K, self.P
        c = self._forward(eps, True)
        a, b, q = self._elbo_from(c)
        x, mu, eps, F, V, logZ = c["x"], c["mu"], c["eps"], c["F"], c["V"], c["logZ"]
        gamma, log_gamma, log_alpha, llp, logmu = c["gamma"], c["log_gamma"], c["log_alpha"], c["llp"], c["logmu"]
        # data-term sweeps
        coef = -gamma[None] * self.s[None, :, None] / (S * np.exp(logZ))      # [S,N,C]
        dmu = np.zeros((S, G))
        dV = np.zeros((G, self.D))
        dF = np.zeros((N, self.D))
        for b0 in range(0, N, self.block):
            sl = slice(b0, min(N, b0 + self.block))
            E = np.exp(F[sl] @ V.T) if self.D > 0 else np.ones((sl.stop - sl.start, G))
            deta = np.zeros_like(E)
            for s_ in range(S):
                t = coef[s_, sl] @ self.L.T                  # [n,G]  sum_c coef * l_gc
                u = E * t
                dmu[s_] += u.sum(0)
                deta += u * mu[s_][None, :]
            if self.D > 0:
                dF[sl] = deta @ V
                dV += deta.T @ F[sl]
        YtPsi = self.Y.T @ self.psi.astype(np.float64) if K > 0 else np.zeros((G, 0))
        sg = gamma.sum(0)
        if self.allreduce is not None:          # one all-reduce per train pass: [dmu | dV | Y^T psi | sum gamma]
            pk = self._ar(np.concatenate([dmu.ravel(), dV.ravel(), YtPsi.ravel(), sg]))
            o = 0
            dmu = pk[o:o + dmu.size].reshape(dmu.shape); o += dmu.size
            dV = pk[o:o + dV.size].reshape(dV.shape); o += dV.size
            YtPsi = pk[o:o + YtPsi.size].reshape(YtPsi.shape); o += YtPsi.size
            sg = pk[o:o + C]
        g = {}
        # q(mu) parameters
        dmu_tot = self.colsum[None, :] / (S * mu) + dmu - logmu / (S * mu)
        sig = sigmoid(x)
        dx = dmu_tot * sig + (1.0 - sig) / S
        ls = self.ls.astype(np.float64)
        g["loc"] = dx.sum(0)
        g["ls"] = (dx * eps * np.exp(ls)[None, :]).sum(0) + 1.0
        # latent factors
        if K > 0:
            W = self.W.astype(np.float64)
            v = self.v.astype(np.float64)
            chi = np.exp(v)
            psi = self.psi.astype(np.float64)
            g["W"] = YtPsi + dV[:, :K] - W * chi[None, :]
            g["v"] = -0.5 * chi * (W ** 2).sum(0) + 0.5 * G + 1.0 - chi
            g["psi"] = c["YW"] + dF[:, :K] - psi
        else:
            g["W"] = np.zeros((G, 0))
            g["v"] = np.zeros(0)
            g["psi"] = np.zeros((N, 0))
        if P > 0 and K > 0:
            g["beta"] = self.YtX + dV[:, K:]
        else:
            g["beta"] = np.zeros((G, P))
        # q(z) logits
        f = llp + log_alpha[None, :] - log_gamma
        fbar = (gamma * f).sum(1, keepdims=True)
        g["gamma_logits"] = gamma * (f - fbar)
        # alpha
        alpha = np.exp(log_alpha)
        # d/d log_alpha_c of [sum gamma log_alpha + dirichlet(alpha + 1e-3)], then through log_softmax
        dla = sg + (1.0 / C - 1.0) * alpha / (alpha + 1e-3)
        g["alpha_unconstr"] = dla - alpha * dla.sum()
        return g, a + b - q

    def step(self, eps):
        g, _ = self.gradients(eps)
        p = self.pdt
        lr_t = p(self.lr) * np.sqrt(p(1.0) - self.b2p) / (p(1.0) - self.b1p)
        for n in self.VAR_NAMES:
            grad = (-g[n]).astype(p)
            self.m[n] = (p(self.b1) * self.m[n] + p(1.0 - self.b1) * grad).astype(p)
            self.vv[n] = (p(self.b2) * self.vv[n] + p(1.0 - self.b2) * grad * grad).astype(p)
            upd = lr_t * self.m[n] / (np.sqrt(self.vv[n]) + p(self.adam_eps))
            setattr(self, n, (getattr(self, n) - upd).astype(p))
        self.b1p = p(self.b1p * p(self.b1))
        self.b2p = p(self.b2p * p(self.b2))

    def get_params(self):
        gl = self.gamma_logits.astype(np.float64)
        au = self.alpha_unconstr.astype(np.float64)
        out = {
            "mu": softplus(self.loc.astype(np.float64)),
            "clone_probs": np.exp(gl - logsumexp(gl, 1, keepdims=True)),
            "s": self.s.copy(),
            "alpha": np.exp(au - logsumexp(au)),
        }
        if self.P > 0:
            out["beta"] = self.beta.astype(np.float64)
        if self.K > 0:
            out["psi"] = self.psi.astype(np.float64)
            out["W"] = self.W.astype(np.float64)
            out["chi"] = np.exp(self.v.astype(np.float64))
        return out

    def get_state(self):
        return {n: getattr(self, n).astype(np.float64).copy() for n in self.VAR_NAMES}

    def close(self):
        pass
