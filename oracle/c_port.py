"""ORACLE (test infrastructure, NOT product code) -- ctypes wrapper of the plain C + OpenMP port
``oracle/c/clonealign_oracle.c`` (same model as fused_numpy.py; reference R/inference-tflow.R:240-346).
Serves as the CPU baseline of bench.py ("kind": "port") and as a checker at sizes numpy is slow at.
PARITY STATUS: parity unpinned against TensorFlow itself; pinned via the goldens (tests/test_oracle_c.py).
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIB = os.path.join(_DIR, "libclonealign_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _DIR])
        lib = C.CDLL(_LIB)
        lib.co_create.restype = C.c_void_p
        lib.co_create.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_int]
        lib.co_destroy.argtypes = [C.c_void_p]
        lib.co_elbo.restype = C.c_double
        lib.co_elbo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.co_gradients.restype = C.c_double
        lib.co_gradients.argtypes = [C.c_void_p, C.c_void_p]
        lib.co_gamma_init.argtypes = [C.c_void_p, C.c_void_p]
        lib.co_step.argtypes = [C.c_void_p, C.c_void_p]
        lib.co_get.restype = C.c_long
        lib.co_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        lib.co_set.restype = C.c_long
        lib.co_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        lib.co_num_threads.restype = C.c_int
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_LIB_SIMD = os.path.join(_DIR, "libclonealign_simd.so")
_lib_simd = None


def load_simd():
    """The float32 SIMD sibling (oracle/c/clonealign_simd.c): a CPU baseline, checked against the float64 port."""
    global _lib_simd
    if _lib_simd is None:
        if not os.path.exists(_LIB_SIMD):
            subprocess.check_call(["make", "-C", _DIR])
        lib = C.CDLL(_LIB_SIMD)
        lib.cs_create.restype = C.c_void_p
        lib.cs_create.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_double]
        lib.cs_destroy.argtypes = [C.c_void_p]
        lib.cs_elbo.restype = C.c_double
        lib.cs_elbo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.cs_gamma_init.argtypes = [C.c_void_p, C.c_void_p]
        lib.cs_step.argtypes = [C.c_void_p, C.c_void_p]
        lib.cs_get.restype = C.c_long
        lib.cs_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        lib.cs_num_threads.restype = C.c_int
        _lib_simd = lib
    return _lib_simd


class CPortModel:
    VAR_NAMES = ("W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits")

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None, learning_rate=0.1, dtype="float64"):
        self.lib = load()
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        L = np.ascontiguousarray(L, dtype=np.float64)
        self.N, self.G = Y.shape
        self.C, self.K, self.S = L.shape[1], int(K), int(S)
        X = None if X is None else np.ascontiguousarray(np.asarray(X, dtype=np.float64).reshape(self.N, -1))
        self.P = 0 if X is None else X.shape[1]
        assert self.C <= 64 and self.K + self.P <= 8 and self.S * self.C <= 512
        psi0 = np.ascontiguousarray(np.asarray(psi0, dtype=np.float64).reshape(self.N, self.K))
        loc0 = np.ascontiguousarray(loc0, dtype=np.float64)
        ex = None if extra_loglik is None else np.ascontiguousarray(extra_loglik, dtype=np.float64)
        self.h = self.lib.co_create(self.N, self.G, self.C, self.K, self.P, self.S, _p(Y), _p(L), _p(psi0), _p(loc0),
                                    _p(X), _p(ex), float(learning_rate), int(dtype == "float32"))

    def _shape(self, n):
        N, G, Cn, K, P = self.N, self.G, self.C, self.K, self.P
        return {"W": (G, K), "v": (K,), "psi": (N, K), "beta": (G, P), "alpha_unconstr": (Cn,), "loc": (G,), "ls": (G,),
                "gamma_logits": (N, Cn), "s": (N,)}[n]

    def _eps(self, eps):
        return np.ascontiguousarray(np.asarray(eps, dtype=np.float32).reshape(-1))

    def get(self, name, grad=False):
        out = np.zeros(self._shape(name))
        if out.size:
            self.lib.co_get(self.h, name.encode(), _p(out), int(grad))
        return out

    def set(self, name, value):
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(self._shape(name)))
        if v.size:
            self.lib.co_set(self.h, name.encode(), _p(v))

    def elbo(self, eps):
        e = self._eps(eps)
        return float(self.lib.co_elbo(self.h, _p(e), None))

    def elbo_terms(self, eps):
        e = self._eps(eps)
        t = np.zeros(3)
        self.lib.co_elbo(self.h, _p(e), _p(t))
        return tuple(t)

    def gamma_init(self, eps):
        e = self._eps(eps)
        self.lib.co_gamma_init(self.h, _p(e))

    def gradients(self, eps):
        e = self._eps(eps)
        el = float(self.lib.co_gradients(self.h, _p(e)))
        return {n: self.get(n, True) for n in self.VAR_NAMES}, el

    def step(self, eps):
        e = self._eps(eps)
        self.lib.co_step(self.h, _p(e))

    def get_state(self):
        return {n: self.get(n) for n in self.VAR_NAMES}

    def get_params(self):
        from scipy.special import logsumexp
        gl, au = self.get("gamma_logits"), self.get("alpha_unconstr")
        out = {"mu": np.logaddexp(0, self.get("loc")), "clone_probs": np.exp(gl - logsumexp(gl, 1, keepdims=True)),
               "s": self.get("s"), "alpha": np.exp(au - logsumexp(au))}
        if self.P > 0:
            out["beta"] = self.get("beta")
        if self.K > 0:
            out.update(psi=self.get("psi"), W=self.get("W"), chi=np.exp(self.get("v")))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.co_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class CSimdModel(CPortModel):
    """Same interface over the float32 SIMD port (S = 1; no ``set`` / ``gradients`` -- a baseline, not a checker)."""

    def __init__(self, Y, L, psi0, loc0, K, S=1, X=None, extra_loglik=None, learning_rate=0.1, dtype="float32"):
        assert S == 1 and dtype == "float32"
        lib = load_simd()
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        L = np.ascontiguousarray(L, dtype=np.float64)
        self.N, self.G = Y.shape
        self.C, self.K, self.S = L.shape[1], int(K), 1
        X = None if X is None else np.ascontiguousarray(np.asarray(X, dtype=np.float64).reshape(self.N, -1))
        self.P = 0 if X is None else X.shape[1]
        assert self.C <= 64 and self.K + self.P <= 8
        psi0 = np.ascontiguousarray(np.asarray(psi0, dtype=np.float64).reshape(self.N, self.K))
        loc0 = np.ascontiguousarray(loc0, dtype=np.float64)
        ex = None if extra_loglik is None else np.ascontiguousarray(extra_loglik, dtype=np.float64)
        self.h = lib.cs_create(self.N, self.G, self.C, self.K, self.P, _p(Y), _p(L), _p(psi0), _p(loc0), _p(X), _p(ex),
                               float(learning_rate))

        class _Names:   # the parent's methods call self.lib.co_*: route them to the cs_* entry points
            co_elbo, co_gamma_init, co_step, co_get, co_destroy = lib.cs_elbo, lib.cs_gamma_init, lib.cs_step, lib.cs_get, lib.cs_destroy
        self.lib = _Names

    def set(self, name, value):
        raise NotImplementedError

    def gradients(self, eps):
        raise NotImplementedError


def time_baseline(Yh, L, psi0, loc0, K, N_full, budget_s=20.0, simd=False):
    """Iterations/s of the C port on all host cores, on a cell sample, scaled to N_full cells.  ``simd``: the float32 SIMD port."""
    n, G = Yh.shape
    m = (CSimdModel if simd else CPortModel)(Yh, L, psi0[:n], loc0, K, 1, dtype="float32")
    rng = np.random.default_rng(0)
    e = lambda: rng.normal(size=G).astype(np.float32)  # noqa: E731
    m.gamma_init(e())
    m.step(e()); m.elbo(e())
    t0 = time.perf_counter()
    it = 0
    while True:
        m.step(e()); m.elbo(e())
        it += 1
        if time.perf_counter() - t0 > budget_s or it >= 200:
            break
    dt = time.perf_counter() - t0
    threads = load().co_num_threads()
    m.close()
    what = ("C+OpenMP float32 SIMD port (oracle/c/clonealign_simd.c: libmvec exp/log, AVX-512 where the host has it)" if simd
            else "C+OpenMP float64 fused oracle (oracle/c)")
    return {"value": it / dt * n / N_full, "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": f"{what}, first {n} of {N_full} cells x {G} genes, {it} iterations "
                      f"in {dt:.1f} s on {threads} threads, rate scaled by {n}/{N_full}"}
