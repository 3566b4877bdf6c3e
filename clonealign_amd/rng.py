"""Counter-based noise stream for the reparametrised q(mu) samples.

The reference draws ``mu_samples = qmu$sample(S, seed = get_next_seed())``
(R/inference-tflow.R:49-51,269): every ``sess$run`` that touches the sample gets a
fresh standard-normal ``eps[S,G]`` from TensorFlow's Philox stream.  Bit-reproducing
TF's stream is not a goal; the engine instead consumes an explicit
``eps`` stream.  This module is the documented default generator: Philox4x32-10
(Salmon et al., SC'11) keyed by ``seed``, counter = (block index, draw index), followed
by Box-Muller.  ``csrc/philox_host.h`` implements the same function in C++ for callers
that do not go through Python; ``tests/test_rng.py`` pins both against the Random123
known-answer vectors.

Draw order (R/inference-tflow.R:368,372,401,403,447): draw 0 = gamma_init, draw 1 =
initial ELBO, iteration i (1-based) uses draws 2i (train) and 2i+1 (monitor), then the
20 final ELBO evaluations continue the sequence.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(counter, key, rounds=10):
    """counter: uint32 array [..., 4]; key: (k0, k1) -> uint32 array [..., 4]."""
    c = np.asarray(counter, dtype=np.uint32)
    c0, c1, c2, c3 = (c[..., i].astype(np.uint64) for i in range(4))
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & _MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & _MASK, lo0
        with np.errstate(over="ignore"):
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def normal_draw(seed, draw, n):
    """``n`` standard normals (float32) for draw index ``draw`` of stream ``seed``."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    nblk = (n + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint64) & 0xFFFFFFFF
    ctr[:, 1] = np.arange(nblk, dtype=np.uint64) >> 32
    ctr[:, 2] = np.uint32(int(draw) & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((int(draw) >> 32) & 0xFFFFFFFF)
    r = philox4x32(ctr, (seed & 0xFFFFFFFF, seed >> 32)).astype(np.float64)
    u = (r + 0.5) * (1.0 / 4294967296.0)          # (0,1)
    rad0 = np.sqrt(-2.0 * np.log(u[:, 0]))
    rad1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    a0 = 2.0 * np.pi * u[:, 1]
    a1 = 2.0 * np.pi * u[:, 3]
    z = np.stack([rad0 * np.cos(a0), rad0 * np.sin(a0), rad1 * np.cos(a1), rad1 * np.sin(a1)], 1)
    return z.reshape(-1)[:n].astype(np.float32)


class EpsStream:
    """Sequential eps[S,G] draws; mirrors the per-``sess$run`` sampling of the reference."""

    def __init__(self, seed, S, G, start=0):
        self.seed, self.S, self.G, self.draw = int(seed), int(S), int(G), int(start)

    def next(self):
        e = normal_draw(self.seed, self.draw, self.S * self.G).reshape(self.S, self.G)
        self.draw += 1
        return e

    def block(self, n_draws):
        """[n_draws, S, G] float32, advancing the stream."""
        out = np.stack([self.next() for _ in range(n_draws)], 0) if n_draws else \
            np.zeros((0, self.S, self.G), np.float32)
        return out
