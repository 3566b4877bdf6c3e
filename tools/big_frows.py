"""The device-side f-rows on a matrix beyond 2^32 elements: preprocessing sums (ca_preprocess), per-clone gene sums (ca_clone_gene_sums),
device PCA init (ca_init_psi_pca): exact / close against numpy on the host.  python tools/big_frows.py (needs ~60 GB of host memory)"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from clonealign_amd import engine as E
N, G, C = 900_000, 5000, 4
rng = np.random.default_rng(9)
L = rng.integers(1, 5, size=(G, C)).astype(np.float64)
lib = rng.lognormal(0, 0.6, size=(N, 1))
Y = rng.poisson(0.7 * lib * rng.lognormal(0, 1, size=(1, G))).astype(np.int32)   # (library-size and gene effects: a first PC to find)
Y[-5:, :] += 3
print("elements %.2e" % Y.size, flush=True)
t0 = time.time()
kg, kc, gs, cs = E.preprocess_masks(Y, L)
print("ca_preprocess %.1f s: gene sums %s, cell sums %s" % (time.time() - t0, "exact" if np.array_equal(gs, Y.sum(0, dtype=np.float64)) else "MISMATCH",
                                                         "exact" if np.array_equal(cs[kc], Y[:, kg].sum(1, dtype=np.float64)[kc]) or np.array_equal(cs, Y[:, kg].sum(1, dtype=np.float64)) else "MISMATCH"), flush=True)
eng = E.HipEngine(Y, L, rng.normal(size=(N, 1)), np.zeros(G) + 0.5, 1)
z = rng.integers(-1, C, size=N).astype(np.int32)
T, Syy = eng.clone_gene_sums(z)
Tw = np.stack([Y[z == c].sum(0, dtype=np.float64) for c in range(C)], 1)
Sw = (Y[z >= 0].astype(np.float64) ** 2).sum(0)
print("ca_clone_gene_sums: T %s, Syy max rel %.1e" % ("exact" if np.array_equal(T, Tw) else "MISMATCH %.3e" % np.abs(T - Tw).max(), np.abs(Syy - Sw).max() / Sw.max()), flush=True)
psi = eng.pca_init(None, 40, 1)
if psi is not None:
    r = np.corrcoef(psi[:, 0], np.log(Y.sum(1, dtype=np.float64)))[0, 1]
    print("ca_init_psi_pca: finite %s, mean %.2e, sd %.3f, |correlation| of PC1 with the log library size %.3f" % (bool(np.all(np.isfinite(psi))), float(psi.mean()), float(psi.std()), abs(r)), flush=True)
eng.close()
