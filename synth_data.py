"""Seeded synthetic clonealign problems.

Generator: L_gc ~ categorical over {1,2,3,4} with the example's weights (87/114/68/31 of
300), rows with identical copy number in every clone re-drawn; true clone z_n uniform; gene
base rate mu_g ~ LogNormal(0, 1.6) (the spread of the example's per-gene means); library
size s_n ~ LogNormal(log median, 0.5) clipped at >= 200 (>= 20 for tiny medians);
counts y_ng ~ Poisson(s_n * mu_g L_{g z_n} / sum_g mu_g L_{g z_n}) -- the Poissonised
multinomial, so that rows can be generated independently (and on the GPU).
"""
import numpy as np

CN_VALUES = np.array([1.0, 2.0, 3.0, 4.0])
CN_WEIGHTS = np.array([87.0, 114.0, 68.0, 31.0]) / 300.0

CONFIGS = {
    # BASELINE.json "configs"
    "cfg1": dict(N=200, G=100, C=3, median_s=71),
    "cfg2": dict(N=10_000, G=2_000, C=4, median_s=2000),
    "cfg3": dict(N=100_000, G=5_000, C=8, median_s=2000),
    "cfg5": dict(N=50_000, G=3_000, C=6, median_s=2000),
}


def make_copy_number(G, C, rng):
    L = rng.choice(CN_VALUES, size=(G, C), p=CN_WEIGHTS)
    if C > 1:
        for _ in range(64):
            same = (L == L[:, :1]).all(1)
            if not same.any():
                break
            L[same] = rng.choice(CN_VALUES, size=(int(same.sum()), C), p=CN_WEIGHTS)
    return L


def make_globals(N, G, C, seed, median_s=2000):
    """Everything except the counts: (L, mu, z, s) -- small arrays."""
    rng = np.random.default_rng(seed)
    L = make_copy_number(G, C, rng)
    mu = rng.lognormal(0.0, 1.6, size=G)
    z = rng.integers(0, C, size=N)
    s = np.maximum(rng.lognormal(np.log(median_s), 0.5, size=N), 200.0 if median_s >= 400 else 20.0)
    return L, mu, z, s


def make_problem(N, G, C, seed=20240, median_s=2000, rows=None):
    """numpy generator.  Returns dict(Y int32 [n,G], L [G,C], z, s_target, psi0, loc0).

    ``rows``: optional (start, stop) to generate only a shard of the cells (same data as the
    full matrix restricted to those rows: the row RNG is keyed by the row block)."""
    L, mu, z, s = make_globals(N, G, C, seed, median_s)
    lo, hi = (0, N) if rows is None else rows
    M = mu[:, None] * L                                   # [G,C]
    P = M / M.sum(0, keepdims=True)
    Y = np.empty((hi - lo, G), dtype=np.int32)
    blk = 4096
    for b0 in range(lo - lo % blk, hi, blk):
        r = np.random.default_rng([seed, 1, b0 // blk])
        b1 = min(b0 + blk, N)
        lam = s[b0:b1, None] * P[:, z[b0:b1]].T
        yb = r.poisson(lam).astype(np.int32)
        a, b = max(b0, lo), min(b1, hi)
        Y[a - lo:b - lo] = yb[a - b0:b - b0]
    Y[:, 0] += (Y.sum(1) == 0)                            # no empty cells
    return dict(Y=Y, L=L, z=z[lo:hi], s_target=s[lo:hi], mu_true=mu)


def make_hard_problem(N, G, C, seed=20246, median_s=400, informative=0.03):
    """A problem on which clone assignment is NOT trivially separable (on the default generator every max-gamma ends at exactly 1.0,
    so "0 label flips" said nothing): shallow libraries (median 400 counts per cell, the example's order of magnitude, R/data example_sce: 71) and a
    copy-number profile that is THE SAME in every clone except on a few percent of the genes -- what a real tumour's subclones look like.  After the
    default fit roughly half of the cells stay below the 0.95 call threshold, about a quarter end with max-gamma in [0.9, 0.99), and tens of cells sit
    within 1e-3 of the threshold (tests/test_gpu_scale.py::test_full_default_fit_on_a_hard_problem_* print the numbers).  Returns dict(Y int32 [N, G'], L [G', C], z) with empty genes removed."""
    rng = np.random.default_rng(seed)
    base = rng.choice(CN_VALUES, size=(G, 1), p=CN_WEIGHTS)
    L = np.repeat(base, C, 1)
    inf = rng.choice(G, max(C, int(round(informative * G))), replace=False)
    L[inf] = make_copy_number(len(inf), C, rng)
    mu = rng.lognormal(0.0, 1.6, size=G)
    z = rng.integers(0, C, size=N)
    s = np.maximum(rng.lognormal(np.log(median_s), 0.5, size=N), 20.0)
    M = mu[:, None] * L
    P = M / M.sum(0, keepdims=True)
    Y = np.empty((N, G), dtype=np.int32)
    for b0 in range(0, N, 4096):
        b1 = min(b0 + 4096, N)
        Y[b0:b1] = np.random.default_rng([seed, 2, b0 // 4096]).poisson(s[b0:b1, None] * P[:, z[b0:b1]].T)
    Y[:, 0] += (Y.sum(1) == 0)
    keep = Y.sum(0) > 0
    return dict(Y=np.ascontiguousarray(Y[:, keep]), L=L[keep], z=z, informative_genes=int(np.isin(np.flatnonzero(keep), inf).sum()))


def cheap_init(Y, K=1, seed=0):
    """Initialisation for benchmarks: psi0 ~ N(0,1) (the iteration rate does not depend on
    it; prcomp at 100k x 5k would dwarf the loop) and the reference's
    data-driven loc0 (R/inference-tflow.R:220-235,262)."""
    from clonealign_amd.hostprep import mu_guess, safe_inverse_softplus
    rng = np.random.default_rng(seed)
    psi0 = rng.normal(size=(Y.shape[0], K))
    loc0 = safe_inverse_softplus(np.maximum(mu_guess(Y, True), 1e-6))
    return psi0, loc0


def make_problem_torch(N, G, C, seed=20240, median_s=2000, rows=None, device="cuda"):
    """Same distribution, counts drawn with torch on the GPU (no PCIe, no host pass over N*G).
    Returns (Y_dev int32 torch tensor [n,G], aux dict with L, z, loc0, colsum...)."""
    import torch
    L, mu, z, s = make_globals(N, G, C, seed, median_s)
    lo, hi = (0, N) if rows is None else rows
    M = mu[:, None] * L
    P = torch.tensor((M / M.sum(0, keepdims=True)).T, dtype=torch.float32, device=device)   # [C,G]
    zt = torch.tensor(z[lo:hi], device=device)
    st = torch.tensor(s[lo:hi], dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 1000003 + lo)
    Y = torch.empty((hi - lo, G), dtype=torch.int32, device=device)
    blk = 8192
    for b0 in range(0, hi - lo, blk):
        b1 = min(b0 + blk, hi - lo)
        lam = st[b0:b1, None] * P[zt[b0:b1]]
        Y[b0:b1] = torch.poisson(lam, generator=gen).to(torch.int32)
    empty = (Y.sum(1) == 0)
    Y[:, 0] += empty.to(torch.int32)
    # the matrix is handed to the engine as a raw device pointer and read on the ENGINE's stream: the generator's kernels (torch's
    # stream) must have finished -- an engine built straight after this call scanned a half-written matrix now and then and chose
    # float32 storage for it (round 5: tools/stair_time.py showed 120 us iterations out of nowhere)
    torch.cuda.synchronize(device)
    return Y, dict(L=L, z=z[lo:hi], s_target=s[lo:hi], mu_true=mu)
