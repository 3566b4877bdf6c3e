"""Host-side preparation that the reference does in R before building its graph.

Everything here is O(N*G) one-shot work that stays on the host in the reference too
(SURVEY.md §8 rows a1, a2); the iteration loop itself runs in the HIP engine.
"""
import numpy as np


def softplus(x):
    """R/inference-tflow.R:13-15."""
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64))


def inverse_softplus(x):
    """R/inference-tflow.R:1-3."""
    return np.log(np.exp(np.asarray(x, dtype=np.float64)) - 1.0)


def safe_inverse_softplus(x):
    """R/inference-tflow.R:5-10: ``log(1 - exp(-|x|)) + max(x, 0)``; errors on x < 0."""
    x = np.asarray(x, dtype=np.float64)
    if np.any(x < 0):
        raise ValueError("Inverse softplus only takes positive values")
    with np.errstate(divide="ignore"):
        return np.log(1.0 - np.exp(-np.abs(x))) + np.maximum(x, 0.0)


def saturate(x, threshold=4):
    """R/clonealign.R:394-397: clip copy number above ``threshold``."""
    x = np.array(x, dtype=np.float64, copy=True)
    x[x > threshold] = threshold
    return x


def gene_filter(Y, L, gene_filter_threshold=0):
    """R/inference-tflow.R:117-124: drop genes with ``colSums(Y) <= threshold``.

    Returns (Y_kept, L_kept, keep_mask)."""
    zero_gene_means = Y.sum(0, dtype=np.float64) <= gene_filter_threshold
    keep = ~zero_gene_means
    if keep.all():              # nothing to drop: no copy of the count matrix
        return Y, L, keep
    return np.ascontiguousarray(Y[:, keep]), L[keep, :], keep


def selected_sums(Y, rows, cols, axis):
    """colSums (axis=0) / rowSums (axis=1) of Y[rows][:, cols] in float64 WITHOUT materialising the sub-matrix.
    Integer counts: the full sums minus the sums over the dropped rows / columns when less is dropped than kept (masks
    from preprocessing keep almost everything) -- exact, every partial sum is an integer below 2^53.  Floating-point
    counts are always summed directly over the selection: the subtraction would leave rounding residue (a true 0 coming
    out as 1e-12) exactly where the gene filter compares against its threshold (R/inference-tflow.R:117-124).
    rows / cols: sorted index arrays or None (= all)."""
    N, G = Y.shape
    rows = None if rows is None or len(rows) == N else np.asarray(rows)
    cols = None if cols is None or len(cols) == G else np.asarray(cols)
    exact = Y.dtype.kind in "iu"
    if axis == 0:        # per column, over the selected rows
        if rows is None:
            out = Y.sum(0, dtype=np.float64)
        elif not exact or len(rows) * 2 < N:
            out = Y[rows].sum(0, dtype=np.float64)
        else:
            drop = np.setdiff1d(np.arange(N), rows, assume_unique=True)
            out = Y.sum(0, dtype=np.float64) - Y[drop].sum(0, dtype=np.float64)
        return out if cols is None else out[cols]
    if cols is None:     # per row, over the selected columns
        out = Y.sum(1, dtype=np.float64)
    elif not exact or len(cols) * 2 < G:
        out = Y[:, cols].sum(1, dtype=np.float64)
    else:
        drop = np.setdiff1d(np.arange(G), cols, assume_unique=True)
        out = Y.sum(1, dtype=np.float64) - Y[:, drop].sum(1, dtype=np.float64)
    return out if rows is None else out[rows]


def r_scale(x):
    """R's ``scale(x)``: centre columns, divide by the (n-1) standard deviation."""
    x = np.asarray(x, dtype=np.float64)
    xc = x - x.mean(0, keepdims=True)
    sd = np.sqrt((xc ** 2).sum(0, keepdims=True) / (x.shape[0] - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return xc / sd


def pca_init(Y, K, noise=None):
    """R/inference-tflow.R:204-208.

    ``prcomp(log2(Y+1), center=TRUE, scale=TRUE)$x[, 1:K]`` -> ``scale()`` -> ``+ noise``
    where the reference's noise is ``rnorm(N*K, 0, 0.05)`` filled column-major; here the
    caller supplies ``noise[N,K]`` (already multiplied by 0.05) or None.
    The sign of a principal component is not defined by prcomp (it is whatever LAPACK's SVD returns); here, as in the
    device routine (ca_init_psi_pca), each component is oriented so that its loading of largest magnitude is positive --
    one convention on both sides, so that host- and device-initialised fits of the same seed are the same fit.
    """
    N, G = Y.shape
    K = int(K)
    if K == 0:
        return np.zeros((N, 0))
    X = np.log2(np.asarray(Y, dtype=np.float64) + 1.0)
    Xc = X - X.mean(0, keepdims=True)
    sd = np.sqrt((Xc ** 2).sum(0) / (N - 1))
    if np.any(sd == 0):
        raise ValueError("cannot rescale a constant/zero column to unit variance")
    Xs = Xc / sd
    if N * G <= 4_000_000 or K > 8:
        _, _, Vt = np.linalg.svd(Xs, full_matrices=False)
        V = Vt[:K].T
    else:
        V = _top_eigvecs(Xs, K)
    V = V * np.where(V[np.abs(V).argmax(0), np.arange(V.shape[1])] < 0, -1.0, 1.0)[None, :]
    pcs = r_scale(Xs @ V)
    if noise is not None:
        pcs = pcs + np.asarray(noise, dtype=np.float64).reshape(N, K)
    return pcs


def _top_eigvecs(Xs, K, iters=60, seed=0):
    """Blocked subspace iteration on X^T X for large N*G (same subspace as the SVD)."""
    rng = np.random.default_rng(seed)
    Q = np.linalg.qr(rng.normal(size=(Xs.shape[1], K + 4)))[0]
    for _ in range(iters):
        Q = np.linalg.qr(Xs.T @ (Xs @ Q))[0]
    B = Xs @ Q
    _, _, Wt = np.linalg.svd(B, full_matrices=False)
    return (Q @ Wt.T)[:, :K]


def mu_guess(Y, data_init_mu=True, row_sums=None):
    """R/inference-tflow.R:220-235.  ``row_sums`` (rowSums(Y), float64) saves one pass over a large matrix."""
    G = Y.shape[1]
    if isinstance(data_init_mu, (bool, np.bool_)):
        if data_init_mu:
            if Y.size <= 4_000_000:
                Yd = np.asarray(Y, dtype=np.float64)
                return (Yd / Yd.mean(1, keepdims=True)).mean(0)
            # same quantity, mean_n(y_ng / mean_g' y_ng'), as row-block products in the matrix's own dtype: no N x G
            # float64 temporaries (equal to the expression above up to the rounding of 1/mean, ~1e-16 relative)
            N = Y.shape[0]
            rs = np.asarray(row_sums, dtype=np.float64) if row_sums is not None else Y.sum(1, dtype=np.float64)
            w = G / rs                                   # 1 / rowMeans(Y)
            acc = np.zeros(G)
            step = max(1, 8_000_000 // max(G, 1))
            for i in range(0, N, step):
                acc += w[i:i + step] @ np.asarray(Y[i:i + step], dtype=np.float64)
            return acc / N
        return np.ones(G)
    v = np.asarray(data_init_mu, dtype=np.float64)
    if v.dtype.kind in "fiu" and v.size > 0:
        # the reference's length check is vacuous (``length(data_init_mu == data$G)``)
        return v / v.mean()
    raise ValueError("object 'mu_guess' not found")
