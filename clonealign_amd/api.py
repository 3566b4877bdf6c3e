"""User-facing API: ``clonealign()``, ``run_clonealign()`` and the ``clonealign_fit`` object.

Mirrors ``R/clonealign.R`` (exports in NAMESPACE:3-7) argument for argument; the only
compute-heavy callee, ``inference_tflow``, runs on the MI355X engine.
"""
import string
import warnings
from collections import Counter

import numpy as np

from . import hostprep
from .inference import inference_tflow


class ClonealignFit(dict):
    """The ``clonealign_fit`` S3 list (R/clonealign.R:303): a dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __repr__(self):                                             # print.clonealign_fit, :348-357
        N = len(self["clone"])
        G = len(self["ml_params"]["mu"])
        C = self["ml_params"]["clone_probs"].shape[1]
        return (f"A clonealign_fit for {N} cells, {G} genes, and {C} clones\n"
                "To access clone assignments, call x$clone\n"
                "To access ML parameter estimates, call x$ml_params\n")

    __str__ = __repr__


def clone_assignment(gamma, clone_names, clone_assignment_probability=0.95):
    """R/inference-tflow.R:22-29: arg-max clone (first maximum) or ``"unassigned"``."""
    gamma = np.asarray(gamma)
    best = gamma.argmax(1)
    mx = gamma.max(1)
    names = np.asarray(list(clone_names), dtype=object)
    out = names[best]
    out[mx < clone_assignment_probability] = "unassigned"
    return out


def recompute_clone_assignment(ca, clone_assignment_probability=0.95):
    """R/inference-tflow.R:36-46."""
    ca = ClonealignFit(ca)
    ca["clone"] = clone_assignment(ca["ml_params"]["clone_probs"], ca["clone_names"],
                                   clone_assignment_probability)
    return ca


def compute_correlations(Y, L, clones, clone_names):
    """R/clonealign.R:318-334: per-gene Pearson correlation of scaled counts vs the
    copy number of the assigned clone (NaN where undefined, like R's NA)."""
    Y = np.asarray(Y, dtype=np.float64)
    L = np.asarray(L, dtype=np.float64)
    clones = np.asarray(clones, dtype=object)
    unassigned = clones == "unassigned"
    Y = Y[~unassigned]
    clones = clones[~unassigned]
    idx = {c: i for i, c in enumerate(clone_names)}
    ci = np.array([idx[c] for c in clones], dtype=np.int64)
    G = Y.shape[1]
    out = np.full(G, np.nan)
    if Y.shape[0] < 2:
        return out
    Ys = hostprep.r_scale(Y)
    X = L[:, ci].T                                                  # [n, G]
    with np.errstate(divide="ignore", invalid="ignore"):
        xc = X - X.mean(0, keepdims=True)
        yc = Ys - Ys.mean(0, keepdims=True)
        num = (xc * yc).sum(0)
        den = np.sqrt((xc ** 2).sum(0) * (yc ** 2).sum(0))
        out = np.where(den > 0, num / den, np.nan)
    return out


def correlations_from_sums(T, Syy, L, clone_counts):
    """Pearson r per gene from the device-side sums (ca_clone_gene_sums): x = copy number of the assigned clone,
    y = counts, over assigned cells.  Equals compute_correlations() (R's scale() does not change r)."""
    T = np.asarray(T, dtype=np.float64)
    L = np.asarray(L, dtype=np.float64)
    nc = np.asarray(clone_counts, dtype=np.float64)
    n = nc.sum()
    out = np.full(T.shape[0], np.nan)
    if n < 2:
        return out
    Sx, Sxx = L @ nc, (L ** 2) @ nc
    Sy, Sxy = T.sum(1), (L * T).sum(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        vx = n * Sxx - Sx ** 2
        vy = n * np.asarray(Syy, dtype=np.float64) - Sy ** 2
        den = np.sqrt(vx * vy)
        r = (n * Sxy - Sx * Sy) / den
    # constant x or y (R: NA with a warning); guard against round-off making a zero variance slightly positive
    ok = (vx > 1e-9 * np.maximum(n * Sxx, 1.0)) & (vy > 1e-9 * np.maximum(n * np.asarray(Syy), 1.0))
    out[ok] = r[ok]
    return out


def _counts_array(a):
    """The count matrix in its own dtype when the engine can upload it as it is (no float64 copy of N x G), else float64."""
    a = np.asarray(a)
    if a.dtype in (np.float64, np.float32, np.int32, np.uint16, np.uint8):
        return a
    if a.dtype.kind in "iu" and a.size and 0 <= a.min() and a.max() <= np.iinfo(np.int32).max:
        return a.astype(np.int32)            # e.g. numpy's default int64 counts: half the bytes of a float64 copy
    return a.astype(np.float64)


def _parse_expression(gene_expression_data):
    """R/clonealign.R:207-222.  Returns (Y[cells,genes], gene_names or None)."""
    g = gene_expression_data
    if hasattr(g, "assays") or (isinstance(g, dict) and "assays" in g):        # SCE / SE stand-in
        assays = g["assays"] if isinstance(g, dict) else g.assays
        if "counts" not in assays:
            raise ValueError("counts not in assays(gene_expression_data). Available assays: "
                             + ",".join(assays))
        counts = assays["counts"]                                    # genes x cells
        names = None
        if hasattr(counts, "index"):
            names = [str(i) for i in counts.index]
        elif isinstance(g, dict) and "rownames" in g:
            names = list(g["rownames"])
        return _counts_array(counts).T, names
    if hasattr(g, "columns") and hasattr(g, "values"):               # pandas DataFrame cells x genes
        return _counts_array(g.values), [str(c) for c in g.columns]
    if isinstance(g, np.ndarray) and g.ndim == 2:
        return _counts_array(g), None
    raise TypeError("Input gene_expression_data must be SingleCellExperiment, SummarizedExperiment, or matrix")


def _parse_cnv(copy_number_data):
    """R/clonealign.R:237-243.  Returns (L[genes,clones], clone_names or None)."""
    c = copy_number_data
    if hasattr(c, "columns") and hasattr(c, "values"):
        return np.asarray(c.values, dtype=np.float64), [str(n) for n in c.columns]
    if isinstance(c, np.ndarray) and c.ndim == 2:
        return np.asarray(c, dtype=np.float64), None
    raise TypeError("copy_number_data must be a matrix, data.frame or DataFrame. Current class: "
                    + type(c).__name__)


def _default_gene_names(G):
    # R/clonealign.R:256-258 uses ``letters`` (only 26 of them); beyond that we number.
    lt = string.ascii_lowercase
    return [f"gene_{lt[i]}" if i < 26 else f"gene_{i + 1}" for i in range(G)]


def clonealign(gene_expression_data, copy_number_data, max_iter=200, rel_tol=1e-6,
               gene_filter_threshold=0, learning_rate=0.1, x=None, clone_allele=None,
               cov=None, ref=None, fix_alpha=False, dtype="float32", saturate=True,
               saturation_threshold=6, K=None, mc_samples=1, verbose=True, initial_shrink=5,
               clone_call_probability=0.95, data_init_mu=True, *, seed=None, engine=None,
               engine_opts=None, clone_names=None, allele_ref="cov", cell_index=None, gene_index=None, devices=None, _reuse=None):
    """Assign scRNA-seq cells to clones.  Arguments as R/clonealign.R:184-203.

    Keyword-only extras: ``allele_ref`` -- "cov" (default) reproduces the reference, which forwards ``ref = cov`` to
    inference_tflow (R/clonealign.R:271) so that the allele-specific term sees alt = 0; "ref" forwards the caller's ``ref``
    (the evident intent: an explicit opt-in, the default stays reference-identical).  ``cell_index`` / ``gene_index``: masks or
    index arrays from ``preprocess_for_clonealign(..., return_masks=True)``; the raw matrix is then fitted on that selection
    without a filtered copy (``copy_number_data`` etc. are given for the selected genes / cells).  ``devices``: HIP ordinals -- this ONE fit
    cell-sharded over those devices of this process (forwarded untouched to ``inference_tflow``; ``run_clonealign(devices=)`` is the
    other thing: independent restarts dealt over devices)."""
    if allele_ref not in ("cov", "ref"):
        raise ValueError("allele_ref must be 'cov' (reference behaviour) or 'ref'")
    Y, gene_names = _parse_expression(gene_expression_data)
    N, G = Y.shape
    sel_g = None if gene_index is None else (np.flatnonzero(np.asarray(gene_index)) if np.asarray(gene_index).dtype == bool
                                             else np.asarray(gene_index, dtype=np.int64))
    sel_c = None if cell_index is None else (np.flatnonzero(np.asarray(cell_index)) if np.asarray(cell_index).dtype == bool
                                             else np.asarray(cell_index, dtype=np.int64))
    if sel_g is not None:
        G = len(sel_g)
        if gene_names is not None:
            gene_names = [gene_names[i] for i in sel_g]
    if sel_c is not None:
        N = len(sel_c)
    if K is None:
        K = 1                                                        # :226-232 (both branches give 1)
    L, cn = _parse_cnv(copy_number_data)
    if L.shape[0] != G:
        raise ValueError("copy_number_data must have same number of genes (rows) as gene_expression_data")
    C = L.shape[1]
    if clone_names is None:
        clone_names = cn
    if clone_names is None:
        clone_names = [f"clone_{string.ascii_lowercase[i]}" for i in range(C)]   # :251-253
    if gene_names is None:
        gene_names = _default_gene_names(G)
    # NB the reference forwards ``ref = cov`` (R/clonealign.R:271); kept for drop-in behaviour
    def _post(eng, rlist):
        # device-side sums for compute_correlations: no second pass over Y on the host
        if not hasattr(eng, "clone_gene_sums"):
            return None
        labels = clone_assignment(rlist["clone_probs"], clone_names, clone_call_probability)
        lut = {c: i for i, c in enumerate(clone_names)}
        idx = np.array([lut.get(c, -1) for c in labels], dtype=np.int32)
        T, Syy = eng.clone_gene_sums(idx)
        return dict(T=T, Syy=Syy, counts=np.bincount(idx[idx >= 0], minlength=C))

    res = inference_tflow(Y, L, max_iter=max_iter, rel_tol=rel_tol, learning_rate=learning_rate,
                          gene_filter_threshold=gene_filter_threshold, x=x,
                          clone_allele=clone_allele, cov=cov, ref=(cov if allele_ref == "cov" else ref), fix_alpha=fix_alpha,
                          dtype=dtype, saturate=saturate, saturation_threshold=saturation_threshold,
                          K=K, mc_samples=mc_samples, verbose=verbose, initial_shrink=initial_shrink,
                          data_init_mu=data_init_mu, gene_names=gene_names, seed=seed,
                          engine=engine, engine_opts=engine_opts, post=_post, cell_index=sel_c, gene_index=sel_g, devices=devices,
                          _reuse=_reuse)
    res = ClonealignFit(res)
    res["clone"] = clone_assignment(res["ml_params"]["clone_probs"], clone_names,
                                    clone_call_probability)          # :283
    res["clone_names"] = list(clone_names)                           # colnames(clone_probs), :286
    # the retained genes as the boolean mask inference_tflow applied (matching by NAME would mark a filtered gene that shares
    # its symbol with a retained one as kept, and L[keep] would then have more rows than the fitted matrix)
    keep = np.asarray(res.pop("retained_mask"), dtype=bool)
    post = res.pop("post", None)
    if post is not None:
        res["correlations"] = correlations_from_sums(post["T"], post["Syy"], L[keep, :], post["counts"])   # :292-294
    else:
        Ysel = Y if (sel_c is None and sel_g is None) else Y[np.ix_(np.arange(Y.shape[0]) if sel_c is None else sel_c,
                                                                    np.arange(Y.shape[1]) if sel_g is None else sel_g)]
        res["correlations"] = compute_correlations(Ysel[:, keep], L[keep, :], res["clone"], clone_names)   # :292-294
    cor = res["correlations"]
    if np.any(~np.isnan(cor)):
        if np.nanquantile(cor, 0.25) < 0:                            # :296-300
            warnings.warn("Less than 75% of genes positively correlated with expression - "
                          "assignment may have failed\n")
    return res


def run_clonealign(gene_expression_data, copy_number_data, initial_shrinks=(0, 5, 10),
                   n_repeats=3, print_elbos=True, *, seed=None, devices=None, **kwargs):
    """R/clonealign.R:35-75: fit across restarts and keep the best final ELBO.

    ``devices``: list of GPU ordinals; restarts are dealt round-robin over them (one fit per
    GPU at a time, independent replicas -- SURVEY.md §8e config 5).  Default: device 0.
    """
    from .multirun import run_restarts
    jobs = []
    ss = np.random.SeedSequence(seed)
    for is_ in initial_shrinks:
        for _ in range(int(n_repeats)):
            jobs.append(dict(initial_shrink=is_))
    seeds = [int(s.generate_state(1)[0]) for s in ss.spawn(len(jobs))]
    fits = run_restarts(gene_expression_data, copy_number_data, jobs, seeds, devices, kwargs)
    final_elbos = np.array([f["convergence_info"]["final_elbo"] for f in fits])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        median_correlations = np.array([np.nanmedian(f["correlations"]) if np.any(~np.isnan(f["correlations"]))
                                        else np.nan for f in fits])
    if print_elbos:
        print("ELBOs:  " + " ".join(repr(float(e)) for e in final_elbos))       # :61-63
    best = fits[int(np.nanargmax(final_elbos))]                                   # which.max, :65
    best["multirun_info"] = {
        "clone_prevalences_at_different_shrinks": [dict(Counter(f["clone"])) for f in fits],   # :69
        "elbos": final_elbos,
        "median_correlations": median_correlations,
    }
    return best
