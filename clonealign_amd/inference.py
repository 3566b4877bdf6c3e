"""``inference_tflow`` -- host mirror of the reference's only inference routine.

Reference: ``R/inference-tflow.R:71-481`` (signature ``:71-89``, man/inference_tflow.Rd).
The R-side preparation (gene filter, saturation, PCA / mu initialisation) is restated in
``hostprep.py``; the TensorFlow graph + session loop (``:240-457``) is replaced by the HIP
engine behind the C ABI of ``include/clonealign_hip.h``.  The function keeps the
reference's name, argument names, defaults, return structure and error messages so the
reference's own tests (tests/testthat/test_clonealign.R) translate line by line.
"""
import math

import numpy as np
from scipy.special import gammaln, logsumexp

from . import hostprep
from .rng import EpsStream

N_FINAL_ELBO = 20  # R/inference-tflow.R:447


def beta_binomial_log_prob(k, n, alpha, beta):
    """R/allele-specific.R:52-58."""
    ll = gammaln(n + 1) - gammaln(k + 1) - gammaln(n - k + 1)
    ll = ll + gammaln(k + alpha) + gammaln(n - k + beta) - gammaln(alpha + beta + n)
    return ll - gammaln(alpha) - gammaln(beta) + gammaln(alpha + beta)


def construct_ai_likelihood(clone_allele, alt, cov):
    """R/allele-specific.R:17-48.  clone_allele [V,C], alt/cov [V,N] -> [N,C] (parameter free)."""
    p1_low = math.log(0.5) + beta_binomial_log_prob(alt, cov, 0.1, 1.9)
    p1_high = math.log(0.5) + beta_binomial_log_prob(alt, cov, 1.9, 0.1)
    p1 = np.logaddexp(p1_low, p1_high)                      # [V,N]
    p2 = beta_binomial_log_prob(alt, cov, 2.0, 2.0)
    is2 = (np.asarray(clone_allele) == 2)                   # [V,C]
    Lcvn = np.where(is2.T[:, :, None], p2[None], p1[None])  # [C,V,N]
    return Lcvn.sum(1).T                                    # [N,C]


def sanitize_allele_info(V, clone_allele, cov, ref, N, C):
    """R/allele-specific.R:61-70."""
    assert clone_allele.shape[1] == C
    assert cov.shape[0] == N and ref.shape[0] == N
    assert ref.shape[1] == V and cov.shape[1] == V


def _default_engine_factory(group=False):
    from .engine import HipEngine, HipGroupEngine  # fails loudly when the HIP library is missing
    return HipGroupEngine if group else HipEngine


def run_vi_loop(eng, eps, max_iter, rel_tol, verbose=False):
    """The session loop of R/inference-tflow.R:368-417 driven call by call.

    ``eng`` exposes gamma_init/elbo/step; ``eps.next()`` yields eps[S,G] per ``sess$run``."""
    eng.gamma_init(eps.next())                                   # :368-369
    elbo_val = eng.elbo(eps.next())                              # :372
    if math.isnan(elbo_val):
        raise FloatingPointError("Initial elbo is NA")           # :374-376
    elbo_diffs = [1e3] * 10                                      # :379
    elbos = [elbo_val]
    for _ in range(int(max_iter)):                               # :394
        eng.step(eps.next())                                     # :401
        elbo_new = eng.elbo(eps.next())                          # :403
        elbo_diff = (elbo_new - elbo_val) / abs(elbo_val)
        elbo_diffs = elbo_diffs[1:] + [elbo_diff]
        elbos.append(elbo_new)
        elbo_val = elbo_new
        mean_change = float(np.mean(np.abs(elbo_diffs)))
        if math.isnan(mean_change):
            raise FloatingPointError("missing value where TRUE/FALSE needed")  # R's `if (NA)` at :414
        if mean_change < rel_tol:                                # :414
            break
    return elbos


def inference_tflow(Y_dat, L_dat, max_iter=100, rel_tol=1e-5, learning_rate=0.1,
                    gene_filter_threshold=0, x=None, clone_allele=None, cov=None, ref=None,
                    fix_alpha=False, dtype="float32", saturate=True, saturation_threshold=6,
                    K=1, mc_samples=1, verbose=True, initial_shrink=5, data_init_mu=True,
                    *, gene_names=None, seed=None, engine=None, engine_opts=None,
                    psi_noise=None, eps_stream=None, psi_init="auto", post=None, allele_on="auto", cell_index=None,
                    gene_index=None, devices=None, _reuse=None):
    """EM/VI inference on the MI355X engine.  Arguments as R/inference-tflow.R:71-89.

    Keyword-only extras (no reference counterpart): ``seed`` (replaces R's ``set.seed``
    state feeding ``rnorm`` at :208 and ``get_next_seed`` at :269), ``gene_names``
    (``colnames(Y_dat)``), ``engine`` (engine class; default the HIP engine),
    ``psi_noise`` / ``eps_stream`` to inject the two noise sources explicitly, ``psi_init`` in
    {"auto", "host", "device"}: where the PCA initialisation of :204-208 runs ("auto": on the device, by
    subspace iteration over the resident count matrix, once N*G exceeds 4e6; exact SVD on the host below that);
    ``allele_on`` in {"auto", "host", "device"}: where the parameter-free allele term of :166-187 is evaluated ("auto": on
    the device, ca_allele_loglik, once cells x variants exceeds 2e5);
    ``post(engine, ml_params)`` runs before the engine is closed (clonealign() uses it for the device-side
    correlation sums) and its result is returned under ``"post"``.
    ``cell_index`` / ``gene_index`` (sorted integer arrays or boolean masks): fit only these rows / columns of ``Y_dat`` --
    the masks of ``preprocess_for_clonealign(..., return_masks=True)``.  ``L_dat`` (and ``x``, ``cov``, ``ref``) are then given
    for the SELECTED genes / cells.  Above 4e6 selected counts the HIP engine takes the raw matrix and the index lists
    (``ca_problem.cell_index / gene_index``): no filtered copy of the matrix is made on the host, neither here nor for the gene
    filter of :117-124 (the reference copies in R: R/preprocess.R:141-147, R/inference-tflow.R:117-124).
    ``devices``: HIP ordinals, e.g. ``range(8)`` -- ONE fit cell-sharded over these devices of this process (SURVEY.md section 8b/8e,
    BASELINE configs[3]): rank r holds the cells ``sharding.cell_range(N, r, W)``, one worker thread and one engine handle per device,
    joined by the first transport that passes its known-answer test (peer-to-peer by address -> RCCL -> host reduction;
    ``engine_opts={"transport": ...}`` insists on one).  The return value is the one-device fit's for all cells (trace to the
    grouping of the fp64 cell sums).  One ordinal, or None, is the plain single-device fit on that device.
    ``_reuse``: a dict owned by run_clonealign()'s restart loop (multirun.py).  The first fit leaves its prepared inputs and
    its engine in it; later fits on the SAME data and settings skip the host passes and the upload and restart the resident
    engine (``ca_reinit``).  The owner closes the engine.
    """
    log = (lambda m: print(m)) if verbose else (lambda m: None)
    log("Constructing HIP engine")                               # :102-104 ("Constructing tensorflow graph")
    if dtype not in ("float32", "float64"):
        raise ValueError("'arg' should be one of 'float32', 'float64'")   # match.arg, :112
    if psi_init not in ("auto", "host", "device"):
        raise ValueError("psi_init must be 'auto', 'host' or 'device'")
    if dtype == "float64":
        raise NotImplementedError(
            "dtype='float64': the reference graph cannot be built for float64 "
            "(R/inference-tflow.R:323 divides a float64 tensor by tf$to_float(S)); only float32 is supported")
    if devices is not None:
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one device")
        if engine is not None:
            raise ValueError("devices selects the HIP engine's device group; it cannot be combined with engine=")
        if int((engine_opts or {}).get("world", 1)) != 1:
            raise ValueError("devices shards the fit inside this process; rank/world (one process per GPU) are the other way to shard")
        engine_opts = dict(engine_opts or {})
        if len(devices) == 1:
            engine_opts["device"] = devices[0]
            devices = None
        else:
            engine_opts["devices"] = devices
    cached = None if _reuse is None else _reuse.get("prep")
    if cached is None:
        Y_dat = np.asarray(Y_dat)
        if Y_dat.dtype not in (np.float64, np.float32, np.int32, np.uint16, np.uint8):   # dtypes the engine uploads as they are
            fits = Y_dat.dtype.kind in "iu" and Y_dat.size and 0 <= Y_dat.min() and Y_dat.max() <= np.iinfo(np.int32).max
            Y_dat = Y_dat.astype(np.int32 if fits else np.float64)
        L_dat = np.asarray(L_dat, dtype=np.float64)
        # optional selection of rows / columns of the raw matrix (masks of preprocess_for_clonealign)
        as_index = lambda m, n: None if m is None else (np.flatnonzero(np.asarray(m)) if np.asarray(m).dtype == bool  # noqa: E731
                                                        else np.asarray(m, dtype=np.int64))
        ci, gi = as_index(cell_index, Y_dat.shape[0]), as_index(gene_index, Y_dat.shape[1])
        n_sel = Y_dat.shape[0] if ci is None else len(ci)
        g_sel = Y_dat.shape[1] if gi is None else len(gi)
        if L_dat.shape[0] != g_sel:
            raise ValueError("nrow(L_dat) == G is not TRUE")               # :139
        device_cut = (engine is None and n_sel * g_sel > 4_000_000 and
                      int((engine_opts or {}).get("world", 1)) == 1)      # the engine cuts the raw matrix at upload
        sel = None
        if device_cut:
            col = hostprep.selected_sums(Y_dat, ci, gi, axis=0)            # colSums over the selected cells, selected genes
            keep = ~(col <= gene_filter_threshold)                         # :117-124 on the selection
            gi_full = np.arange(Y_dat.shape[1]) if gi is None else gi
            if not keep.all() or ci is not None or gi is not None:
                sel = dict(cell_index=ci, gene_index=gi_full[keep].astype(np.int32) if (gi is not None or not keep.all()) else None)
                if sel["cell_index"] is None and sel["gene_index"] is None:
                    sel = None
            L_dat = L_dat[keep, :]
            if psi_init == "host" and K > 0:
                # the caller insists on the exact host SVD (:204-208): it needs the filtered matrix on the host after all
                rows = np.arange(Y_dat.shape[0]) if ci is None else ci
                Y_dat = Y_dat[np.ix_(rows, gi_full[keep])]
                device_cut, sel, row_sums = False, None, None
            else:
                row_sums = hostprep.selected_sums(Y_dat, ci, None if sel is None else sel["gene_index"], axis=1)
        else:
            if ci is not None or gi is not None:
                Y_dat = Y_dat[np.ix_(np.arange(Y_dat.shape[0]) if ci is None else ci, np.arange(Y_dat.shape[1]) if gi is None else gi)]
            Y_dat, L_dat, keep = hostprep.gene_filter(Y_dat, L_dat, gene_filter_threshold)   # :117-124
            row_sums = None
        log(f"Removing {int((~keep).sum())} genes with low counts")
        if gene_names is not None:
            retained_genes = [g for g, k in zip(gene_names, keep) if k]     # :126-131
        else:
            retained_genes = np.flatnonzero(keep)                           # 0-based (R: which(), 1-based)
        N, G = (n_sel, int(keep.sum())) if device_cut else Y_dat.shape
        C = L_dat.shape[1]
        K = int(K)
        if L_dat.shape[0] != G:
            raise ValueError("nrow(L_dat) == G is not TRUE")               # :139
        if saturate:
            L_dat = hostprep.saturate(L_dat, saturation_threshold)          # :142-144
        P = 0
        if x is not None:                                                   # :147-153
            x = np.asarray(x, dtype=np.float64)
            if x.ndim == 1:
                x = x.reshape(-1, 1)
            if x.ndim != 2:
                raise ValueError("is.matrix(x) is not TRUE")
            P = x.shape[1]
            if x.shape[0] != N:
                raise ValueError("nrow(x) == N is not TRUE")
        # allelic imbalance (:166-187): a parameter-free [N,C] additive term
        use_allele = clone_allele is not None and ref is not None and cov is not None
        extra = None
        clone_probs_from_snv = None
        if use_allele:
            log("Using allelic imbalance info")
            clone_allele = np.asarray(clone_allele, dtype=np.float64)
            cov = np.asarray(cov, dtype=np.float64)
            ref = np.asarray(ref, dtype=np.float64)
            V = clone_allele.shape[0]
            sanitize_allele_info(V, clone_allele, cov, ref, N, C)
            if allele_on in ("device", "auto") and engine is None and (allele_on == "device" or N * V > 200_000):
                from .engine import allele_loglik                          # 12 lgamma per (variant, cell)
                dev = int((engine_opts or {}).get("device", (devices or [0])[0]))
                extra = allele_loglik(clone_allele, cov, ref, device=dev)  # [N,C]
            else:
                alt = cov.T - ref.T
                extra = construct_ai_likelihood(clone_allele, alt, cov.T)  # [N,C]
            clone_probs_from_snv = np.exp(extra - logsumexp(extra, 1, keepdims=True))   # :436-440
        if _reuse is not None:
            _reuse["prep"] = dict(Y_dat=Y_dat, L_dat=L_dat, keep=keep, retained_genes=retained_genes, N=N, G=G, C=C, K=K, P=P,
                                  x=x, extra=extra, clone_probs_from_snv=clone_probs_from_snv, sel=sel, row_sums=row_sums,
                                  device_cut=device_cut)
    else:
        Y_dat, L_dat, keep, retained_genes = cached["Y_dat"], cached["L_dat"], cached["keep"], cached["retained_genes"]
        N, G, C, K, P, x = cached["N"], cached["G"], cached["C"], cached["K"], cached["P"], cached["x"]
        extra, clone_probs_from_snv = cached["extra"], cached["clone_probs_from_snv"]
        sel, row_sums, device_cut = cached["sel"], cached["row_sums"], cached["device_cut"]
    rng = np.random.default_rng(seed)
    # initialisation (:204-235)
    if psi_noise is None:
        psi_noise = rng.normal(0.0, 0.05, size=(K, N)).T if K > 0 else np.zeros((N, 0))  # column-major fill
    Engine = engine if engine is not None else _default_engine_factory(group=devices is not None)
    # a matrix that is cut on the device (decided on the size BEFORE the gene filter) takes both device-side initialisations,
    # whatever is left after the filter: the filtered copy does not exist on the host
    big = N * G > 4_000_000 or device_cut
    device_pca = K > 0 and hasattr(Engine, "pca_init") and (psi_init == "device" or (psi_init == "auto" and big))
    pcs = np.zeros((N, K)) if device_pca else hostprep.pca_init(Y_dat, K, psi_noise)
    if cached is not None and "loc0" in cached:
        loc0 = cached["loc0"]                                           # same data: same s_init check, same mu_guess
    else:
        s_init = row_sums if row_sums is not None else Y_dat.sum(1, dtype=np.float64)
        if np.any(s_init == 0):
            raise ValueError("Some cells have no counts mapping")      # :212-214
        if (isinstance(data_init_mu, (bool, np.bool_)) and bool(data_init_mu) and big
                and getattr(Engine, "DEVICE_MU_INIT", False) and int((engine_opts or {}).get("world", 1)) == 1):
            loc0 = None  # the engine takes mu_guess (:220-235) and loc0 (:262) from the resident matrix: no host pass
        else:
            if device_cut and isinstance(data_init_mu, (bool, np.bool_)) and bool(data_init_mu):
                raise ValueError("data_init_mu=True on a device-cut matrix needs the engine's device-side initialisation")
            mu_g = hostprep.mu_guess(np.empty((0, G)) if device_cut else Y_dat, data_init_mu, row_sums=s_init)
            loc0 = hostprep.safe_inverse_softplus(mu_g)                 # :262
        if _reuse is not None:
            _reuse["prep"]["loc0"] = loc0
    S = int(mc_samples)
    if eps_stream is None:
        eps_seed = int(rng.integers(1, 2**31 - 1))                      # get_next_seed(), :49-51
        eps_stream = EpsStream(eps_seed, S, G)

    eng = None if _reuse is None else _reuse.get("eng")
    if eng is not None:
        eng.reinit(pcs, loc0)                                           # restart on the resident data (ca_reinit)
    else:
        eng = Engine(Y_dat, L_dat, pcs, loc0, K, S, X=x, extra_loglik=extra,
                     learning_rate=learning_rate, **(engine_opts or {}), **(sel or {}))
        if _reuse is not None and hasattr(eng, "reinit"):
            _reuse["eng"] = eng
    keep_open = _reuse is not None and _reuse.get("eng") is eng
    try:
        if device_pca:
            if cached is not None and "const_gene" in cached:
                const_gene = cached["const_gene"]
            else:   # (a device-cut matrix is checked by the device routine itself: same message from ca_init_psi_pca)
                const_gene = False if device_cut else bool(np.any(Y_dat.min(0) == Y_dat.max(0)))
                if _reuse is not None:
                    _reuse["prep"]["const_gene"] = const_gene
            if const_gene:                               # prcomp(scale = TRUE) refuses constant genes (sd == 0)
                raise ValueError("cannot rescale a constant/zero column to unit variance")
            try:
                eng.pca_init(psi_noise, seed=int(rng.integers(0, 2**31 - 1)))
            except RuntimeError as e:
                if "constant/zero column" in str(e):
                    raise ValueError("cannot rescale a constant/zero column to unit variance") from e
                raise
        log("Optimizing ELBO")
        if hasattr(eng, "run"):
            elbos = eng.run(eps_stream, max_iter, rel_tol)
        else:
            elbos = run_vi_loop(eng, eps_stream, max_iter, rel_tol, verbose)
        log("\nELBO converged or reached max iterations")
        rlist = eng.get_params()                                        # :424-434
        post_out = post(eng, rlist) if post is not None else None
        log("Computing final ELBO")
        if hasattr(eng, "final_elbo"):
            final = eng.final_elbo(eps_stream, N_FINAL_ELBO)
        else:
            final = [eng.elbo(eps_stream.next()) for _ in range(N_FINAL_ELBO)]   # :447-449
    except BaseException:
        if keep_open:                                                   # do not hand a failed engine to the next restart
            _reuse.pop("eng", None)
            keep_open = False
        raise
    finally:
        if not keep_open:
            eng.close()                                                 # :457
    final = np.asarray(final, dtype=np.float64)
    convergence_info = {
        "final_elbo": float(final.mean()),
        "sd_final_elbo": float(final.std(ddof=1)),
        "elbo": np.asarray(elbos, dtype=np.float64),
    }
    # ordering/naming of :465-473
    order = ["mu", "clone_probs", "s", "alpha"]
    if P > 0 and K > 0:
        order += ["beta", "psi", "W", "chi"]
    elif K > 0:
        order += ["psi", "W", "chi"]
    elif P > 0:
        order += ["beta"]        # the reference leaves this fifth element unnamed (:471-473)
    ml_params = {k: rlist[k] for k in order if k in rlist}
    return {
        "ml_params": ml_params,
        "convergence_info": convergence_info,
        "retained_genes": retained_genes,
        "retained_mask": np.asarray(keep, dtype=bool),      # the same genes as a mask over the (selected) input genes
        "clone_probs_from_snv": clone_probs_from_snv,
        **({"post": post_out} if post is not None else {}),
    }
