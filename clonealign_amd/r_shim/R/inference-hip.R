# R/inference-hip.R -- the R side of the MI355X engine.  Drop this file into the reference package's R/ directory IN PLACE OF
# R/inference-tflow.R's `inference_tflow()` (keep that file's small helpers: safe_inverse_softplus, clone_assignment,
# recompute_clone_assignment, round2 -- R/inference-tflow.R:1-56), add src/clonealign_hip_shim.c, src/init.c, src/Makevars from
# this directory's src/, and `useDynLib(clonealign, .registration = TRUE)` to NAMESPACE.  clonealign() (R/clonealign.R:184-305)
# and run_clonealign() (:35-75) call inference_tflow() exactly as before.
#
# Every .Call entry point takes R matrices as they are (column-major, numeric or integer -- no `storage.mode<-`, no transpose,
# no copy).  What each block replaces is cited by file:line of kieranrcampbell/clonealign.

# ---------------------------------------------------------------------------------------------------------------------------
# 1. inference_tflow(): same 19 arguments, same order, same defaults as R/inference-tflow.R:71-89; same return list as :475-480.
#    R-side preparation (:117-235) is re-expressed here; the TensorFlow graph, session loop, fetches and the 20 final ELBOs
#    (:240-457) are ONE .Call.
#
#    Deliberate differences, all visible to the caller:
#      * data_init_mu defaults to TRUE (the reference's default `data_init_mu = data_init_mu` is a promise that refers to itself
#        and fails when the argument is missing; clonealign() always passes it, R/clonealign.R:280).
#      * dtype = "float64" stops with a message: the engine holds its variables in float32, TensorFlow's default and the
#        reference's (R/clonealign.R:195); the reference's own float64 graph cannot be built (:323 divides by tf$to_float(S)).
#      * the Monte-Carlo noise is drawn with rnorm() here and handed to the engine, so set.seed() governs the fit exactly as it
#        does through get_next_seed() (:49-51,269) -- tests/testthat/test_clonealign.R:61-64 holds.
#      * no per-iteration progress bar: the loop runs inside the library (it polls R_CheckUserInterrupt() every iteration, so
#        Ctrl-C works as in the R-level loop of :394-417).
#      * fix_alpha and initial_shrink are accepted and not used -- as in the reference, which never reads them either.
#      * `devices` (a 20th, trailing argument; default options(clonealign.devices), else device 0): HIP device ordinals.  More than
#        one -- options(clonealign.devices = 0:7), which reaches this function through clonealign() and run_clonealign() without
#        touching either (R/clonealign.R:262-280 passes a fixed argument list) -- makes this ONE fit cell-sharded over those
#        devices inside the .Call: cells [N r / W, N (r + 1) / W) on device r, one all-reduce of the per-gene sums per iteration
#        (SURVEY.md section 8e).  Inputs and outputs are unchanged: whole matrices in, whole matrices out.
inference_tflow <- function(Y_dat,
                            L_dat,
                            max_iter = 100,
                            rel_tol = 1e-5,
                            learning_rate = 0.1,
                            gene_filter_threshold = 0,
                            x = NULL,
                            clone_allele = NULL,
                            cov = NULL,
                            ref = NULL,
                            fix_alpha = FALSE,
                            dtype = c("float32", "float64"),
                            saturate = TRUE,
                            saturation_threshold = 6,
                            K = 1,
                            mc_samples = 1,
                            verbose = TRUE,
                            initial_shrink = 5,
                            data_init_mu = TRUE,
                            devices = getOption("clonealign.devices", 0L)) {
  say <- function(...) if (verbose) message(...)
  say("Constructing HIP engine inputs")
  dtype <- match.arg(dtype)
  if (dtype == "float64") {
    stop("dtype = 'float64' is not available: the MI355X engine holds its variables in float32 (the reference's default)")
  }

  # gene filter (:117-131)
  low <- colSums(Y_dat) <= gene_filter_threshold
  say(sprintf("Removing %d genes with low counts", sum(low)))
  Y_dat <- Y_dat[, !low]
  L_dat <- L_dat[!low, ]
  retained_genes <- if (!is.null(colnames(Y_dat))) colnames(Y_dat) else which(!low)

  N <- nrow(Y_dat)
  G <- ncol(Y_dat)
  C <- ncol(L_dat)
  K <- as.integer(K)
  S <- as.integer(mc_samples)
  stopifnot(nrow(L_dat) == G)                                              # :139
  if (saturate) L_dat <- saturate(L_dat, saturation_threshold)             # :142-144 (R/clonealign.R:394-397)
  storage.mode(L_dat) <- "double"

  # covariates (:147-153)
  P <- 0L
  if (!is.null(x)) {
    if (is.vector(x)) x <- matrix(x, ncol = 1)
    stopifnot(is.matrix(x))
    P <- ncol(x)
    stopifnot(nrow(x) == N)
    storage.mode(x) <- "double"
  }

  # allelic imbalance (:166-187): a parameter-free N x C addend to the log-likelihood, computed once on the device
  use_allele <- !is.null(clone_allele) && !is.null(ref) && !is.null(cov)
  v_log_prob <- NULL
  if (use_allele) {
    say("Using allelic imbalance info")
    V <- nrow(clone_allele)
    sanitize_allele_info(V, clone_allele, cov, ref, N, C)                  # R/allele-specific.R (unchanged)
    v_log_prob <- allele_loglik_hip(clone_allele, cov, ref)
  }

  # initial values (:204-235,262): latent space from PCA, size factors, per-gene means.  Above 4e6 counts the two O(N G) host passes --
  # prcomp() (minutes and gigabytes at 100k x 5k, for a fit that takes 60 ms) and colMeans(Y / rowMeans(Y)) -- are taken ON THE DEVICE
  # from the matrix the fit holds resident anyway: psi0 = NULL asks the engine for prcomp + scale by subspace iteration (plus the
  # reference's N(0, 0.05^2) noise, drawn here with rnorm() so that set.seed() still governs it), loc0 = NULL for the mu guess.
  # options(clonealign.device_init = TRUE / FALSE) forces either way.
  device_init <- getOption("clonealign.device_init", as.numeric(N) * G > 4e6)
  pcs <- matrix(0, N, K)
  pcs_noise <- NULL
  if (K > 0) {
    pcs_noise <- matrix(rnorm(N * K, mean = 0, sd = 0.05), nrow = N)      # :208 (same draw, same place in the RNG stream, either way)
    if (device_init) {
      pcs <- NULL
    } else {
      pca <- prcomp(log2(Y_dat + 1), center = TRUE, scale. = TRUE)
      # ONE sign convention on both sides of the device_init switch: prcomp's signs are LAPACK's and arbitrary, the device routine makes each
      # component's loading of largest magnitude positive -- the same rule here, so that one seed gives one fit whichever side initialises
      sgn <- apply(pca$rotation[, seq_len(K), drop = FALSE], 2, function(v) sign(v[which.max(abs(v))]))
      pcs <- scale(sweep(pca$x[, seq_len(K), drop = FALSE], 2, sgn, `*`))
      pcs <- pcs + pcs_noise
      attributes(pcs) <- list(dim = c(N, K))                               # plain matrix: drop scale()'s attributes
      pcs_noise <- NULL
    }
  }
  if (any(rowSums(Y_dat) == 0)) stop("Some cells have no counts mapping")  # :210-214
  loc0 <- NULL
  if (is.logical(data_init_mu)) {
    if (!(isTRUE(data_init_mu) && device_init)) {                          # (TRUE + device_init: loc0 stays NULL, the engine makes the same guess)
      mu_guess <- if (isTRUE(data_init_mu)) colMeans(Y_dat / rowMeans(Y_dat)) else rep(1, G)
      loc0 <- safe_inverse_softplus(mu_guess)                              # :262
    }
  } else if (is.numeric(data_init_mu)) {
    say("Using user-provided mu values to start")
    stopifnot(length(data_init_mu) == G)
    loc0 <- safe_inverse_softplus(data_init_mu / mean(data_init_mu))
  } else {
    stop("data_init_mu must be TRUE, FALSE or a numeric vector with one value per gene")
  }

  # the fit (:240-457): gamma initialisation, initial ELBO, the loop with its window-10 stop rule, the fetches, 20 final ELBOs
  say("Optimizing ELBO")
  n_draws <- 2L + 2L * as.integer(max_iter) + 20L                          # 1 gamma init + 1 initial ELBO + 2 per iteration + 20 final
  eps <- rnorm(n_draws * S * G)
  res <- .Call("C_clonealign_fit", Y_dat, L_dat, pcs, pcs_noise, loc0, x, v_log_prob, K, S, as.integer(max_iter), as.numeric(rel_tol),
               as.numeric(learning_rate), eps, as.integer(devices), PACKAGE = "clonealign")
  say("\nELBO converged or reached max iterations")

  # fetches in the reference's order, then the reference's naming (:424-434,465-473) -- including its quirk: with covariates and
  # K = 0 the list has five elements and the names vector four, so the fifth (beta) ends up with the name NA
  rlist <- list(res$mu, res$clone_probs, res$s, res$alpha)
  if (P > 0) rlist$beta <- res$beta
  if (K > 0) {
    rlist$psi <- res$psi
    rlist$W <- res$W
    rlist$chi <- res$chi
  }
  names(rlist) <- if (P > 0 && K > 0) {
    c("mu", "clone_probs", "s", "alpha", "beta", "psi", "W", "chi")
  } else if (K > 0) {
    c("mu", "clone_probs", "s", "alpha", "psi", "W", "chi")
  } else {
    c("mu", "clone_probs", "s", "alpha")
  }

  clone_probs_from_snv <- NULL
  if (use_allele) {                                                        # :436-440: softmax over clones of the allele term
    shifted <- v_log_prob - apply(v_log_prob, 1, max)
    clone_probs_from_snv <- exp(shifted) / rowSums(exp(shifted))
  }

  say("Computing final ELBO")                                              # (already done inside the call; :447-454)
  convergence_info <- list(final_elbo = mean(res$final_elbos), sd_final_elbo = sd(res$final_elbos), elbo = res$elbo)

  list(ml_params = rlist,
       convergence_info = convergence_info,
       retained_genes = retained_genes,
       clone_probs_from_snv = clone_probs_from_snv)
}

# ---------------------------------------------------------------------------------------------------------------------------
# 2. R/clonealign.R -- run_clonealign(), :50-56: the loop `for(is in initial_shrinks) for(r in seq_len(n_repeats))
#    fits[[s]] <- do.call(clonealign, args)` can become ONE call: the count matrix is uploaded once per device, the restarts of
#    a device run on its resident engine (ca_reinit), one worker thread per device.  :58-72 (which.max(final_elbos),
#    multirun_info) unchanged.  (`initial_shrink` is accepted and never read by the reference's inference_tflow(); restarts
#    differ through the RNG state only, which is what the per-restart psi0 and eps carry.)
hip_fit_to_rlist <- function(res, x, K) {
  convergence_info <- list(final_elbo = mean(res$final_elbos), sd_final_elbo = sd(res$final_elbos), elbo = res$elbo)
  rlist <- res[c("mu", "clone_probs", "s", "alpha")]
  if (!is.null(x)) rlist$beta <- res$beta
  if (K > 0) { rlist$psi <- res$psi; rlist$W <- res$W; rlist$chi <- res$chi }
  list(ml_params = rlist, convergence_info = convergence_info)
}

run_restarts_hip <- function(Y_dat, L_dat, pcs, mu_guess, x, v_log_prob, K, mc_samples, max_iter, rel_tol, learning_rate,
                             n_restarts, devices = 0L, device_pca = FALSE, want_correlation_sums = TRUE) {
  N <- nrow(Y_dat); G <- ncol(Y_dat); S <- as.integer(mc_samples)
  n_draws <- 2L + 2L * as.integer(max_iter) + 20L
  noise <- lapply(seq_len(n_restarts), function(r) matrix(rnorm(N * K, 0, 0.05), N, K))      # :208, once per restart
  psi0 <- if (device_pca) NULL else lapply(noise, function(e) pcs + e)
  eps <- lapply(seq_len(n_restarts), function(r) rnorm(n_draws * S * G))
  fits <- .Call("C_clonealign_multifit", Y_dat, L_dat, psi0, if (device_pca) noise else NULL,
                if (is.null(mu_guess)) NULL else safe_inverse_softplus(mu_guess), x, v_log_prob, as.integer(K), S,
                as.integer(max_iter), as.numeric(rel_tol), as.numeric(learning_rate), eps, as.integer(devices),
                as.logical(want_correlation_sums), 0.95, PACKAGE = "clonealign")
  lapply(fits, function(res) c(hip_fit_to_rlist(res, x, K), list(gene_sums = res$T, gene_sumsq = res$Syy)))
}

# ---------------------------------------------------------------------------------------------------------------------------
# 3. R/clonealign.R -- clonealign(), :283-303 (after inference_tflow() returns):
#    :292-294   correlations <- compute_correlations(Y, L, clones)   -- ships nothing back when the fit carries the sums:
correlations_from_sums <- function(T, Syy, L, clone_sizes) {
  # Pearson r per gene between the copy number of a cell's assigned clone and its count (R/clonealign.R:318-334), from
  # T[g, c] = sum of y over the cells assigned to clone c, Syy[g] = sum of y^2 over assigned cells, clone_sizes[c]
  n <- sum(clone_sizes)
  sx <- as.vector(L %*% clone_sizes); sxx <- as.vector(L^2 %*% clone_sizes)
  sy <- rowSums(T); sxy <- rowSums(L * T)
  (n * sxy - sx * sy) / sqrt((n * sxx - sx^2) * (n * Syy - sy^2))
}

# ---------------------------------------------------------------------------------------------------------------------------
# 4. R/preprocess.R -- preprocess_for_clonealign(), :93-147: the two O(N G) statistics (colSums, rowSums over the kept genes)
#    and the O(G) decisions come back as masks; the subsetting lines :141-147 stay as they are and use them.
preprocess_masks_hip <- function(Y, L, min_counts_per_gene = 20, min_counts_per_cell = 100, remove_outlying_genes = TRUE,
                                 nmads = 10, max_copy_number = 6, remove_genes_same_copy_number = TRUE, device = 0L) {
  .Call("C_clonealign_preprocess", Y, L, as.numeric(min_counts_per_gene), as.numeric(min_counts_per_cell),
        as.logical(remove_outlying_genes), as.numeric(nmads), as.numeric(max_copy_number),
        as.logical(remove_genes_same_copy_number), as.integer(device), PACKAGE = "clonealign")
}

# ---------------------------------------------------------------------------------------------------------------------------
# 5. R/inference-tflow.R :166-187 with R/allele-specific.R:17-58: the parameter-free allele addend, computed once per fit.
#    cov and ref are the cell x variant matrices clonealign() hands over (before the reference's "legacy" transposes, :173-174).
allele_loglik_hip <- function(clone_allele, cov, ref, device = 0L) {
  storage.mode(clone_allele) <- "double"; storage.mode(cov) <- "double"; storage.mode(ref) <- "double"
  .Call("C_clonealign_allele_loglik", clone_allele, cov, ref, as.integer(device), PACKAGE = "clonealign")   # N x C; feeds `extra`
}
