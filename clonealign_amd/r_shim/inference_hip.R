# R side of the MI355X engine: what changes in the reference package (file:line of kieranrcampbell/clonealign as surveyed).
# Every .Call entry point lives in clonealign_hip_shim.c (src/) and takes R matrices as they are (column-major, numeric or
# integer -- no `storage.mode<-`, no transpose, no copy).
#
# ---------------------------------------------------------------------------------------------------------------------------
# 1. R/inference-tflow.R -- inference_tflow()
#    :204-208   pca <- prcomp(log2(Y_dat + 1), center = TRUE, scale = TRUE); pcs <- scale(pca$x[, seq_len(K)]) + rnorm(...)
#               UNCHANGED for small inputs.  For large N x G pass `psi0 = NULL, psi_noise = matrix(rnorm(N * K, 0, 0.05), N, K)`
#               to the fit below: the device does prcomp + scale by subspace iteration over the resident counts (ca_init_psi_pca).
#    :220-235   mu_guess: UNCHANGED, or pass `loc0 = NULL` (the device takes colMeans(Y / rowMeans(Y)) and
#               safe_inverse_softplus() from the resident matrix).
#    :240-457   (graph build, session loop, fetch, 20 final ELBOs, sess$close)  REPLACED by inference_hip_core() below.
#    :459-480   naming / return list: UNCHANGED (rlist and convergence_info come back under the reference's own names).
inference_hip_core <- function(Y_dat, L_dat, pcs, mu_guess, x, v_log_prob, K, mc_samples,
                               max_iter, rel_tol, learning_rate) {
  G <- ncol(Y_dat); S <- as.integer(mc_samples)
  n_draws <- 2L + 2L * as.integer(max_iter) + 20L
  eps <- rnorm(n_draws * S * G)            # R's RNG => set.seed() governs the fit (cf. get_next_seed(), :49-51,269)
  res <- .Call("C_clonealign_fit", Y_dat, L_dat, pcs, safe_inverse_softplus(mu_guess), x, v_log_prob,
               as.integer(K), S, as.integer(max_iter), as.numeric(rel_tol), as.numeric(learning_rate), eps,
               PACKAGE = "clonealign")
  hip_fit_to_rlist(res, x, K)
}

hip_fit_to_rlist <- function(res, x, K) {
  convergence_info <- list(final_elbo = mean(res$final_elbos), sd_final_elbo = sd(res$final_elbos), elbo = res$elbo)
  rlist <- res[c("mu", "clone_probs", "s", "alpha")]
  if (!is.null(x)) rlist$beta <- res$beta
  if (K > 0) { rlist$psi <- res$psi; rlist$W <- res$W; rlist$chi <- res$chi }
  list(ml_params = rlist, convergence_info = convergence_info)
}

# ---------------------------------------------------------------------------------------------------------------------------
# 2. R/clonealign.R -- run_clonealign(), :50-56: the loop `for(is in initial_shrinks) for(r in seq_len(n_repeats))
#    fits[[s]] <- do.call(clonealign, args)` REPLACED by ONE call: the count matrix is uploaded once per device, the restarts of
#    a device run on its resident engine (ca_reinit), one worker thread per device.  :58-72 (which.max(final_elbos),
#    multirun_info) UNCHANGED.  (`initial_shrink` is accepted and never read by the reference's inference_tflow(); restarts
#    differ through the RNG state only, which is what the per-restart psi0 and eps carry.)
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
allele_loglik_hip <- function(clone_allele, cov, ref, device = 0L) {
  .Call("C_clonealign_allele_loglik", clone_allele, cov, ref, as.integer(device), PACKAGE = "clonealign")   # N x C; feeds `extra`
}
