# Drop-in replacement for the TensorFlow part of inference_tflow() (R/inference-tflow.R:240-457).
# Paste between the initialisation block (:204-235) and the return-list assembly (:459-480);
# every line above :236 and below :458 of the reference function stays as it is.
#
#   pcs, s_init, mu_guess, Y_dat, L_dat, x, K, mc_samples, max_iter, rel_tol, learning_rate
#   are the reference's own local variables at that point.
inference_hip_core <- function(Y_dat, L_dat, pcs, mu_guess, x, v_log_prob, K, mc_samples,
                               max_iter, rel_tol, learning_rate) {
  G <- ncol(Y_dat); S <- as.integer(mc_samples)
  n_draws <- 2L + 2L * as.integer(max_iter) + 20L
  eps <- rnorm(n_draws * S * G)            # R's RNG => set.seed() governs the fit (cf. get_next_seed(), :49-51,269)
  storage.mode(Y_dat) <- "double"
  res <- .Call("C_clonealign_fit", Y_dat, L_dat, pcs, safe_inverse_softplus(mu_guess), x, v_log_prob,
               as.integer(K), S, as.integer(max_iter), as.numeric(rel_tol), as.numeric(learning_rate), eps,
               PACKAGE = "clonealign")
  convergence_info <- list(final_elbo = mean(res$final_elbos), sd_final_elbo = sd(res$final_elbos), elbo = res$elbo)
  rlist <- res[c("mu", "clone_probs", "s", "alpha")]
  if (!is.null(x)) rlist$beta <- res$beta
  if (K > 0) { rlist$psi <- res$psi; rlist$W <- res$W; rlist$chi <- res$chi }
  list(ml_params = rlist, convergence_info = convergence_info)
}
