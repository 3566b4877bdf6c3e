// Host side of the clonealign VI engine + the C ABI of include/clonealign_hip.h.
//
// One ca_engine = one fit = what the reference holds in one TF graph + session
// (R/inference-tflow.R:99,351,457).  Everything runs on one HIP stream per handle; the only
// host synchronisation inside the loop is the ELBO read-back the reference's convergence
// test needs (:403-415).  Built for gfx950 only:  hipcc --offload-arch=gfx950 (see build.py).
#include "clonealign_hip.h"

#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "ca_kernels.hip.h"
#include "ca_poly.h"
#include "philox_host.h"

#include "ca_eng_state.inc"   // engine state (struct ca_engine), error / launch macros, profiling wrappers, variant switches, template dispatch of the VALU sweeps
#include "ca_eng_loop.inc"   // the loop: host matrix helpers, count-matrix products per parameter state, transports' all-reduce, backward / update halves, plain and fused passes (sweeps or series form), eps staging
#include "ca_eng_ingest.inc"   // ingestion: storage scan and conversion, selection gather, host -> device pipeline (float64 narrowed on the host), fit constants
#include "ca_eng_create.inc"   // create_impl: buffers, decomposition picks (every threshold measured: DESIGN.md section 5), setup sums; find_param
}  // namespace

// =============================================================================== C ABI
template <typename ST>
static int preprocess_t(const ST* src, int64_t N, int G, int C, int layout, const double* L, const ca_preprocess_params& pp,
                        uint8_t* keep_gene, uint8_t* keep_cell, double* gene_sums, double* cell_sums, std::string& msg) {
  const int64_t sn = layout == CA_COL_MAJOR ? 1 : G, sg = layout == CA_COL_MAJOR ? N : 1;
  double *part = nullptr, *dcol = nullptr, *drow = nullptr; unsigned char* dkeep = nullptr;
  auto cleanup = [&]() { hipFree(part); hipFree(dcol); hipFree(drow); hipFree(dkeep); };
#define PCK2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { msg = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return CA_ERR_HIP; } } while (0)
  const int rpb = 1024, nrb = cdiv(N, rpb);
  PCK2(hipMalloc((void**)&part, (size_t)nrb * G * sizeof(double)));
  PCK2(hipMalloc((void**)&dcol, (size_t)G * sizeof(double)));
  PCK2(hipMalloc((void**)&drow, (size_t)N * sizeof(double)));
  PCK2(hipMalloc((void**)&dkeep, (size_t)G));
  hipLaunchKernelGGL((k_pre_colsum<ST>), dim3(cdiv(G, CA_TB), nrb), dim3(CA_TB), 0, 0, src, N, G, sn, sg, rpb, part);
  hipLaunchKernelGGL(k_pre_colsum_final, dim3(cdiv(G, CA_TB)), dim3(CA_TB), 0, 0, part, nrb, G, dcol);
  PCK2(hipGetLastError());
  std::vector<double> col((size_t)G);
  PCK2(hipMemcpy(col.data(), dcol, (size_t)G * sizeof(double), hipMemcpyDeviceToHost));
  // ---- O(G) decisions, in the reference's order
  std::vector<uint8_t> kg((size_t)G, 1);
  for (int g = 0; g < G; ++g) {                                                   // :114-116
    double mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = std::max(mx, L[hidx(layout, g, c, G, C)]);
    if (mx > pp.max_copy_number) kg[g] = 0;
  }
  for (int g = 0; g < G; ++g) if (kg[g] && !(col[g] > pp.min_counts_per_gene)) kg[g] = 0;   // :118-120
  if (pp.remove_outlying_genes) {                                                 // :59-63, 123-128
    std::vector<double> gm;
    for (int g = 0; g < G; ++g) if (kg[g]) gm.push_back(col[g] / (double)N);
    if (!gm.empty()) {
      auto median = [](std::vector<double> v) {
        std::sort(v.begin(), v.end());
        const size_t n = v.size();
        return n % 2 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
      };
      const double med = median(gm);
      std::vector<double> dev(gm.size());
      for (size_t i = 0; i < gm.size(); ++i) dev[i] = std::fabs(gm[i] - med);
      const double md = 1.4826 * median(dev);
      double mean = 0.0;
      for (double v : gm) mean += v;
      mean /= (double)gm.size();
      const double thr = mean + pp.nmads * md;
      for (int g = 0; g < G; ++g) if (kg[g] && col[g] / (double)N > thr) kg[g] = 0;
    }
  }
  if (pp.remove_genes_same_copy_number && C >= 2) {                               // :131-135 (rowVars == 0)
    for (int g = 0; g < G; ++g) {
      if (!kg[g]) continue;
      double m = 0.0;
      for (int c = 0; c < C; ++c) m += L[hidx(layout, g, c, G, C)];
      m /= C;
      double v = 0.0;
      for (int c = 0; c < C; ++c) { const double d = L[hidx(layout, g, c, G, C)] - m; v += d * d; }
      if (v / (C - 1) == 0.0) kg[g] = 0;
    }
  }
  PCK2(hipMemcpy(dkeep, kg.data(), (size_t)G, hipMemcpyHostToDevice));
  if (layout == CA_COL_MAJOR)
    hipLaunchKernelGGL((k_pre_rowsum_cm<ST>), dim3(cdiv(N, CA_TB)), dim3(CA_TB), 0, 0, src, dkeep, N, G, sn, sg, drow);
  else
    hipLaunchKernelGGL((k_pre_rowsum<ST>), dim3(cdiv(N, CA_TB / 64)), dim3(CA_TB), 0, 0, src, dkeep, N, G, sn, sg, drow);
  PCK2(hipGetLastError());
  std::vector<double> row((size_t)N);
  PCK2(hipMemcpy(row.data(), drow, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t n = 0; n < N; ++n) keep_cell[n] = row[n] > pp.min_counts_per_cell ? 1 : 0;   // :138-139
  memcpy(keep_gene, kg.data(), (size_t)G);
  if (gene_sums) memcpy(gene_sums, col.data(), (size_t)G * sizeof(double));
  if (cell_sums) memcpy(cell_sums, row.data(), (size_t)N * sizeof(double));
#undef PCK2
  cleanup();
  return CA_OK;
}

// ca_run's gated update: what the host's bookkeeping of a step changes (train_from_lookahead + the merged branch of train_update) -- taken before
// the launch is queued, put back when the host's answer is "stop" (the launch then stores nothing)
struct gate_snapshot {
  bool look_valid, bwd_ready, pre_valid, em_stale, y_defer, ys_quant_ready, ycache_valid, yfin_pending, fold_now;
  int64_t pre_A, pre_B, hint_A, hint_B, gaux_slot;
  int vmm_at_idx, gaux_idx, ys_steps, ys_namax[3];
  float b1p, b2p; uint64_t adam_steps;
  int poly_xglob_steps;
  float *vchi, *vchi_alt, *alpha_u, *alpha_u_alt;
  void take(const ca_engine* h) {
    look_valid = h->look_valid; bwd_ready = h->bwd_ready; pre_valid = h->pre_valid; em_stale = h->em_stale; y_defer = h->y_defer;
    ys_quant_ready = h->ys_quant_ready; ycache_valid = h->ycache_valid; yfin_pending = h->yfin_pending; fold_now = h->fold_now;
    pre_A = h->pre_A; pre_B = h->pre_B; hint_A = h->hint_A; hint_B = h->hint_B; gaux_slot = h->gaux_slot;
    vmm_at_idx = h->vmm_at_idx; gaux_idx = h->gaux_idx; ys_steps = h->ys_steps;
    for (int i = 0; i < 3; ++i) ys_namax[i] = h->ys_namax[i];
    b1p = h->b1p; b2p = h->b2p; adam_steps = h->adam_steps;
    poly_xglob_steps = h->poly_xglob_steps;
    vchi = h->vchi; vchi_alt = h->vchi_alt; alpha_u = h->alpha_u; alpha_u_alt = h->alpha_u_alt;
  }
  void restore(ca_engine* h) const {
    h->look_valid = look_valid; h->bwd_ready = bwd_ready; h->pre_valid = pre_valid; h->em_stale = em_stale; h->y_defer = y_defer;
    h->ys_quant_ready = ys_quant_ready; h->ycache_valid = ycache_valid; h->yfin_pending = yfin_pending; h->fold_now = fold_now;
    h->pre_A = pre_A; h->pre_B = pre_B; h->hint_A = -1; h->hint_B = -1; h->gaux_slot = gaux_slot;
    h->vmm_at_idx = vmm_at_idx; h->gaux_idx = gaux_idx; h->ys_steps = ys_steps;
    for (int i = 0; i < 3; ++i) h->ys_namax[i] = ys_namax[i];
    h->b1p = b1p; h->b2p = b2p; h->adam_steps = adam_steps;
    h->poly_xmax_ready = false;   // (the cancelled launch added nothing to the series form's max |psi| word)
    // Sharded series: the slots describe the state this launch did NOT change -- the count goes back, it is NOT invalidated: a launch that gave up is a rank's own
    // affair (its host was late), and a refresh of the slots is a collective -- one rank refreshing alone paired its four doubles with its peers' next all-reduce
    // (their cell sums landed in its slots: max |psi| of 1e8, "cannot cover the exponent range", the peers timing out; four ranks on one device, found late round 6)
    h->poly_xglob_steps = poly_xglob_steps;
    h->vchi = vchi; h->vchi_alt = vchi_alt; h->alpha_u = alpha_u; h->alpha_u_alt = alpha_u_alt;
  }
};
// ... and what queuing the NEXT forward sweep behind that launch changes (fused_pass in its one-launch form): taken after the update is queued, put back
// first when the answer is "stop" (every block of the sweep then returns at its first instruction)
struct fwd_snapshot {
  double *gene_part, *gene_part_alt, *gene_partB, *gene_partB_alt;
  bool pre_valid, em_stale, yfin_pending, ys_quant_ready, ycache_valid, look_valid, bwd_ready;
  int ys_slot, ys_steps;
  int64_t look_slot;
  unsigned long long host_seq, host_seq_next;
  ca_small_args mon_tail;
  void take(const ca_engine* h) {
    gene_part = h->gene_part; gene_part_alt = h->gene_part_alt; gene_partB = h->gene_partB; gene_partB_alt = h->gene_partB_alt;
    pre_valid = h->pre_valid; em_stale = h->em_stale; yfin_pending = h->yfin_pending; ys_quant_ready = h->ys_quant_ready; ycache_valid = h->ycache_valid;
    look_valid = h->look_valid; bwd_ready = h->bwd_ready; ys_slot = h->ys_slot; ys_steps = h->ys_steps; look_slot = h->look_slot;
    host_seq = h->host_seq; host_seq_next = h->host_seq_next; mon_tail = h->mon_tail;
  }
  void restore(ca_engine* h) const {
    h->gene_part = gene_part; h->gene_part_alt = gene_part_alt; h->gene_partB = gene_partB; h->gene_partB_alt = gene_partB_alt;
    h->pre_valid = pre_valid; h->em_stale = em_stale; h->yfin_pending = yfin_pending; h->ys_quant_ready = ys_quant_ready; h->ycache_valid = ycache_valid;
    h->look_valid = look_valid; h->bwd_ready = bwd_ready; h->ys_slot = ys_slot; h->ys_steps = ys_steps; h->look_slot = look_slot;
    h->host_seq = host_seq; h->host_seq_next = host_seq_next; h->mon_tail = mon_tail;
  }
};
// will fused_pass(slotA, slotB) be exactly ONE launch, the merged forward sweep with the int8 stream riding (k_fwd_cell_mix_ys)?  (prologue and quantised
// images made by the merged update, exponent bound taken by the sweep itself, finisher left for the backward sweep, ELBO assembly left pending)
inline bool fwd_is_one_launch(const ca_engine* h, int64_t slotA, int64_t slotB) {
  return h->fused_ok && !h->s2 && h->fwd_cell && h->ride_ys && !h->ycache_valid && !h->y_defer && !h->y_pending && h->pre_valid && h->pre_A == slotA &&
         h->pre_B == slotB && h->ys_quant_ready && h->yfin_split && h->em_stale && h->D > 0 && h->tail_fuse && !is_sharded(h) && h->gate_local;
}
// the host's answer to the launch queued last: go (1) or stop (0)
inline void gate_answer(ca_engine* h, int go) {
  volatile unsigned long long* w = reinterpret_cast<volatile unsigned long long*>(h->host_pinned + 48);
  std::atomic_thread_fence(std::memory_order_release);
  *w = (h->gate_seq << 1) | (unsigned long long)(go ? 1 : 0);
  std::atomic_thread_fence(std::memory_order_seq_cst);
}
// Answer the gated launch and learn what it DID: 1 = it ran (the answer was "go" and reached the relay block in time), 0 = it stored nothing
// (answer "stop", or the relay had given up before the answer came: a slow poll hook, a descheduled or stopped host).  The relay's clock
// starts no earlier than gate_t0, so an answer written within half its patience has been seen; otherwise its verdict is read from the
// pinned word it leaves (it always leaves one: it either sees the answer or runs out of patience).  Negative: an error (stream failure).
int gate_resolve(ca_engine* h, int go) {
  gate_answer(h, go);
  if (!go) return 0;
  // The shortcut compares two clocks (the host's, and the relay's s_memrealtime that starts later) and needs the pinned write to land before the
  // relay's last poll: it is taken only with room to spare -- a patience of 200 us or more, answered within half of it (with 10-50 us the
  // margin was 5-25 us, and a relay that had already given up while the host assumed "ran" would be a silently wrong fit).  Shorter patience: always
  // the relay's own word.
  const double patience_us = (double)h->gate_ticks / h->ticks_per_us;
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->gate_t0).count();
  if (patience_us >= 200.0 && us < 0.5 * patience_us) return 1;
  volatile unsigned long long* ack = reinterpret_cast<volatile unsigned long long*>(h->host_pinned + 56);
  unsigned spins = 0;
  auto t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
  for (;;) {
    const unsigned long long a = *ack;
    if ((a >> 2) == h->gate_seq) return (a & 3ull) == 1ull ? 1 : 0;
    if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() >= t_next) {   // (the launch may still be waiting for its turn: a big backward sweep, a co-tenant)
      t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        const unsigned long long a2 = *ack;
        if ((a2 >> 2) == h->gate_seq) return (a2 & 3ull) == 1ull ? 1 : 0;
        h->err = "ca_run: the gated update completed without leaving its verdict";
        return -CA_ERR_STATE;
      }
      if (q != hipErrorNotReady) { h->err = std::string("hipStreamQuery: ") + hipGetErrorString(q); return -CA_ERR_HIP; }
    }
  }
}
// A poll hook (or anything else on this thread) enters a read-only API call while a gated launch waits for the host: close the window first -- answer
// "store nothing", wait for the stream (the launch ends at once), put the host's bookkeeping of the queued step back.  ca_run then sees gate_aborted
// and queues the update again after its decision (the lock-step order); the variables a hook reads are those after the last completed iteration.
int gate_close(ca_engine* h) {
  if (!h->gate_open) return CA_OK;
  h->gate_open = false; h->gate_aborted = true;
  gate_answer(h, 0);
  HIPCK(h, hipStreamSynchronize(h->stream));
  if (h->gate_fsnap) h->gate_fsnap->restore(h);
  if (h->gate_snap) h->gate_snap->restore(h);
  return CA_OK;
}
#define CA_NOT_IN_RUN(h)                                                                                                                  \
  do {                                                                                                                                    \
    if ((h) && (h)->in_run) {                                                                                                             \
      (h)->err = "this call changes the engine's state and cannot be made from a ca_run_ex poll hook (read-only calls can: ca_get_param, " \
                 "ca_get_gradient, ca_get_info, ca_synchronize, ca_get_kernel_times)";                                                   \
      return CA_ERR_STATE;                                                                                                                \
    }                                                                                                                                     \
  } while (0)


extern "C" {

int ca_abi_version(void) { return CA_ABI_VERSION; }
#ifdef CA_LAB   // timing-lab builds only: readers of the block stamps (tools/lab/)
#include "../../tools/lab/ca_lab_host.inc"
#endif
// (ca_build_id(): ca_build_id.cpp -- its own translation unit, so that the hash of ALL sources does not recompile this one)

int ca_device_count(int32_t* n) {
  int c = 0;
  const hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
  if (n) *n = c;
  return e == hipSuccess ? CA_OK : CA_ERR_HIP;
}

int ca_default_options(ca_options* o) {
  if (!o) return CA_ERR_INVALID;
  memset(o, 0, sizeof(*o));
  o->learning_rate = 0.1;
  o->beta1 = 0.9; o->beta2 = 0.999; o->adam_eps = 1e-8;
  o->seed = 0x5eed5eedull;
  o->device = 0; o->y_storage = CA_YSTORE_AUTO; o->rank = 0; o->world = 1; o->profile = 0;
  return CA_OK;
}

const char* ca_last_error(ca_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int ca_create(const ca_problem* p, const ca_options* o, ca_handle* out) {
  g_last_error.clear();
  if (!p || !out) { g_last_error = "null argument"; return CA_ERR_INVALID; }
  *out = nullptr;
  ca_options opt;
  if (o) opt = *o; else ca_default_options(&opt);
  auto bad = [&](const char* m) { g_last_error = m; return CA_ERR_INVALID; };
  if (p->N <= 0 || p->G <= 0 || p->C <= 0) return bad("N, G and C must be positive");
  if (p->C > 256) return bad("C > 256 clones not supported");
  if (p->K < 0 || p->P < 0 || p->S < 1) return bad("K >= 0, P >= 0, S >= 1 required");
  const int D = p->K > 0 ? p->K + p->P : 0;
  if (D > 8) return bad("K + P > 8 not supported");
  if (!p->Y || !p->L) return bad("Y and L are required");
  if (p->K > 0 && !p->psi0) return bad("psi0 is required when K > 0");
  if (p->P > 0 && !p->X) return bad("X is required when P > 0");
  if (opt.world < 1 || opt.rank < 0 || opt.rank >= opt.world) return bad("bad rank/world");
#ifndef CA_LAB
  if (opt.variant_on & CA_LAB_VARX)
    return bad("ca_options.variant_on asks for a variant that is built into the lab library only (CA_VARX_Y_MFMA2, CA_VARX_RIDE_SEQ, CA_VARX_BAL_TILES: measured slower than "
               "what ships; `make -C clonealign_amd/csrc lab`)");
#endif
  if ((p->cell_index || p->gene_index) && (p->N_src < p->N || p->G_src < p->G)) return bad("N_src / G_src must be at least N / G when a selection is given");
  ca_engine* h = new ca_engine();
  h->N = p->N; h->G = p->G; h->C = p->C; h->K = p->K; h->P = p->P; h->S = p->S; h->D = D;
  h->layout = p->layout; h->opt = opt; h->device = opt.device;
  int rc;
  try { rc = create_impl(h, p); }
  catch (const std::exception& ex) { h->err = std::string("ca_create: ") + ex.what(); rc = CA_ERR_NOMEM; }   // (bad_alloc / system_error never cross the C ABI)
  if (rc != CA_OK) {
    g_last_error = h->err;
    ca_destroy(h);
    return rc;
  }
  *out = h;
  return CA_OK;
}

int ca_destroy(ca_handle h) {
  if (!h) return CA_OK;
  CA_NOT_IN_RUN(h);
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
  if (h->ev_params) hipEventDestroy(h->ev_params);
  if (h->ev_ydone) hipEventDestroy(h->ev_ydone);
  if (h->stream3) { hipStreamSynchronize(h->stream3); hipStreamDestroy(h->stream3); }
  if (h->ev_poly0) hipEventDestroy(h->ev_poly0);
  if (h->ev_poly1) hipEventDestroy(h->ev_poly1);
  if (h->ev_ywdone) hipEventDestroy(h->ev_ywdone);
  if (h->ev_stage) hipEventDestroy(h->ev_stage);
  if (h->comm) g_rccl.CommDestroy(h->comm);
  if (h->p2p) {
    for (void* q : h->p2p->opened) if (q) hipIpcCloseMemHandle(q);
    if (h->p2p->err_host) hipHostFree(h->p2p->err_host);
    if (h->p2p->err_local) hipFree(h->p2p->err_local);
    if (h->p2p->slab) hipFree(h->p2p->slab);
    if (h->p2p->peers_dev) hipFree(h->p2p->peers_dev);
    delete h->p2p;
  }
  for (auto& e : h->ev_pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (void* q : h->allocs) hipFree(q);
  if (h->eps_dev) hipFree(h->eps_dev);
  if (h->elbo_dev) hipFree(h->elbo_dev);
  if (h->host_pinned) hipHostFree(h->host_pinned);
  if (h->eps_stage) hipHostFree(h->eps_stage);
  if (h->host_ar_buf) hipHostFree(h->host_ar_buf);
  if (h->poly_ring) hipHostFree(h->poly_ring);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return CA_OK;
}

int ca_get_info(ca_handle h, ca_info* i) {
  if (!h || !i) return CA_ERR_INVALID;
  memset(i, 0, sizeof(*i));
  i->N = h->N; i->G = h->G; i->C = h->C; i->K = h->K; i->P = h->P; i->S = h->S;
  i->y_storage = h->ystore; i->y_bytes_per_elem = h->ybytes; i->y_device_bytes = h->y_dev_bytes; i->device_bytes = h->dev_bytes;
  i->gsplit = h->gsplit; i->csplit = h->csplit; i->n_cu = h->n_cu; i->fused_sweep = h->fused_ok ? 1 : 0;
  i->fwd_mfma = ((h->fused_ok && h->fwd_mfma) || h->pfwd) ? 1 : 0; i->bwd_mfma = h->bwd_mfma ? 1 : 0; i->fsplit = h->fsplit; i->fwd_cell = (h->fused_ok && h->fwd_cell) ? 1 : 0;
  i->y_mfma = h->y_ys ? 2 : h->y_mfma ? 1 : 0;
  i->y_ride = (h->ride_ok || h->ride_ys) ? 1 : 0;
  i->transport = (h->p2p && h->p2p->connected) ? CA_TRANSPORT_P2P : h->comm ? CA_TRANSPORT_RCCL : h->host_ar ? CA_TRANSPORT_HOST : CA_TRANSPORT_NONE;
  i->red_n = h->red_n;
  i->fwd_block_cells = (h->fused_ok && h->fwd_cell) ? 16 * h->fc_tl : 0; i->fwd_blocks_big = h->fc_nbig;
  i->fwd_balanced = h->fwd_bal ? h->bal_q : 0;
  i->fwd_series = h->poly ? 1 : 0; i->series_passes = h->n_series; i->series_fallbacks = h->n_series_fallback;
  i->fold_gsum = (h->fold_gsum && !is_sharded(h)) ? 1 : 0; i->yfin_split = (h->yfin_split && !is_sharded(h)) ? 1 : 0;
  i->update_merge = (h->upd_merge && h->fused_ok && h->fwd_cell && h->K > 0) ? 1 : 0;
  return CA_OK;
}

int ca_stream_busy(ca_handle h, int32_t* busy) {
  if (!h || !busy) return CA_ERR_INVALID;
  const hipError_t q = hipStreamQuery(h->stream);
  if (q != hipSuccess && q != hipErrorNotReady) HIPCK(h, q);
  *busy = q == hipErrorNotReady ? 1 : 0;
  if (q == hipErrorNotReady) (void)hipGetLastError();
  return CA_OK;
}

int ca_synchronize(ca_handle h) {
  if (!h) return CA_ERR_INVALID;
  HIPCK(h, hipSetDevice(h->device));
  CACK(gate_close(h));   // (called from a ca_run_ex poll hook while a gated launch waits: close that window first)
  SYNC(h);
  return CA_OK;
}

#include "ca_eng_comm.inc"   // C ABI, transports: RCCL communicator, one-shot peer-to-peer set-up (two-phase), benchmark, known-answer test, host callback
static int stage_one(ca_handle h, const float* eps) { return stage_eps(h, eps, 1, 1); }

int ca_gamma_init(ca_handle h, const float* eps) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_one(h, eps));
  CACK(run_pass(h, 0, CA_MODE_GINIT, 0, nullptr));
  SYNC(h);
  return CA_OK;
}

int ca_elbo(ca_handle h, const float* eps, double* elbo) {
  if (!h || !elbo) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_one(h, eps));
  CACK(run_pass(h, 0, CA_MODE_ELBO, 0, h->elbo_dev));
  return read_doubles(h, h->elbo_dev, elbo, 1);
}

int ca_elbo_terms(ca_handle h, const float* eps, double terms[3]) {
  if (!h || !terms) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_one(h, eps));
  CACK(run_pass(h, 0, CA_MODE_ELBO, 0, h->elbo_dev));
  return read_doubles(h, h->terms_dev, terms, 3);
}

int ca_step(ca_handle h, const float* eps) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_one(h, eps));
  CACK(run_pass(h, 0, CA_MODE_TRAIN, 1, h->elbo_dev));
  SYNC(h);
  return CA_OK;
}

int ca_gradients(ca_handle h, const float* eps, double* elbo) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_one(h, eps));
  CACK(run_pass(h, 0, CA_MODE_TRAIN, 0, h->elbo_dev));
  double e;
  CACK(read_doubles(h, h->elbo_dev, &e, 1));
  if (elbo) *elbo = e;
  return CA_OK;
}

int ca_run(ca_handle h, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws, double* trace, int32_t* n_elbo) {
  return ca_run_ex(h, max_iter, rel_tol, eps_stream, n_draws, trace, n_elbo, nullptr, nullptr);
}

static int run_loop(ca_engine* h, int32_t max_iter, double rel_tol, double* trace, int32_t* n_elbo, ca_poll_fn poll, void* user) {
  CACK(run_pass(h, 0, CA_MODE_GINIT, 0, nullptr));                      // :368-369
  h->host_seq_next = ++h->host_seq;
  CACK(monitor_pass(h, 1, max_iter >= 1 ? 2 : -1, h->elbo_dev));        // :372 (+ forward half of the first train pass)
  double val;
  CACK(train_bwd_speculative(h));
  CACK(flush_mon_tail(h));
  CACK(wait_host_elbo(h, h->host_seq, h->elbo_dev, &val));
  trace[0] = val; *n_elbo = 1;
  if (std::isnan(val)) { h->err = "Initial elbo is NA"; return CA_ERR_NAN; }   // :374-376
  // (a stop request leaves the speculative backward sweep of the next train pass queued: it changes no variable)
  if (poll && poll(user, 0, val) != 0) { h->err = "interrupted by the poll callback"; return CA_INTERRUPTED; }
  double diffs[10];
  for (double& d : diffs) d = 1e3;                                      // :379
  // Round 4: the host's look at every ELBO no longer costs a launch and a round trip.  Until now an iteration of this loop was ... backward sweep,
  // k_final_small (the ELBO, 7.4 us), THEN the host read it, decided, and only then queued the update (11 us of idle GPU: tools/gaps.py on the
  // bench trace) -- 19 us per iteration that ca_iterate does not pay.  Now the update half of train pass i + 1 is queued right behind the
  // backward sweep: its monitor block assembles ELBO i and mirrors it to the host, every other block waits on the device for the verdict of
  // the launch's relay block, which polls the word the host writes once it has decided (ca_merge_args::gate).  "Stop" (tolerance, poll hook,
  // NaN) makes the queued launch a no-op and the host takes back the bookkeeping of the step it had queued: the state is what the lock-step
  // loop leaves, bit for bit (tests).
  // Round 5: so does NO ANSWER within the relay's patience (~1 ms: a poll hook that shows a progress bar, sits in a debugger or sleeps; a
  // descheduled or stopped process) -- the launch gives up, stores nothing, the device goes idle, and once the host has decided it queues
  // the update again the lock-step way.  A slow hook costs that iteration the gate, never the fit (r4: CA_ERR_STATE after a 10 s GPU spin).
  bool queued = false;   // the update half of train pass i was queued (gated, and it ran) by the previous turn of the loop
  bool fwd_queued = false;   // ... and the forward sweep of monitor pass i / train pass i + 1 behind it
  for (int i = 1; i <= max_iter; ++i) {
    if (!queued) {
      h->hint_A = 2 * (int64_t)i + 1; h->hint_B = i < max_iter ? 2 * (int64_t)i + 2 : -1;
      CACK(train_pass(h, 2 * (int64_t)i));                              // :401
    }
    queued = false;
    if (!fwd_queued) {
      h->host_seq_next = ++h->host_seq;
      CACK(monitor_pass(h, 2 * (int64_t)i + 1, i < max_iter ? 2 * (int64_t)i + 2 : -1, h->elbo_dev + i));   // :403
    }
    fwd_queued = false;
    double nv;
    CACK(train_bwd_speculative(h));     // backward half of train pass i+1 runs while the host looks at ELBO i
    // the gated update of train pass i + 1 (needs: a next pass, its look-ahead forward and backward halves in place, the one-launch update)
    gate_snapshot snap;
    bool gated = false;
    if (h->run_gate && i < max_iter && h->bwd_ready && h->look_valid && h->look_slot == 2 * (int64_t)(i + 1) && (!h->mon_tail.enabled || h->mon_tail.host_out)) {
      h->hint_A = 2 * (int64_t)(i + 1) + 1; h->hint_B = i + 1 < max_iter ? 2 * (int64_t)(i + 1) + 2 : -1;
      if (update_merges(h, 1, nullptr)) {
        snap.take(h);
        h->gate_req = true;
        const int rc = train_pass(h, 2 * (int64_t)(i + 1));
        gated = h->gate_armed;
        h->gate_armed = false;
        if (rc != CA_OK) { if (gated) gate_answer(h, 0); return rc; }
      } else {
        h->hint_A = h->hint_B = -1;
      }
    }
    if (!gated) CACK(flush_mon_tail(h));
    const unsigned long long seq_i = h->host_seq;   // the flag ELBO i comes with
    // ... and the forward sweep that follows the gated update (monitor pass i + 1 with the forward half of train pass i + 2), queued behind it before the
    // host has seen ELBO i: the launch latency between the host's "go" and that sweep (4-5 us per iteration on small problems) is gone
    fwd_snapshot fsnap;
    bool fwd_ahead = false;
    if (gated && h->run_fwd && i + 1 < max_iter && fwd_is_one_launch(h, 2 * (int64_t)(i + 1) + 1, 2 * (int64_t)(i + 1) + 2)) {
      fsnap.take(h);
      h->host_seq_next = ++h->host_seq;
      h->fwd_gate = true;
      const int rcf = monitor_pass(h, 2 * (int64_t)(i + 1) + 1, 2 * (int64_t)(i + 1) + 2, h->elbo_dev + i + 1);
      h->fwd_gate = false;
      if (rcf != CA_OK) { gate_answer(h, 0); fsnap.restore(h); snap.restore(h); return rcf; }
      fwd_ahead = true;
    }
    // from here to the answer a gated launch waits on the device (at most the relay's patience): the window a re-entrant API call must close first
    h->gate_aborted = false;
    if (gated) { h->gate_snap = &snap; h->gate_fsnap = fwd_ahead ? &fsnap : nullptr; h->gate_open = true; }
    auto undo = [&]() { if (h->gate_open) { h->gate_open = false; if (fwd_ahead) fsnap.restore(h); snap.restore(h); } };   // (gate_close() has done it otherwise)
    int rc = wait_host_elbo(h, seq_i, h->elbo_dev + i, &nv);
    if (rc == CA_OK && *reinterpret_cast<volatile unsigned long long*>(h->host_pinned + 40) != 0ull) {
      h->err = "ca_run: blocks of a gated update never got their relay block's verdict (launch #" + std::to_string(*reinterpret_cast<volatile unsigned long long*>(h->host_pinned + 40)) + "); the engine's state is undefined";
      rc = CA_ERR_STATE;
    }
    if (rc != CA_OK) { if (gated) { gate_answer(h, 0); undo(); } return rc; }
    const double diff = (nv - val) / std::fabs(val);
    for (int j = 0; j < 9; ++j) diffs[j] = diffs[j + 1];
    diffs[9] = diff;
    trace[i] = nv; *n_elbo = i + 1;
    val = nv;
    double mean = 0.0;
    for (double d : diffs) mean += std::fabs(d);
    mean /= 10.0;
    int stop = 0;   // 0 go on, else the return code (CA_OK = converged)
    if (std::isnan(mean)) { h->err = "missing value where TRUE/FALSE needed"; stop = CA_ERR_NAN; }  // R's if (NA) at :414
    else if (poll && poll(user, i, nv) != 0) { h->err = "interrupted by the poll callback"; stop = CA_INTERRUPTED; }
    else if (mean < rel_tol) stop = -1;                                 // :414-415
    if (gated) {
      int ran = 0;
      if (h->gate_open) {
        ran = gate_resolve(h, stop == 0 ? 1 : 0);
        if (ran < 0) { undo(); return -ran; }
      }   // (else: the hook called back into the API and gate_close() answered "store nothing" and restored the bookkeeping)
      if (ran == 1) { h->gate_open = false; queued = true; fwd_queued = fwd_ahead; }
      else undo();   // the launch stored nothing: converged / interrupted, or the relay had given up -- then the next turn queues the update itself
    }
    if (stop > 0) return stop;
    if (stop < 0) break;
  }
  return CA_OK;
}

int ca_run_ex(ca_handle h, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws, double* trace, int32_t* n_elbo,
              ca_poll_fn poll, void* user) {
  if (!h || !trace || !n_elbo || max_iter < 0) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  const int64_t need = 2 + 2 * (int64_t)max_iter;
  CACK(stage_eps(h, eps_stream, n_draws, need));
  CACK(ensure_elbo_cap(h, 1 + (int64_t)max_iter));
  *n_elbo = 0;
  h->in_run = true;
  const int rc = run_loop(h, max_iter, rel_tol, trace, n_elbo, poll, user);
  h->in_run = false; h->gate_open = false; h->gate_snap = nullptr; h->gate_fsnap = nullptr;
  return rc;
}

int ca_iterate(ca_handle h, int32_t n_iter, const float* eps_stream, int64_t n_draws, double* last_elbo) {
  if (!h || n_iter < 0) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  // ABI 6: with one draw MORE than the 2 n the call consumes (or the built-in stream, which can look one draw ahead), the last sweep carries the forward
  // half of the NEXT ca_iterate call's first train pass instead of a duplicate of its own draw, and that call -- if its first draw IS that draw, bit for
  // bit -- starts from it: back-to-back calls of n iterations then run n sweeps each, not n + 1 (a 20-iteration call paid 21/20 of the steady state).
  const int64_t per = (int64_t)h->S * h->G;
  const bool can_carry = n_iter > 0 && h->fused_ok && !h->s2 && (h->bwd_mfma || !is_sharded(h));
  const bool carry_out = can_carry && (eps_stream ? n_draws >= 2 * (int64_t)n_iter + 1 : true);
  bool carry_in = can_carry && h->carry && h->look_valid && h->look_slot == 0;
  if (carry_in) carry_in = eps_stream ? (!h->carry_builtin && (int64_t)h->carry_eps.size() == per && n_draws >= 1 &&
                                         memcmp(h->carry_eps.data(), eps_stream, (size_t)per * sizeof(float)) == 0)
                                      : h->carry_builtin;
  h->carry = false;
  CACK(stage_eps(h, eps_stream, n_draws, 2 * (int64_t)n_iter + (carry_out ? 1 : 0)));   // (clears the look-ahead: the staged slots change ...)
  if (!eps_stream && carry_out) h->draw -= 1;                                             // (built-in stream: the extra draw was a look ahead, the next call draws it again)
  if (carry_in) { h->look_valid = true; h->look_slot = 0; }                               // (... but slot 0 of the new block is the draw the carried half was made with)
  CACK(ensure_elbo_cap(h, std::max(1, n_iter) + 1));
  const auto t_host0 = std::chrono::steady_clock::now();
  // The first train pass has no monitor pass before it to share a sweep with.  Instead of the plain kernels (fp32 VALU sweep,
  // separate cell epilogue, the Y stream in line: 0.65 ms at cfg-3, 4 % of a 20-iteration call) its forward half takes the fused
  // matrix-core sweep with its own draw in both column halves; the monitor half's ELBO goes to a scratch slot.
  // (sharded with the general backward sweep the extra monitor tail would cost a collective of its own: plain kernels there)
  if (n_iter > 0 && h->fused_ok && !h->s2 && !h->look_valid && (h->bwd_mfma || !is_sharded(h)))
    CACK(fused_pass(h, 0, 0, h->elbo_dev + n_iter));
  for (int i = 0; i < n_iter; ++i) {
    // (the last monitor pass has no train pass to share its sweep with: its own draw in both halves, as above -- the plain
    //  kernels with the Y stream in line cost 0.33 ms at cfg-3 against 0.2)
    const int64_t mon = 2 * (int64_t)i + 1, next = (i + 1 < n_iter || carry_out) ? mon + 1 : (h->fused_ok ? mon : -1);
    h->hint_A = mon; h->hint_B = next;
    CACK(train_pass(h, 2 * (int64_t)i));
    CACK(monitor_pass(h, mon, next, h->elbo_dev + i));
  }
  CACK(flush_mon_tail(h));
  if (carry_out && h->look_valid && h->look_slot == 2 * (int64_t)n_iter) {
    h->carry = true; h->carry_builtin = eps_stream == nullptr; h->look_slot = 0;
    if (eps_stream) h->carry_eps.assign(eps_stream + 2 * (int64_t)n_iter * per, eps_stream + (2 * (int64_t)n_iter + 1) * per);
  } else {
    h->look_valid = false;   // the duplicate half of the last sweep is nobody's look-ahead
  }
  if (verbose(h) && n_iter > 0)
    fprintf(stderr, "[clonealign_hip] ca_iterate: host enqueue %.1f us per iteration\n",
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count() / n_iter);
  if (last_elbo && n_iter > 0) return read_doubles(h, h->elbo_dev + (n_iter - 1), last_elbo, 1);
  SYNC(h);
  return CA_OK;
}

int ca_final_elbo(ca_handle h, int32_t n_rep, const float* eps_stream, int64_t n_draws, double* values, double* mean, double* sd) {
  if (!h || n_rep < 1) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(stage_eps(h, eps_stream, n_draws, n_rep));
  CACK(ensure_elbo_cap(h, n_rep));
  // two draws per sweep where the fused matrix-core path exists (one exp per (cell, gene) serves both); single-shard only:
  // a sharded monitor pass has its own (3 + C)-double all-reduce
  const bool pairs = h->fused_ok && h->fwd_cell && !is_sharded(h) && h->pair_elbo;
  for (int i = 0; i < n_rep; ++i) {
    if (pairs && i + 1 < n_rep) { CACK(fused_pass(h, i, i + 1, h->elbo_dev + i, h->elbo_dev + i + 1)); ++i; }
    else if (h->fused_ok && (h->c16 || h->s2)) { CACK(monitor_pass(h, i, -1, h->elbo_dev + i)); CACK(flush_mon_tail(h)); }   // one pass per matrix-core sweep
    else CACK(run_pass(h, i, CA_MODE_ELBO, 0, h->elbo_dev + i));
  }
  std::vector<double> v((size_t)n_rep);
  HIPCK(h, hipMemcpyAsync(v.data(), h->elbo_dev, (size_t)n_rep * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  double m = 0.0;
  for (double x : v) m += x;
  m /= n_rep;
  double ss = 0.0;
  for (double x : v) ss += (x - m) * (x - m);
  if (values) memcpy(values, v.data(), (size_t)n_rep * sizeof(double));
  if (mean) *mean = m;
  if (sd) *sd = n_rep > 1 ? std::sqrt(ss / (n_rep - 1)) : NAN;
  return CA_OK;
}

#include "ca_eng_init.inc"   // C ABI, once per fit on the resident matrix: PCA initialisation of psi (blocked subspace iteration), per-clone gene sums
static int get_generic(ca_handle h, const char* name, double* out, bool grad) {
  if (!h || !name || !out) return CA_ERR_INVALID;
  HIPCK(h, hipSetDevice(h->device));
  CACK(gate_close(h));   // (called from a ca_run_ex poll hook while a gated launch waits: close that window first)
  ParamRef r;
  if (!find_param(h, name, grad, r)) { h->err = std::string("unknown parameter name: ") + name; return CA_ERR_INVALID; }
  if (r.rows * r.cols == 0) return CA_OK;
  if (r.d) {
    HIPCK(h, hipMemcpyAsync(out, r.d, (size_t)r.rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    return CA_OK;
  }
  std::vector<float> buf;
  CACK(download_f(h, buf, r.f, r.rows * std::max<int64_t>(r.stride, 1)));
  std::vector<double> row((size_t)std::max<int64_t>(r.cols, 1));
  if (r.xform == 4) {  // softmax over a vector
    double mx = -INFINITY, se = 0.0;
    for (int64_t i = 0; i < r.rows; ++i) mx = std::max(mx, (double)buf[i]);
    for (int64_t i = 0; i < r.rows; ++i) se += std::exp((double)buf[i] - mx);
    for (int64_t i = 0; i < r.rows; ++i) out[i] = std::exp((double)buf[i] - mx - std::log(se));
    return CA_OK;
  }
  for (int64_t i = 0; i < r.rows; ++i) {
    for (int64_t c = 0; c < r.cols; ++c) row[c] = (double)buf[i * r.stride + r.off + c];
    if (r.xform == 1) for (auto& x : row) x = x > 0 ? x + std::log1p(std::exp(-x)) : std::log1p(std::exp(x));
    if (r.xform == 3) for (auto& x : row) x = std::exp(x);
    if (r.xform == 2) {
      double mx = -INFINITY, se = 0.0;
      for (int64_t c = 0; c < r.cols; ++c) mx = std::max(mx, row[c]);
      for (int64_t c = 0; c < r.cols; ++c) se += std::exp(row[c] - mx);
      const double lse = mx + std::log(se);
      for (int64_t c = 0; c < r.cols; ++c) row[c] = std::exp(row[c] - lse);
    }
    for (int64_t c = 0; c < r.cols; ++c) out[r.matrix ? hidx(h->layout, i, c, r.rows, r.cols) : i] = row[c];
  }
  return CA_OK;
}

int ca_get_param(ca_handle h, const char* name, double* out) { return get_generic(h, name, out, false); }
int ca_get_gradient(ca_handle h, const char* name, double* out) { return get_generic(h, name, out, true); }

int ca_set_param(ca_handle h, const char* name, const double* in) {
  if (!h || !name || !in) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  ParamRef r;
  if (!find_param(h, name, false, r) || r.xform != 0 || r.d) { h->err = std::string("cannot set parameter: ") + name; return CA_ERR_INVALID; }
  if (r.rows * r.cols == 0) return CA_OK;
  std::vector<float> buf;
  CACK(download_f(h, buf, r.f, r.rows * std::max<int64_t>(r.stride, 1)));
  for (int64_t i = 0; i < r.rows; ++i)
    for (int64_t c = 0; c < r.cols; ++c)
      buf[i * r.stride + r.off + c] = (float)in[r.matrix ? hidx(h->layout, i, c, r.rows, r.cols) : i];
  CACK(upload_f(h, r.f, buf));
  CACK(refresh_derived(h));
  SYNC(h);
  return CA_OK;
}

// A new restart on the same data (R/clonealign.R:50-56 runs every restart through the whole of inference_tflow): all eight
// variables back to their initial values (R/inference-tflow.R:240-273), Adam slots and beta powers cleared; the count matrix, its
// fit constants and the column sums stay resident.
int ca_reinit(ca_handle h, const double* psi0, const double* loc0) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  if (h->K > 0 && !psi0) { h->err = "psi0 is required when K > 0"; return CA_ERR_INVALID; }
  HIPCK(h, hipSetDevice(h->device));
  CACK(wait_y(h, true));
  // A forward-sweep block that gave up on a left-over tile leaves a sticky word (comm_check: "the engine's state is undefined").  A restart defines every
  // variable and every Adam slot again, so THIS call -- and only this one -- clears it; a dead peer-to-peer transport stays dead.
  HIPCK(h, hipStreamSynchronize(h->stream));
  if (h->host_pinned) { *reinterpret_cast<volatile unsigned int*>(h->host_pinned + 41) = 0u; *reinterpret_cast<volatile unsigned int*>(h->host_pinned + 42) = 0u; }
  SYNC(h);
  const int64_t N = h->N; const int G = h->G, C = h->C, K = h->K, D = h->D;
  auto zero = [&](float* p, int64_t n) { return p && n > 0 ? hipMemsetAsync(p, 0, (size_t)n * sizeof(float), h->stream) : hipSuccess; };
  const int64_t GD = (int64_t)G * std::max(D, 1), NK = N * std::max(K, 1);
  HIPCK(h, zero(h->V, GD)); HIPCK(h, zero(h->m_V, GD)); HIPCK(h, zero(h->v_V, GD));
  HIPCK(h, zero(h->ls, G)); HIPCK(h, zero(h->m_ls, G)); HIPCK(h, zero(h->v_ls, G));
  HIPCK(h, zero(h->m_loc, G)); HIPCK(h, zero(h->v_loc, G));
  HIPCK(h, zero(h->vchi, std::max(K, 1))); HIPCK(h, zero(h->m_v, std::max(K, 1))); HIPCK(h, zero(h->v_v, std::max(K, 1)));
  HIPCK(h, zero(h->alpha_u, C)); HIPCK(h, zero(h->m_a, C)); HIPCK(h, zero(h->v_a, C));
  HIPCK(h, zero(h->glogit, N * C)); HIPCK(h, zero(h->m_gl, N * C)); HIPCK(h, zero(h->v_gl, N * C));
  HIPCK(h, zero(h->m_psi, NK)); HIPCK(h, zero(h->v_psi, NK));
  if (loc0) {
    std::vector<float> l0((size_t)G);
    for (int g = 0; g < G; ++g) l0[g] = (float)loc0[g];
    CACK(upload_f(h, h->loc, l0));
  } else {
    HIPCK(h, hipMemcpyAsync(h->loc, h->loc_init, (size_t)G * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  }
  if (K > 0) {   // psi columns of F; the covariate columns stay
    std::vector<float> Fh;
    CACK(download_f(h, Fh, h->F, N * D));
    for (int64_t n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) Fh[(size_t)n * D + k] = (float)psi0[hidx(h->layout, n, k, N, K)];
    CACK(upload_f(h, h->F, Fh));
  }
  h->b1p = (float)h->opt.beta1;
  h->b2p = (float)h->opt.beta2;
  h->mon_tail.enabled = 0;
  h->bwd_ready = false;
  h->hint_A = h->hint_B = -1;
  CACK(refresh_derived(h));
  SYNC(h);
  return CA_OK;
}

int ca_get_kernel_times(ca_handle h, double ms[CA_KERNEL_COUNT], int64_t launches[CA_KERNEL_COUNT]) {
  if (!h) return CA_ERR_INVALID;
  HIPCK(h, hipSetDevice(h->device));
  CACK(gate_close(h));   // (called from a ca_run_ex poll hook while a gated launch waits: close that window first)
  CACK(prof_flush(h));
  for (int i = 0; i < CA_KERNEL_COUNT; ++i) {
    if (ms) ms[i] = h->k_ms[i];
    if (launches) launches[i] = h->k_n[i];
  }
  return CA_OK;
}

int ca_reset_kernel_times(ca_handle h) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);   // (changes the profiling state around a launch that may be undone and queued again: not from a poll hook)
  CACK(prof_flush(h));
  for (int i = 0; i < CA_KERNEL_COUNT; ++i) { h->k_ms[i] = 0; h->k_n[i] = 0; }
  return CA_OK;
}

int ca_set_profile(ca_handle h, int32_t mask) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  CACK(prof_flush(h));
  h->opt.profile = mask;
  for (unsigned& c : h->prof_seen) c = 0;
  return CA_OK;
}

int ca_eps_draw(uint64_t seed, uint64_t draw, int64_t n, float* out) {
  if (!out || n < 0) return CA_ERR_INVALID;
  ca_philox::normal_draw(seed, draw, n, out);
  return CA_OK;
}

int ca_allele_loglik(int64_t N, int32_t V, int32_t C, int32_t layout, const double* clone_allele, const double* cov,
                     const double* ref, int32_t device, double* out, char* err) {
  auto fail = [&](int code, const std::string& m) { if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; } return code; };
  if (N < 1 || V < 1 || C < 1 || !clone_allele || !cov || !ref || !out) return fail(CA_ERR_INVALID, "ca_allele_loglik: bad arguments");
  if (layout != CA_ROW_MAJOR && layout != CA_COL_MAJOR) return fail(CA_ERR_INVALID, "ca_allele_loglik: bad layout");
  double *dcov = nullptr, *dref = nullptr, *dout = nullptr; unsigned char* dis2 = nullptr;
  auto cleanup = [&]() { hipFree(dcov); hipFree(dref); hipFree(dout); hipFree(dis2); };
#define ACK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(CA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
  ACK(hipSetDevice(device));
  std::vector<unsigned char> is2((size_t)V * C);
  for (int v = 0; v < V; ++v)
    for (int c = 0; c < C; ++c) is2[(size_t)v * C + c] = clone_allele[hidx(layout, v, c, V, C)] == 2.0 ? 1 : 0;
  const size_t nv = (size_t)N * V;
  ACK(hipMalloc((void**)&dcov, nv * sizeof(double)));
  ACK(hipMalloc((void**)&dref, nv * sizeof(double)));
  ACK(hipMalloc((void**)&dout, (size_t)N * C * sizeof(double)));
  ACK(hipMalloc((void**)&dis2, is2.size()));
  ACK(hipMemcpy(dcov, cov, nv * sizeof(double), hipMemcpyHostToDevice));
  ACK(hipMemcpy(dref, ref, nv * sizeof(double), hipMemcpyHostToDevice));
  ACK(hipMemcpy(dis2, is2.data(), is2.size(), hipMemcpyHostToDevice));
  const int64_t sn = layout == CA_COL_MAJOR ? 1 : V, sv = layout == CA_COL_MAJOR ? N : 1;
  const int64_t on = layout == CA_COL_MAJOR ? 1 : C, oc = layout == CA_COL_MAJOR ? N : 1;
  const int vtile = std::min(V, 4096);
  auto cab = [](double a, double b) { return std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b); };
  for (int64_t n0 = 0; n0 < N; n0 += 1 << 30) {   // grid.x limit
    const int64_t nb = std::min<int64_t>(N - n0, (int64_t)1 << 30);
    hipLaunchKernelGGL(k_allele_loglik, dim3((unsigned)nb), dim3(CA_TB), (size_t)vtile * sizeof(double), 0,
                       dcov + n0 * sn, dref + n0 * sn, sn, sv, dis2, dout + n0 * on, on, oc, nb, (int)V, (int)C, vtile,
                       cab(0.1, 1.9), cab(1.9, 0.1), cab(2.0, 2.0));
    ACK(hipGetLastError());
  }
  ACK(hipMemcpy(out, dout, (size_t)N * C * sizeof(double), hipMemcpyDeviceToHost));
#undef ACK
  cleanup();
  return CA_OK;
}

int ca_preprocess(int64_t N, int32_t G, int32_t C, int32_t layout, int32_t y_dtype, int32_t y_on_device, const void* Y,
                  const double* L, const ca_preprocess_params* params, int32_t device, uint8_t* keep_gene,
                  uint8_t* keep_cell, double* gene_sums, double* cell_sums, char* err) {
  auto fail = [&](int code, const std::string& m) { if (err) { strncpy(err, m.c_str(), 255); err[255] = 0; } return code; };
  if (N < 1 || G < 1 || C < 1 || !Y || !L || !params || !keep_gene || !keep_cell) return fail(CA_ERR_INVALID, "ca_preprocess: bad arguments");
  if (layout != CA_ROW_MAJOR && layout != CA_COL_MAJOR) return fail(CA_ERR_INVALID, "ca_preprocess: bad layout");
  if (hipSetDevice(device) != hipSuccess) return fail(CA_ERR_HIP, "ca_preprocess: hipSetDevice failed");
  const size_t esz = y_dtype == CA_F64 ? 8 : (y_dtype == CA_F32 || y_dtype == CA_I32) ? 4 : y_dtype == CA_U16 ? 2 : 1;
  const void* src = Y;
  void* staging = nullptr;
  if (!y_on_device) {
    if (hipMalloc(&staging, (size_t)N * G * esz) != hipSuccess) return fail(CA_ERR_NOMEM, "ca_preprocess: device allocation of the count matrix failed");
    // (the statistics are fixed-order fp64 sums of the caller's OWN values, so nothing is narrowed here; the runtime's pageable copy runs at 98 % of the pinned rate)
    if (hipMemcpy(staging, Y, (size_t)N * G * esz, hipMemcpyHostToDevice) != hipSuccess) { hipFree(staging); return fail(CA_ERR_HIP, "ca_preprocess: upload failed"); }
    src = staging;
  }
  std::string msg;
  int rc;
  switch (y_dtype) {
    case CA_F64: rc = preprocess_t<double>((const double*)src, N, G, C, layout, L, *params, keep_gene, keep_cell, gene_sums, cell_sums, msg); break;
    case CA_F32: rc = preprocess_t<float>((const float*)src, N, G, C, layout, L, *params, keep_gene, keep_cell, gene_sums, cell_sums, msg); break;
    case CA_I32: rc = preprocess_t<int32_t>((const int32_t*)src, N, G, C, layout, L, *params, keep_gene, keep_cell, gene_sums, cell_sums, msg); break;
    case CA_U16: rc = preprocess_t<uint16_t>((const uint16_t*)src, N, G, C, layout, L, *params, keep_gene, keep_cell, gene_sums, cell_sums, msg); break;
    case CA_U8: rc = preprocess_t<uint8_t>((const uint8_t*)src, N, G, C, layout, L, *params, keep_gene, keep_cell, gene_sums, cell_sums, msg); break;
    default: rc = CA_ERR_INVALID; msg = "ca_preprocess: unknown y_dtype";
  }
  if (staging) hipFree(staging);
  return rc == CA_OK ? CA_OK : fail(rc, msg);
}

}  // extern "C"
