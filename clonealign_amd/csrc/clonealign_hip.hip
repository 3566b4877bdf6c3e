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

namespace {

thread_local std::string g_last_error;

// ---- RCCL, loaded lazily so that single-GPU use never touches it
typedef struct { char internal[128]; } ca_nccl_uid;
typedef void* ca_nccl_comm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(ca_nccl_uid*) = nullptr;
  int (*CommInitRank)(ca_nccl_comm*, int, ca_nccl_uid, int) = nullptr;
  int (*CommDestroy)(ca_nccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ca_nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    const char* cands[] = {getenv("CLONEALIGN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (const char* c : cands) {
      if (!c || !*c) continue;
      lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce) { err = "librccl lacks expected symbols"; return false; }
    return true;
  }
};
RcclApi g_rccl;
constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;      // ncclSum

struct EvPair { hipEvent_t a, b; int kid; };

}  // namespace

struct ca_p2p {
  double* slab = nullptr; size_t slab_bytes = 0; int64_t cap = 0;
  std::vector<void*> opened;             // peers' slabs as mapped here (nullptr for ranks of this process' own slab)
  double** peers_dev = nullptr;          // device array [world]
  unsigned long long seq = 0;
  bool mapped = false;                   // ca_p2p_connect has mapped every peer's slab
  bool connected = false;                // ca_p2p_commit(1): the transport is the engine's all-reduce
  unsigned long long* err_host = nullptr;   // pinned: 0, or the sequence number of the first call whose wait for a peer ran out
  unsigned long long* err_dev = nullptr;    // the same word as the device sees it
  unsigned int* err_local = nullptr;        // device memory: non-zero once a call has given up (what the next calls check at entry)
  unsigned long long timeout_ticks = 0;     // bound of the device-side wait, s_memrealtime ticks (100 MHz)
};
struct ca_p2p_wire {   // what travels in a CA_P2P_HANDLE_BYTES handle
  hipIpcMemHandle_t mem; int64_t cap; int32_t rank, world, device, pid;
  uint64_t local_ptr;   // the slab's address in the exporting process: peers of the SAME process take it as it is (an IPC handle cannot be opened by the process that made it)
};
static_assert(sizeof(ca_p2p_wire) <= CA_P2P_HANDLE_BYTES, "handle too small");

struct ca_engine {
  // ---- problem
  int64_t N = 0;
  int G = 0, C = 0, K = 0, P = 0, S = 0, D = 0;  // D = K + P, or 0 when K == 0 (:279-285)
  int layout = 0;
  int nchunk = 0;
  ca_options opt{};
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream: the Y pass of the next parameter state runs beside the monitor pass's forward sweep
  hipStream_t stream2 = nullptr; hipEvent_t ev_params = nullptr, ev_ydone = nullptr; bool y_pending = false, async_y = true;
  std::string err;
  std::vector<void*> allocs;
  int64_t dev_bytes = 0;
  // ---- count matrix
  int ystore = 0, ybytes = 0, VEC = 0, Gp = 0, nseg = 0, TR = 0, nrb = 0, nrg = 0;   // nrg: rows of YTpart = blocks of 4 row blocks
  void* Y = nullptr;
  int64_t y_dev_bytes = 0;
  // overflow list of the u8 storage (entries > 255), CSR (by cell) and CSC (by gene) orders, host copy kept for setup
  int64_t n_ovf = 0;
  int64_t *ovf_rowptr = nullptr, *ovf_chunk_start = nullptr; int* ovf_col_chunk_ptr = nullptr; float* ovf_csum = nullptr; int n_ovf_chunk = 0;
  int *ovf_col = nullptr, *ovf_row2 = nullptr;
  float *ovf_val = nullptr, *ovf_val2 = nullptr;
  std::vector<int> h_orow, h_ocol; std::vector<float> h_oval;
  // ---- constants
  float* Lb = nullptr;       // [nchunk][G][8]
  double *A = nullptr, *cn = nullptr, *s64 = nullptr, *colsum = nullptr, *YtX = nullptr;
  float* s32 = nullptr;
  // ---- variables + Adam slots
  float *F = nullptr, *m_psi = nullptr, *v_psi = nullptr;           // F [N][D] = (psi | X)
  float *glogit = nullptr, *m_gl = nullptr, *v_gl = nullptr;        // [N][C]
  float *V = nullptr, *m_V = nullptr, *v_V = nullptr;               // V [G][D] = (W | beta)
  float *loc = nullptr, *ls = nullptr, *m_loc = nullptr, *v_loc = nullptr, *m_ls = nullptr, *v_ls = nullptr;
  float* loc_init = nullptr;   // loc as ca_create left it (loc0, or the device-side mu_guess): ca_reinit's default
  float *vchi = nullptr, *m_v = nullptr, *v_v = nullptr;            // [K]
  float *alpha_u = nullptr, *m_a = nullptr, *v_a = nullptr;         // [C]
  // round 4 (merged update, k_update_merged): the chi / alpha step writes the alternate buffers and the host swaps them in; the exponent
  // bound of the stepped state is then made by the next forward sweep itself (em_stale: nobody has made it yet)
  float *vchi_alt = nullptr, *alpha_u_alt = nullptr;
  bool upd_merge = false, em_stale = false;
  // ca_run: the update half of train pass i + 1 is queued before the host has seen ELBO i and gated on a word the host writes (ca_merge_args::gate)
  bool run_gate = false, gate_req = false, gate_armed = false;
  // round 5: the gated launch's relay block gives up after gate_ticks (a launch that stores nothing; not an error) -- gate_t0: host clock right before
  // that launch was queued; gate_open / gate_snap: the window between queuing it and answering it, in which a poll hook may call back into the API
  // (gate_close()); in_run: ca_run_ex is on this thread's stack (hooks may only call the read-only entry points)
  unsigned long long gate_ticks = 100000ull; std::chrono::steady_clock::time_point gate_t0;
  double ticks_per_us = 100.0;   // rate of s_memrealtime on this device (hipDeviceAttributeWallClockRate; 100 MHz on gfx950), the unit of every device-side time limit
  bool gate_open = false, gate_aborted = false, in_run = false; struct gate_snapshot* gate_snap = nullptr; struct fwd_snapshot* gate_fsnap = nullptr;
  bool run_fwd = false;    // ca_run: queue the forward sweep behind a gated update ahead of the host's decision (CA_VAR_RUN_FWD)
  bool fwd_gate = false;   // ca_run: the forward sweep being queued is behind a gated update and must look at that launch's answer (ca_cell_ptrs::gate)
  unsigned long long gate_seq = 0; unsigned long long* gate_local = nullptr;
  bool p2p_ride = false;   // sharded over the peer-to-peer transport: the sweep's column sums and the stream's finishing sums ride (allreduce(), train_bwd)
  double* gaux = nullptr; int64_t gaux_slot = -1;   // [2][5][G]: ca_merge_args::aux_in / aux_out, ping-pong; gaux_slot: the eps draw the current half belongs to (-1: none)
  int gaux_idx = 0;
  int* vmm_at = nullptr; int vmm_at_idx = 0; bool vmm_at_ready = false;   // [2][16] ordered-int range of V' (k_update_merged), the buffer in use alternates
  double dir_const = 0.0;
  float b1p = 0.f, b2p = 0.f;  // running beta powers, float32 like TF's beta*_power variables
  // ---- gradients (d ELBO / d var)
  float *g_loc = nullptr, *g_ls = nullptr, *g_V = nullptr, *g_v = nullptr, *g_a = nullptr, *g_psi = nullptr, *dgl = nullptr;
  // ---- per-pass buffers
  float* eps_dev = nullptr; int64_t eps_cap = 0;  // capacity in draws
  float *mu32 = nullptr, *Mb = nullptr, *Vs = nullptr, *vmm = nullptr, *vmm_part = nullptr, *etamax2 = nullptr;
  double* gene_part = nullptr; int ngblk = 0;
  float *Zpart = nullptr, *coef = nullptr; double* scratch = nullptr;
  double* cell_part = nullptr; int ncblk = 0;
  float *gpart = nullptr, *dFpart = nullptr; int gsplit = 1, gchunk = 0, csplit = 1; int64_t cchunk = 0; int RG = 4, ntile = 0;
  float *YWpart = nullptr, *YTpart = nullptr, *YW = nullptr; double* ytpsi = nullptr;
  double* red = nullptr; int64_t red_n = 0, off_g = 0, off_y = 0;
  double* elbo_dev = nullptr; int64_t elbo_cap = 0; double* terms_dev = nullptr;
  double* host_pinned = nullptr;  // 8 doubles
  bool ycache_valid = false, sums_global = false, sums_started = false;
  std::vector<double> mu_part;   // sharded with loc0 = NULL: this rank's per-gene sums of y_ng / rowMeans(Y)_n, completed over all cells in setup_global_sums
  // fused two-eps sweep (monitor pass of iteration i + forward half of train pass i+1, same parameters)
  bool fused_ok = false, look_valid = false; int64_t look_slot = 0; int frow = 8;
  // ca_iterate, ABI 6: the last sweep of a call may carry the forward half of the NEXT call's first train pass (draw 2n of a 2n + 1 draw stream); `carry` marks
  // that look-ahead as one a following ca_iterate may pick up, carry_eps keeps the draw it was made with (the next call's draw 0 must be this draw, bit for bit)
  bool carry = false, carry_builtin = false; std::vector<float> carry_eps;
  // the series form of the forward / backward contraction (ca_poly.hip; one exponent dimension, one MC sample, 3..8 clones): an overlay on the fused loop --
  // fused_pass makes Z by it and leaves the backward moments in the workspace (poly_fresh: they belong to the look-ahead half look_valid refers to),
  // train_bwd turns them into the per-gene sums, train_update reads d/dF from ONE slab (poly_df)
  // the host's look ahead at the exponent range (poly_guard): every pass of a series-capable engine leaves {seq, max|psi|, min W, max W} of ITS state in a
  // mapped ring; the pass with sequence number s decides from the entry s - CA_POLY_LAG exactly (waiting for it if need be: the queue is never more than
  // that deep) and the bound of an Adam step -- a deterministic decision, the same on every run, and never a truncated series
  int64_t n_series = 0, n_series_fallback = 0;
  double *poly_ring = nullptr, *poly_ring_dev = nullptr; uint64_t poly_seq = 0, poly_seq_base = 1, adam_steps = 0, poly_steps_at[16] = {};
  bool poly = false, poly_side = false, poly_y_defer = false, poly_fresh = false, poly_df = false /* the last backward half was the series form's: d/dF is ONE slab */; ca_poly_ws pws; float* poly_zero = nullptr; unsigned char* poly_mem = nullptr;
  float *Mb2 = nullptr, *mu32B = nullptr, *Zpart2 = nullptr; double* gene_partB = nullptr;
  bool y_defer = false;
  bool ride_ok = false;   // the Y stream's blocks ride on the forward sweep's launch (k_fwd_cell_mix_y) instead of a side stream
  bool ride_ys = false;   // ... as the one-copy int8 matrix-core stream (k_fwd_cell_mix_ys)
  bool ride_seq = false;  // ... fused in sequence into the sweep's own blocks (k_fwd_cell_seq_y)
  bool fold_gsum = false, fold_now = false;   // small problems: the backward sweep's partials are summed inside k_final_gene
  // per-gene prologue of the next fused pass, computed ahead by the train pass before it (ca_pre_args): the loops announce
  // the next (monitor, train) eps slots in hint_*, train_update fills the alternate partial buffers, fused_pass swaps them in
  int64_t hint_A = -1, hint_B = -1, pre_A = -1, pre_B = -1;
  bool pre_valid = false, pre_ok = true, pair_elbo = true;
  double* ee_partB = nullptr;   // pair sweeps: the second draw's per-block cell sums (ca_cell_ptrs::ee_partB)
  double *gene_part_alt = nullptr, *gene_partB_alt = nullptr;
  bool bwd_ready = false; int64_t bwd_slot = -1;
  double* yw_part = nullptr; int n_yw = 0;   // block partials of sum_n psi_n.(YW)_n (k_yw_dot)
  // the riding int8 stream without a finisher launch between the sweeps: its two finishing sums ride on the backward sweep as extra
  // blocks (ca_yfin_args); until that sweep is issued they are pending, and any other consumer gets the launch (yfin_flush)
  bool yfin_split = false, yfin_pending = false;
  ca_small_args mon_tail;          // pending ELBO assembly of a fused monitor pass: rides on the next backward sweep
  bool tail_fuse = true;
  double* host_dev = nullptr;      // device view of host_pinned
  unsigned long long host_seq = 0, host_seq_next = 0;
  float* eps_stage = nullptr; size_t eps_stage_bytes = 0;   // pinned staging buffer of the eps stream (built-in or the caller's)
  hipEvent_t ev_stage = nullptr;                             // completes when the last copy out of eps_stage has been made
  bool fwd_cell = false; int ncblk_f = 0, fc_tl = 4;   // forward sweep + cell epilogue in one kernel (k_fwd_cell)
  int fc_nbig = 0;                                     // > 0: k_fwd_cell_mix, that many blocks of 16 * fc_tl cells, the rest 32-cell blocks
  bool fwd_mfma = false; int fsplit = 1, fkchunk = 1, nk32 = 1; unsigned short* Mq = nullptr;   // matrix-core forward sweep
  // round 5: balanced forward sweep of small problems (k_fwd_bal_ys): bal_q tiles per block, bal_r left-over tiles in bal_nchunk gene chunks each
  bool fwd_bal = false; int bal_q = 0, bal_r = 0, bal_nchunk = 0; unsigned long long* bal_xw = nullptr; unsigned bal_tag = 0;
  // matrix-core backward sweep (k_bwd_mfma): bf16 parts of coef, its own cell split
  bool bwd_mfma = false, bwd_frac = false, c16 = false, s2 = false, s2f = false;   // s2f: mc_samples = 2 with monitor + next train pass in one sweep (CA_VAR_S2_FUSE)
  unsigned short* coefq = nullptr; int64_t N16 = 0, cchunk_m = 0; int csplit_m = 1, nwt = 0;
  int bwd_tl = CA_BWD_TL;   // gene tiles of 16 per wave in the matrix-core backward sweep: 4, or 3 for small problems (more, shorter wave jobs)
  uint64_t draw = 0;  // built-in stream position
  // count-matrix products on the int8 matrix cores (ca_ymfma.hip.h): tiled copies, fixed-point parameter images
  bool y_mfma = false;
  int64_t ym_NT = 0, ym_NS = 0, ym_schunk = 0; int ym_GS = 0, ym_GT = 0, ym_csplit = 1, ym_tl = 4;
  uint4 *Yf = nullptr, *Yb = nullptr, *Wq = nullptr, *Pq = nullptr; unsigned* ym_amax = nullptr; int* ym_out = nullptr;
  hipEvent_t ev_ywdone = nullptr; bool yw_pending = false, on_side = false;
  // ... and from ONE tiled copy through the transposing LDS read (k_ys_mfma; K = 1)
  bool y_ys = false; uint8_t* Ys = nullptr; uint4 *Wr = nullptr, *Pr = nullptr; int *Wsum = nullptr, *Psum = nullptr;
  int* ys_exps = nullptr;        // [3][2]: rotating slots, see ca_ys_quant_body
  unsigned* ys_amax = nullptr;   // [2]: exact maxima of a fresh state (k_ym_absmax), float bit patterns
  float* ys_amaxp = nullptr; int ys_nq = 0;   // [3][ys_ncap][2]: per-block maxima each quantiser run leaves for the next one
  int ys_ncap = 0, ys_namax[3] = {0, 0, 0};   // pairs a slot can hold / holds (a merged update leaves one pair per gene block and per psi block)
  bool ys_quant_ready = false;   // the images of the CURRENT parameter state were made by the quantiser riding on k_adam_cell
  int ys_slot = 0, ys_steps = -1, ys_RS = 256, ys_nrg = 0, ys_nseg = 0; int64_t ys_N64 = 0; float ys_step_bound = -1.f;
  // one-shot peer-to-peer all-reduce (ca_p2p_export / ca_p2p_connect)
  ca_p2p* p2p = nullptr;
  // ---- comm
  ca_nccl_comm comm = nullptr;
  ca_host_allreduce_fn host_ar = nullptr; void* host_ar_user = nullptr; double* host_ar_buf = nullptr; int64_t host_ar_cap = 0;
  // ---- profiling
  std::vector<EvPair> ev_pool; size_t ev_used = 0; bool prof_open = false;
  unsigned prof_seen[CA_KERNEL_COUNT] = {0, 0, 0, 0, 0};   // launches per class since ca_set_profile (sampling stride)
  double k_ms[CA_KERNEL_COUNT] = {0}; int64_t k_n[CA_KERNEL_COUNT] = {0};
  int n_cu = 256;
};

namespace {

#define HIPCK(h, call)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return CA_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define CACK(call)                \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != CA_OK) return rc_; \
  } while (0)

// The peer-to-peer all-reduce gives up inside the kernel when a peer does not show (k_p2p_allreduce); the host learns it here,
// at every point where it has waited for the stream anyway.
int comm_check(ca_engine* h) {
  if (h->host_pinned && *reinterpret_cast<volatile unsigned int*>(h->host_pinned + 41) != 0u) {
    h->err = "a forward-sweep block gave up waiting for a chunk of a left-over tile (k_fwd_bal_ys); the engine's state is undefined";
    return CA_ERR_STATE;
  }
  if (h->host_pinned && *reinterpret_cast<volatile unsigned int*>(h->host_pinned + 42) != 0u) {
    h->err = "the series form of the contraction (CA_VARX_SERIES) cannot cover this fit's exponent range: max|psi| (max W - min W) exceeds 4 x " +
             std::to_string(CA_PL_NB) + " bins' worth; the passes since are truncated -- start over without the series form";
    return CA_ERR_STATE;
  }
  if (h->p2p && h->p2p->err_host && *reinterpret_cast<volatile unsigned long long*>(h->p2p->err_host) != 0ull) {
    h->err = "peer-to-peer all-reduce #" + std::to_string(*reinterpret_cast<volatile unsigned long long*>(h->p2p->err_host)) +
             ": a peer's data did not arrive within the time limit (peer lost or out of step); this engine's transport is dead -- "
             "destroy the engine and start over in a fresh process";
    return CA_ERR_COMM;
  }
  return CA_OK;
}
#define SYNC(h)                                          \
  do {                                                   \
    HIPCK(h, hipStreamSynchronize((h)->stream));         \
    CACK(comm_check(h));                                 \
  } while (0)

template <typename T>
int dalloc(ca_engine* h, T** p, int64_t n) {
  if (n <= 0) n = 1;
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, (size_t)n * sizeof(T));
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc of ") + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e);
    return CA_ERR_NOMEM;
  }
  // Zeroed, and the zeroing DONE before anybody can touch the buffer.  The engine's streams are non-blocking streams: a hipMemcpy on the NULL stream
  // into a buffer whose hipMemsetAsync is still queued on h->stream is not ordered behind it.  Alone on the GPU the memset ran at once and the
  // order came out right by luck; with ANOTHER PROCESS keeping the GPU busy it ran late and zeroed what had just been uploaded -- the overflow
  // list of a 1-byte matrix, i.e. wrong fit constants for every cell with a count above 255 (round 4: profiles/r04_flake.txt).
  e = hipMemsetAsync(q, 0, (size_t)n * sizeof(T), h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = hipGetErrorString(e); return CA_ERR_HIP; }
  h->allocs.push_back(q);
  h->dev_bytes += n * (int64_t)sizeof(T);
  *p = (T*)q;
  return CA_OK;
}

// ---- profiling wrappers ------------------------------------------------------------------
int prof_flush(ca_engine* h) {
  if (h->ev_used == 0) return CA_OK;
  SYNC(h);
  if (h->stream2) HIPCK(h, hipStreamSynchronize(h->stream2));
  for (size_t i = 0; i < h->ev_used; ++i) {
    float ms = 0.f;
    HIPCK(h, hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
    h->k_ms[h->ev_pool[i].kid] += ms;
    h->k_n[h->ev_pool[i].kid] += 1;
  }
  h->ev_used = 0;
  return CA_OK;
}
int prof_begin(ca_engine* h, int kid) {
  h->prof_open = false;
  if (!((h->opt.profile >> kid) & 1)) return CA_OK;
  // bits 8..15 of the mask: sampling stride - 1.  An event pair costs the stream 5-6 us (two markers the packet processor drains
  // the queue for): timing every 8th launch keeps the live measurement and leaves the timed region alone.
  const unsigned stride = ((unsigned)h->opt.profile >> 8 & 0xFFu) + 1u;
  if (h->prof_seen[kid]++ % stride != 0) return CA_OK;
  if (h->ev_used == h->ev_pool.size()) {
    if (h->ev_pool.size() < 2048) {
      EvPair p; p.kid = kid;
      HIPCK(h, hipEventCreate(&p.a));
      HIPCK(h, hipEventCreate(&p.b));
      h->ev_pool.push_back(p);
    } else {
      CACK(prof_flush(h));
    }
  }
  h->ev_pool[h->ev_used].kid = kid;
  HIPCK(h, hipEventRecord(h->ev_pool[h->ev_used].a, h->stream));
  h->prof_open = true;
  return CA_OK;
}
int prof_end(ca_engine* h) {
  if (!h->prof_open) return CA_OK;
  h->prof_open = false;
  HIPCK(h, hipEventRecord(h->ev_pool[h->ev_used].b, h->stream));
  h->ev_used++;
  return CA_OK;
}
#define LAUNCH(h, kid, ...)                 \
  do {                                      \
    CACK(prof_begin(h, kid));               \
    __VA_ARGS__;                            \
    HIPCK(h, hipGetLastError());            \
    CACK(prof_end(h));                      \
  } while (0)

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// grid of `threads`-wide blocks for `total` elements, one thread each, flattened by ca_flat_index(): x stays below 2^22 blocks (2^30 work-items)
inline dim3 ca_grid_flat(int64_t total, int threads) {
  const int64_t nb = (total + threads - 1) / threads;
  const int64_t gx = std::min<int64_t>(nb, (int64_t)1 << 22);
  return dim3((unsigned)std::max<int64_t>(gx, 1), (unsigned)std::max<int64_t>((nb + gx - 1) / std::max<int64_t>(gx, 1), 1));
}

// Configuration comes from ca_options (variant_off / variant_on / tune).  The RELEASE library reads no tuning from the process
// environment at all; a timing-lab build (-DCA_LAB, tools/lab/) consults it when CLONEALIGN_DEBUG_ENV is set (tools/tune.py,
// tools/fuzz_parity.py): NAME=0 switches a variant off, NAME=<n> sets a parameter.
#ifdef CA_LAB
inline bool debug_env() { return getenv("CLONEALIGN_DEBUG_ENV") != nullptr; }
#else
constexpr bool debug_env() { return false; }
#endif
inline bool variant_on(const ca_engine* h, unsigned bit, const char* env) {
  if (h->opt.variant_off & bit) return false;
  if (debug_env()) if (const char* e = getenv(env)) return atoi(e) != 0;
  return true;
}
inline bool variantx_on(const ca_engine* h, unsigned bit, const char* env) {   // opt-in variants (ca_options.variant_on)
  if (h->opt.variant_on & bit) return true;
  if (debug_env()) if (const char* e = getenv(env)) return atoi(e) != 0;
  return false;
}
inline int tune_val(const ca_engine* h, int id, const char* env) {
  int v = h->opt.tune[id];
  if (v == 0 && debug_env()) if (const char* e = getenv(env)) { v = atoi(e); if (id == CA_TUNE_FC_NBIG && v == 0) v = -1; }
  return v;
}
inline bool verbose(const ca_engine* h) { return (h->opt.variant_off & CA_OPT_VERBOSE) || (debug_env() && getenv("CA_VERBOSE")); }

// Split count s in [1, smax] for a grid of xb * s equal blocks on `slots` resident block slots: fewest rounds per unit
// of work (a grid just past a multiple of the slots runs a nearly empty last round), every extra split charged
// `penalty` (its partial results are written and re-read).  Ties go to the smaller split.
inline int pick_split(int64_t xb, int64_t slots, int smax, double penalty) {
  int best = 1;
  double best_score = -1e30;
  for (int s = 1; s <= smax; ++s) {
    const int64_t B = xb * s;
    const int64_t rounds = (B + slots - 1) / slots;
    const double score = (double)B / (double)(rounds * slots) - penalty * s;
    if (score > best_score + 1e-9) { best_score = score; best = s; }
  }
  return best;
}

// ---- template dispatch of the sweeps -------------------------------------------------------
constexpr int kFwdR = 2;  // cells per lane of the forward sweep (R = 4 measured 5 % slower with 16 columns)
template <int NC>
void fwd_nc(int D, dim3 grid, hipStream_t st, const float* F, const float* em, const float* Vs, const float* M, float* Zp,
            int64_t N, int G, int gchunk) {
  const size_t lds = (size_t)gchunk * (CA_CW + (D > 0 ? D : 0)) * sizeof(float);
  switch (D) {
    case 0: hipLaunchKernelGGL((k_fwd_lds<NC, 0, kFwdR>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    case 1: hipLaunchKernelGGL((k_fwd_lds<NC, 1, kFwdR>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    case 2: hipLaunchKernelGGL((k_fwd_lds<NC, 2, kFwdR>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    default: hipLaunchKernelGGL((k_fwd_lds<NC, -1, kFwdR>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
  }
}
void launch_fwd(int nc, int D, dim3 grid, hipStream_t st, const float* F, const float* em, const float* Vs, const float* M,
                float* Zp, int64_t N, int G, int gchunk) {
  switch (nc) {
    case 1: fwd_nc<1>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 2: fwd_nc<2>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 3: fwd_nc<3>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 4: fwd_nc<4>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 5: fwd_nc<5>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 6: fwd_nc<6>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 7: fwd_nc<7>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    default: fwd_nc<8>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
  }
}

template <int NC>
void fwd16_nc(int D, dim3 grid, hipStream_t st, const float* F, const float* em, const float* Vs, const float* M, float* Zp,
              int64_t N, int G, int gchunk) {
  const size_t lds = (size_t)gchunk * (16 + (D > 0 ? D : 0)) * sizeof(float);
  switch (D) {
    case 0: hipLaunchKernelGGL((k_fwd_lds<NC, 0, kFwdR, 16>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    case 1: hipLaunchKernelGGL((k_fwd_lds<NC, 1, kFwdR, 16>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    case 2: hipLaunchKernelGGL((k_fwd_lds<NC, 2, kFwdR, 16>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
    default: hipLaunchKernelGGL((k_fwd_lds<NC, -1, kFwdR, 16>), grid, dim3(CA_TB), lds, st, F, em, Vs, M, Zp, N, G, gchunk, D); break;
  }
}
// forward sweep over 2*C columns [A | B] of a fused two-eps pass; row stride 8 when 2C <= 8, else 16
void launch_fwd_fused(int C, int D, dim3 grid, hipStream_t st, const float* F, const float* em, const float* Vs, const float* M,
                      float* Zp, int64_t N, int G, int gchunk) {
  switch (2 * C) {
    case 10: fwd16_nc<10>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 12: fwd16_nc<12>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 14: fwd16_nc<14>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    case 16: fwd16_nc<16>(D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;
    default: launch_fwd(2 * C, D, grid, st, F, em, Vs, M, Zp, N, G, gchunk); break;   // 2, 4, 6, 8 columns
  }
}

struct BwdArgs {
  const float *coef, *F, *em, *Lb, *mu, *Vs, *V;
  float *gpart, *dFpart;
  int64_t N; int G; int64_t cchunk; int D, S, sidx, first_s, first;
};
template <int NC, int RG>
void bwd_nc(dim3 grid, hipStream_t st, const BwdArgs& a) {
#define CA_BWD(DT)                                                                                                   \
  hipLaunchKernelGGL((k_bwd<NC, DT, RG>), grid, dim3(CA_TB), 0, st, a.coef, a.F, a.em, a.Lb, a.mu, a.Vs, a.V, a.gpart, \
                     a.dFpart, a.N, a.G, a.cchunk, a.D, a.S, a.sidx, a.first_s, a.first)
  switch (a.D) {
    case 0: CA_BWD(0); break;
    case 1: CA_BWD(1); break;
    case 2: CA_BWD(2); break;
    default: CA_BWD(-1); break;
  }
#undef CA_BWD
}
template <int RG>
void bwd_rg(int nc, dim3 grid, hipStream_t st, const BwdArgs& a) {
  switch (nc) {
    case 1: bwd_nc<1, RG>(grid, st, a); break;
    case 2: bwd_nc<2, RG>(grid, st, a); break;
    case 3: bwd_nc<3, RG>(grid, st, a); break;
    case 4: bwd_nc<4, RG>(grid, st, a); break;
    case 5: bwd_nc<5, RG>(grid, st, a); break;
    case 6: bwd_nc<6, RG>(grid, st, a); break;
    case 7: bwd_nc<7, RG>(grid, st, a); break;
    default: bwd_nc<8, RG>(grid, st, a); break;
  }
}
void launch_bwd(int RG, int nc, dim3 grid, hipStream_t st, const BwdArgs& a) {
  if (RG == 1) bwd_rg<1>(nc, grid, st, a);
  else if (RG == 8) bwd_rg<8>(nc, grid, st, a);
  else bwd_rg<4>(nc, grid, st, a);
}

template <typename YT>
void ypass_t(ca_engine* h, int koff, int kk, dim3 grid, const ca_ovf_args& ovf) {
  const YT* Y = (const YT*)h->Y;
  const int nb_main = grid.x;
  grid.x += ovf.nb_rows + ovf.nb_chunks;
#define CA_YP(KK)                                                                                                        \
  hipLaunchKernelGGL((k_ypass<YT, KK>), grid, dim3(CA_TB), 0, h->stream, Y, h->F, h->D, h->V, koff, h->YWpart, h->YTpart, \
                     h->N, h->G, h->Gp, h->nseg, h->nrb, h->TR, h->K, ovf, nb_main)
  switch (kk) {
    case 1: CA_YP(1); break;
    case 2: CA_YP(2); break;
    case 3: CA_YP(3); break;
    default: CA_YP(4); break;
  }
#undef CA_YP
}

// ---- host matrix helpers ------------------------------------------------------------------
// element (r, c) of an R x Cn host matrix in the problem's layout
inline int64_t hidx(int layout, int64_t r, int64_t c, int64_t R, int64_t Cn) {
  return layout == CA_COL_MAJOR ? c * R + r : r * Cn + c;
}

int upload_f(ca_engine* h, float* dst, const std::vector<float>& v) {
  if (v.empty()) return CA_OK;
  HIPCK(h, hipMemcpyAsync(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  SYNC(h);
  return CA_OK;
}
int upload_d(ca_engine* h, double* dst, const std::vector<double>& v) {
  if (v.empty()) return CA_OK;
  HIPCK(h, hipMemcpyAsync(dst, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SYNC(h);
  return CA_OK;
}
int download_f(ca_engine* h, std::vector<float>& v, const float* src, int64_t n) {
  v.resize((size_t)n);
  if (n == 0) return CA_OK;
  HIPCK(h, hipMemcpyAsync(v.data(), src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  return CA_OK;
}

// The side stream's products: the row products (YW, psi.(YW) partials) are done at ev_ywdone, everything (also Y^T psi)
// at ev_ydone.  all = false waits for the row products only (the backward sweep's ELBO tail needs nothing else).
int wait_y(ca_engine* h, bool all) {
  if (all) {
    if (h->y_pending) HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_ydone, 0));
    h->y_pending = false; h->yw_pending = false;
  } else if (h->yw_pending) {
    HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_ywdone, 0));
    h->yw_pending = false;
  }
  return CA_OK;
}

// After a merged update (k_update_merged) the per-cell exponent bound of the stepped state has not been made: the fused forward sweep's
// blocks make it themselves (ca_cell_ptrs::vmm_part); every OTHER consumer of etamax2 calls this first (the two small kernels of setup).
int ensure_etamax(ca_engine* h) {
  if (!h->em_stale) return CA_OK;
  if (h->D > 0) {
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_vmm_final, dim3(1), dim3(64), 0, h->stream, h->vmm_part, h->vmm, h->ngblk, h->D));
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_etamax, dim3(cdiv(h->N, CA_TB)), dim3(CA_TB), 0, h->stream, h->F, h->vmm, h->etamax2, h->N, h->D));
  }
  h->em_stale = false;
  return CA_OK;
}

// ---- derived state that depends on the parameters only (not on eps) -------------------------
int refresh_derived(ca_engine* h) {
  CACK(wait_y(h, true));
  h->y_defer = false;   // a deferred side-stream Y pass would not be ordered after this parameter change: redo it in line
  h->pre_valid = false;
  h->ys_steps = -1;     // arbitrary parameter change: the fixed-point exponents are taken from exact maxima again
  h->ys_quant_ready = false;
  if (h->D > 0) {
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_vprep, dim3(h->ngblk), dim3(CA_TB), 0, h->stream, h->V, h->Vs, h->vmm_part, h->G, h->D));
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_vmm_final, dim3(1), dim3(64), 0, h->stream, h->vmm_part, h->vmm, h->ngblk, h->D));
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_etamax, dim3(cdiv(h->N, CA_TB)), dim3(CA_TB), 0, h->stream, h->F, h->vmm, h->etamax2, h->N, h->D));
  }
  h->em_stale = false;
  h->gaux_slot = -1;
  h->ycache_valid = false;
  h->yfin_pending = false;
  h->look_valid = false;
  h->poly_fresh = false;
  h->poly_seq_base = h->poly_seq + 1;   // (an arbitrary parameter change: ranges seen before it say nothing)
  return CA_OK;
}

// The same two products on the int8 matrix cores (ca_ymfma.hip.h): fixed-point images of W and psi, then one stream
// over each tiled copy of the count matrix.  Runs on h->stream (the side stream when deferred).
template <int TL, int DEPTH>
void launch_yw(ca_engine* h) {
  hipLaunchKernelGGL((k_yw_mfma<TL, DEPTH>), dim3(cdiv(h->ym_NT, 4 * TL)), dim3(CA_YM_TB), 0, h->stream, h->Yf, h->Wq, h->ym_NT, h->ym_GS, h->N,
                     h->K, h->F, h->D, h->V, h->D, h->ym_amax, h->n_ovf > 0 ? h->ovf_rowptr : nullptr, h->ovf_col, h->ovf_val, h->YW, h->yw_part);
}
int ycache_mfma(ca_engine* h) {
  HIPCK(h, hipMemsetAsync(h->ym_amax, 0, 2 * sizeof(unsigned), h->stream));
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_ym_absmax, dim3(cdiv(std::max<int64_t>(h->N, h->G), CA_YM_TB)), dim3(CA_YM_TB), 0, h->stream,
                                                h->V, h->D, (int64_t)h->G, h->F, h->D, h->N, h->K, h->ym_amax));
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_ym_quant, dim3(cdiv((h->ym_GS + h->ym_NS) * 64, CA_YM_TB)), dim3(CA_YM_TB), 0, h->stream,
                                                h->V, h->D, (int64_t)h->G, h->ym_GS, h->F, h->D, h->N, h->ym_NS, h->K, h->ym_amax, h->Wq, h->Pq));
  // row products: YW and the psi.(YW) partials, one block per 64 * TL cells (n_yw blocks when TL = 4)
  CACK(prof_begin(h, CA_KERNEL_YPASS));
  if (h->ym_tl == 4) launch_yw<4, 2>(h);
  else if (h->ym_tl == 2) launch_yw<2, 4>(h);
  else launch_yw<1, 8>(h);
  HIPCK(h, hipGetLastError());
  CACK(prof_end(h));
  if (h->on_side) { HIPCK(h, hipEventRecord(h->ev_ywdone, h->stream)); h->yw_pending = true; }
  // column products: digit sums per cell slice, then Y^T psi into red_y
  ca_ovf_args ovf;
  memset(&ovf, 0, sizeof(ovf));
  int nb_ovf = 0;
  if (h->n_ovf > 0) {
    nb_ovf = cdiv(h->n_ovf_chunk, CA_TB / 64);
    ovf.chunk_start = h->ovf_chunk_start; ovf.row2 = h->ovf_row2; ovf.val2 = h->ovf_val2; ovf.csum = h->ovf_csum; ovf.nchunk = h->n_ovf_chunk;
  }
  const int nb_main = cdiv(h->ym_GT, 4 * 2);
  LAUNCH(h, CA_KERNEL_YPASS, hipLaunchKernelGGL((k_yt_mfma<2, 2>), dim3(nb_main + nb_ovf, h->ym_csplit), dim3(CA_YM_TB), 0, h->stream, h->Yb, h->Pq,
                                                h->ym_GT, h->ym_NS, h->ym_schunk, h->ym_out, nb_main, ovf, h->F, h->D, h->K));
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_yt_finish, dim3(cdiv((int64_t)h->G * h->K, CA_TB)), dim3(CA_TB), 0, h->stream, h->ym_out,
                                                h->ym_csplit, h->ym_GT, h->G, h->K, h->ym_amax, h->n_ovf > 0 ? h->ovf_col_chunk_ptr : nullptr,
                                                h->n_ovf > 0 ? h->ovf_csum : nullptr, h->red + h->off_y));
  h->ycache_valid = true;
  return CA_OK;
}

// Both products from ONE tiled copy through the transposing LDS read (k_ys_mfma): a quantiser launch (fixed-point images of W and
// psi; exact maxima by a separate pass only for the first state after a reset, afterwards bounded from the previous state's),
// the stream, the finisher.  Three launches, like the VALU stream's.
// Arguments of the quantiser for the current parameters (slot ys_slot).  lagged = true: the exponents are bounded from the previous
// state's maxima (`steps` Adam steps ago) -- possible up to 4 steps back; else the caller runs k_ym_absmax first (exact maxima).
ca_ysq_args ys_quant_args(ca_engine* h, int steps, bool* lagged) {
  ca_ysq_args a;
  memset(&a, 0, sizeof(a));
  const int s0 = h->ys_slot, s1 = (s0 + 1) % 3;
  a.nblk = h->ys_nq;
  a.V = h->V; a.Dv = h->D; a.G = h->G; a.GS = h->Gp / 64; a.F = h->F; a.Df = h->D; a.N = h->N; a.NS = h->ys_N64 / 64;
  *lagged = steps >= 0 && steps <= 4 && h->ys_step_bound > 0.f;
  if (*lagged) {
    a.amax_in = h->ys_amaxp + (int64_t)s0 * h->ys_ncap * 2; a.n_in = h->ys_namax[s0];
    a.slack_w = a.slack_p = (float)steps * h->ys_step_bound;
  } else {
    a.amax_in = reinterpret_cast<const float*>(h->ys_amax); a.n_in = 1;
  }
  a.amax_out = h->ys_amaxp + (int64_t)s1 * h->ys_ncap * 2;
  h->ys_namax[s1] = h->ys_nq;   // (what a run of the quantiser's own blocks leaves; the merged update overrides it)
  a.exps = h->ys_exps + 2 * s0; a.Wr = h->Wr; a.Pr = h->Pr; a.Wsum = h->Wsum; a.Psum = h->Psum;
  return a;
}
// quantiser of the one-copy stream as launches of its own (first pass after a reset, call-by-call API); inside the loop it rides
// on k_adam_cell (train_update) and this is a no-op
int ys_quant(ca_engine* h) {
  if (h->ys_quant_ready) return CA_OK;
  bool lagged = false;
  ca_ysq_args a = ys_quant_args(h, h->ys_steps, &lagged);
  if (!lagged) {
    HIPCK(h, hipMemsetAsync(h->ys_amax, 0, 2 * sizeof(unsigned), h->stream));
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_ym_absmax, dim3(cdiv(std::max<int64_t>(h->N, h->G), CA_YM_TB)), dim3(CA_YM_TB), 0, h->stream,
                                                  h->V, h->D, (int64_t)h->G, h->F, h->D, h->N, 1, h->ys_amax));
  }
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_ys_quant, dim3(a.nblk), dim3(CA_YM_TB), 0, h->stream, a));
  return CA_OK;
}
ca_ys_io ys_io(ca_engine* h) {
  ca_ys_io io;
  io.Wr = h->Wr; io.Pr = h->Pr; io.Wsum = h->Wsum; io.Psum = h->Psum; io.exps = h->ys_exps + 2 * h->ys_slot;
  io.YWpart = h->YWpart; io.YTpart = h->YTpart;
  return io;
}
ca_ovf_args ys_ovf(ca_engine* h) {
  ca_ovf_args ovf;
  memset(&ovf, 0, sizeof(ovf));
  if (h->n_ovf > 0) {
    ovf.nb_rows = cdiv(h->N, CA_TB); ovf.nb_chunks = cdiv(h->n_ovf_chunk, CA_TB / 64);
    ovf.rowptr = h->ovf_rowptr; ovf.col = h->ovf_col; ovf.val = h->ovf_val; ovf.YWextra = h->YWpart + (int64_t)h->ys_nseg * h->N;
    ovf.chunk_start = h->ovf_chunk_start; ovf.row2 = h->ovf_row2; ovf.val2 = h->ovf_val2; ovf.csum = h->ovf_csum; ovf.nchunk = h->n_ovf_chunk;
  }
  return ovf;
}
// finisher of the one-copy stream: the vector stream's own (k_yfinish) over the float partial slabs the stream left; advances
// the quantiser's slot ring
int yfin_flush(ca_engine* h) {
  if (!h->yfin_pending) return CA_OK;
  h->yfin_pending = false;
  const int nb_col = cdiv((int64_t)h->Gp, 64);
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_yfinish, dim3(nb_col + h->n_yw), dim3(1024), 0, h->stream, h->YTpart, h->red + h->off_y, h->ys_nrg,
                                                (int64_t)h->Gp, h->Gp, h->n_ovf > 0 ? h->ovf_col_chunk_ptr : nullptr,
                                                h->n_ovf > 0 ? h->ovf_csum : nullptr, 1, h->G, nb_col, h->YWpart,
                                                h->ys_nseg + (h->n_ovf > 0 ? 1 : 0), h->F, h->D, h->N, h->YW, h->yw_part));
  return CA_OK;
}
// the same sums as extra blocks of the backward sweep (k_bwd_mfma)
ca_yfin_args yfin_args(ca_engine* h) {
  ca_yfin_args a;
  memset(&a, 0, sizeof(a));
  a.ncol = cdiv((int64_t)h->Gp, 64); a.nrow = h->n_yw;
  a.part = h->YTpart; a.out = h->red + h->off_y; a.rows = h->ys_nrg; a.ld = h->Gp; a.cols = h->Gp; a.G = h->G;
  if (h->n_ovf > 0) { a.col_chunk_ptr = h->ovf_col_chunk_ptr; a.csum = h->ovf_csum; }
  a.YWpart = h->YWpart; a.nseg = h->ys_nseg + (h->n_ovf > 0 ? 1 : 0); a.F = h->F; a.D = h->D; a.N = h->N; a.YW = h->YW; a.yw_part = h->yw_part;
  return a;
}
// the stream's slabs are complete (launch issued): finisher now, or left pending for the backward sweep that follows in the loop;
// advances the quantiser's slot ring
int ys_finish(ca_engine* h, bool defer = false) {
  h->yfin_pending = true;
  if (!defer) CACK(yfin_flush(h));
  h->ys_slot = (h->ys_slot + 1) % 3;
  h->ys_steps = 0;
  h->ys_quant_ready = false;
  h->ycache_valid = true;
  return CA_OK;
}
int ycache_ys(ca_engine* h) {
  CACK(ys_quant(h));
  const int nb_main = h->ys_nrg * h->ys_nseg;
  if (h->n_ovf > 0) {
    const ca_ovf_args ovf = ys_ovf(h);
    LAUNCH(h, CA_KERNEL_YPASS, hipLaunchKernelGGL(k_ys_mfma_ovf, dim3(nb_main + ovf.nb_rows + ovf.nb_chunks), dim3(CA_YM_TB), CA_YS_LDS_BYTES,
                                                  h->stream, h->Ys, ys_io(h), h->N, h->Gp, h->ys_RS, nb_main, ovf, h->F, h->V, h->D));
  } else {
    LAUNCH(h, CA_KERNEL_YPASS, hipLaunchKernelGGL(k_ys_mfma, dim3(nb_main), dim3(CA_YM_TB), CA_YS_LDS_BYTES, h->stream, h->Ys, ys_io(h), h->N,
                                                  h->Gp, h->ys_RS));
  }
  return ys_finish(h);
}

// Y.W and Y^T.psi for the current parameters (once per parameter state)
int ensure_ycache(ca_engine* h) {
  if (h->y_defer) {   // deferred side-stream start (train_tail): ordered after the parameter update by ev_params
    h->y_defer = false;
    HIPCK(h, hipStreamWaitEvent(h->stream2, h->ev_params, 0));
    std::swap(h->stream, h->stream2);
    h->on_side = true;
    const int rc = ensure_ycache(h);
    h->on_side = false;
    std::swap(h->stream, h->stream2);
    CACK(rc);
    HIPCK(h, hipEventRecord(h->ev_ydone, h->stream2));
    h->y_pending = true;
    if (!h->yw_pending) {   // (the VALU stream has no earlier point: its row products are finished by its last launch)
      HIPCK(h, hipEventRecord(h->ev_ywdone, h->stream2));
      h->yw_pending = true;
    }
    return CA_OK;
  }
  if (h->ycache_valid || h->K == 0) { h->ycache_valid = true; return yfin_flush(h); }
  h->yfin_pending = false;
  if (h->y_ys) return ycache_ys(h);
  if (h->y_mfma) return ycache_mfma(h);
  dim3 grid((unsigned)((int64_t)h->nrg * h->nseg));
  // entries above 255: one extra "segment" of YW and one extra term of Y^T psi; their per-entry work rides on the
  // first stream launch, the per-gene sums on the column-sum launch
  ca_ovf_args ovf;
  memset(&ovf, 0, sizeof(ovf));
  if (h->n_ovf > 0) {
    ovf.nb_rows = cdiv(h->N, CA_TB); ovf.nb_chunks = cdiv(h->n_ovf_chunk, CA_TB / 64);
    ovf.rowptr = h->ovf_rowptr; ovf.col = h->ovf_col; ovf.val = h->ovf_val; ovf.YWextra = h->YWpart + (int64_t)h->nseg * h->N * h->K;
    ovf.chunk_start = h->ovf_chunk_start; ovf.row2 = h->ovf_row2; ovf.val2 = h->ovf_val2; ovf.csum = h->ovf_csum; ovf.nchunk = h->n_ovf_chunk;
  }
  ca_ovf_args none;
  memset(&none, 0, sizeof(none));
  for (int koff = 0; koff < h->K; koff += 4) {
    const int kk = std::min(4, h->K - koff);
    CACK(prof_begin(h, CA_KERNEL_YPASS));
    const ca_ovf_args& o = koff == 0 ? ovf : none;
    if (h->ystore == CA_YSTORE_U8) ypass_t<uint8_t>(h, koff, kk, grid, o);
    else if (h->ystore == CA_YSTORE_U16) ypass_t<uint16_t>(h, koff, kk, grid, o);
    else ypass_t<float>(h, koff, kk, grid, o);
    HIPCK(h, hipGetLastError());
    CACK(prof_end(h));
  }
  // YTpart is [nrb][Gp*K]: column sums over the row blocks (+ the overflow list's chunk sums per gene); ytpsi is laid
  // out [Gp][K] (first G rows used).  Row side: YW = sum of the strips, and the psi.(YW) partials of the ELBO (the fused
  // loop's cell epilogue leaves both to this).  One launch for both (k_yfinish = k_colsum's blocks + k_yw_dot's).
  if (!h->on_side) {
    const int nb_col = cdiv((int64_t)h->Gp * h->K, 64);
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_yfinish, dim3(nb_col + h->n_yw), dim3(1024), 0, h->stream, h->YTpart, h->red + h->off_y, h->nrg,
                                                  (int64_t)h->Gp * h->K, h->Gp * h->K, h->n_ovf > 0 ? h->ovf_col_chunk_ptr : nullptr,
                                                  h->n_ovf > 0 ? h->ovf_csum : nullptr, h->K, h->G, nb_col, h->YWpart,
                                                  h->nseg + (h->n_ovf > 0 ? 1 : 0), h->F, h->D, h->N, h->YW, h->yw_part));
  } else {
    // on the side stream, beside the forward sweep, the two finishers stay two small launches: the merged one's 1024-thread
    // blocks need sixteen free wave slots at once and took 64 us to get through a GPU the sweep has filled (profiles/r02_v1_timeline.txt)
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)h->Gp * h->K, 64)), dim3(1024), 0, h->stream,
                                                  h->YTpart, h->red + h->off_y, h->nrg, (int64_t)h->Gp * h->K, h->Gp * h->K,
                                                  h->n_ovf > 0 ? h->ovf_col_chunk_ptr : nullptr, h->n_ovf > 0 ? h->ovf_csum : nullptr, h->K, h->G));
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_yw_dot, dim3(h->n_yw), dim3(CA_TB), 0, h->stream, h->YWpart,
                                                  h->nseg + (h->n_ovf > 0 ? 1 : 0), h->F, h->D, h->K, h->N, h->YW, h->yw_part));
  }
  h->ycache_valid = true;
  return CA_OK;
}

// Transformed pass over Y with explicit factor buffers (PCA init): row products Y'.Vp -> YWp, column products Y'^T.Fp -> YTp
template <typename YT, int TF>
void ypass_tf_t(ca_engine* h, const float* Fp, const float* Vp, int q, int koff, int kk, float* YWp, float* YTp, dim3 grid) {
  const YT* Y = (const YT*)h->Y;
  ca_ovf_args no_ovf;
  memset(&no_ovf, 0, sizeof(no_ovf));
#define CA_YPT(KK)                                                                                                    \
  hipLaunchKernelGGL((k_ypass<YT, KK, TF>), grid, dim3(CA_TB), 0, h->stream, Y, Fp, q, Vp, koff, YWp, YTp, h->N, h->G, \
                     h->Gp, h->nseg, h->nrb, h->TR, q, no_ovf, (int)grid.x)
  switch (kk) {
    case 1: CA_YPT(1); break;
    case 2: CA_YPT(2); break;
    case 3: CA_YPT(3); break;
    default: CA_YPT(4); break;
  }
#undef CA_YPT
}
template <int TF>
int ypass_tf(ca_engine* h, const float* Fp, const float* Vp, int q, float* YWp, float* YTp, float* csum) {
  dim3 grid((unsigned)((int64_t)h->nrg * h->nseg));
  for (int koff = 0; koff < q; koff += 4) {
    const int kk = std::min(4, q - koff);
    if (h->ystore == CA_YSTORE_U8) ypass_tf_t<uint8_t, TF>(h, Fp, Vp, q, koff, kk, YWp, YTp, grid);
    else if (h->ystore == CA_YSTORE_U16) ypass_tf_t<uint16_t, TF>(h, Fp, Vp, q, koff, kk, YWp, YTp, grid);
    else ypass_tf_t<float, TF>(h, Fp, Vp, q, koff, kk, YWp, YTp, grid);
    HIPCK(h, hipGetLastError());
  }
  if (h->n_ovf > 0) {
    hipLaunchKernelGGL(k_ovf_rows, dim3(cdiv(h->N, CA_TB)), dim3(CA_TB), 0, h->stream, h->ovf_rowptr, h->ovf_col, h->ovf_val, Vp, q,
                       YWp + (int64_t)h->nseg * h->N * q, h->N, q, TF);
    hipLaunchKernelGGL(k_ovf_chunks, dim3(cdiv(h->n_ovf_chunk, CA_TB / 64)), dim3(CA_TB), 0, h->stream, h->ovf_chunk_start, h->ovf_row2,
                       h->ovf_val2, Fp, q, csum, h->n_ovf_chunk, q, TF);
    hipLaunchKernelGGL(k_ovf_cols, dim3(cdiv(h->Gp, CA_TB)), dim3(CA_TB), 0, h->stream, h->ovf_col_chunk_ptr, csum,
                       YTp + (int64_t)h->nrg * h->Gp * q, h->Gp, h->G, q);
    HIPCK(h, hipGetLastError());
  }
  return CA_OK;
}

// modified Gram-Schmidt (twice) on the columns of Q [G][q] (row-major), double precision
void orthonormalize(std::vector<double>& Q, int G, int q) {
  for (int rep = 0; rep < 2; ++rep)
    for (int k = 0; k < q; ++k) {
      for (int j = 0; j < k; ++j) {
        double d = 0.0;
        for (int g = 0; g < G; ++g) d += Q[(size_t)g * q + k] * Q[(size_t)g * q + j];
        for (int g = 0; g < G; ++g) Q[(size_t)g * q + k] -= d * Q[(size_t)g * q + j];
      }
      double nn = 0.0;
      for (int g = 0; g < G; ++g) nn += Q[(size_t)g * q + k] * Q[(size_t)g * q + k];
      nn = std::sqrt(nn);
      if (nn < 1e-300) nn = 1.0;
      for (int g = 0; g < G; ++g) Q[(size_t)g * q + k] /= nn;
    }
}
// cyclic Jacobi eigen-decomposition of a symmetric q x q matrix; eigenvalues descending, eigenvectors in columns of W
void sym_eig(std::vector<double> T, int q, std::vector<double>& lam, std::vector<double>& W) {
  W.assign((size_t)q * q, 0.0);
  for (int i = 0; i < q; ++i) W[(size_t)i * q + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < q; ++i) for (int j = i + 1; j < q; ++j) off += T[(size_t)i * q + j] * T[(size_t)i * q + j];
    if (off < 1e-30) break;
    for (int p_ = 0; p_ < q; ++p_)
      for (int r = p_ + 1; r < q; ++r) {
        const double apr = T[(size_t)p_ * q + r];
        if (std::fabs(apr) < 1e-300) continue;
        const double th = (T[(size_t)r * q + r] - T[(size_t)p_ * q + p_]) / (2.0 * apr);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < q; ++k) {
          const double a = T[(size_t)k * q + p_], b = T[(size_t)k * q + r];
          T[(size_t)k * q + p_] = c * a - sn * b; T[(size_t)k * q + r] = sn * a + c * b;
        }
        for (int k = 0; k < q; ++k) {
          const double a = T[(size_t)p_ * q + k], b = T[(size_t)r * q + k];
          T[(size_t)p_ * q + k] = c * a - sn * b; T[(size_t)r * q + k] = sn * a + c * b;
        }
        for (int k = 0; k < q; ++k) {
          const double a = W[(size_t)k * q + p_], b = W[(size_t)k * q + r];
          W[(size_t)k * q + p_] = c * a - sn * b; W[(size_t)k * q + r] = sn * a + c * b;
        }
      }
  }
  std::vector<int> idx(q);
  for (int i = 0; i < q; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return T[(size_t)a * q + a] > T[(size_t)b * q + b]; });
  lam.resize(q);
  std::vector<double> W2((size_t)q * q);
  for (int j = 0; j < q; ++j) {
    lam[j] = T[(size_t)idx[j] * q + idx[j]];
    for (int k = 0; k < q; ++k) W2[(size_t)k * q + j] = W[(size_t)k * q + idx[j]];
  }
  W = W2;
}

// work that rides in the peer-to-peer all-reduce's launch instead of getting launches of its own in front of it (ca_p2p_args)
struct ca_ar_ride {
  const float* gpart = nullptr; int nslice = 0; int64_t fold_lo = 0, fold_n = 0;
  const double* yw_part = nullptr; int n_yw = 0; int64_t yw_index = -1;
};
inline bool p2p_ride_ok(const ca_engine* h, int64_t n) { return h->p2p && h->p2p->connected && h->p2p_ride && n <= h->p2p->cap; }
int allreduce(ca_engine* h, double* buf, int64_t n, const ca_ar_ride* ride = nullptr) {
  if (h->opt.world <= 1 && !h->comm && !h->host_ar && !(h->p2p && h->p2p->connected)) return CA_OK;   // a 1-rank communicator still reduces (tests)
  if (h->p2p && h->p2p->connected) {
    ca_p2p* pp = h->p2p;
    for (int64_t o = 0; o < n; o += pp->cap) {   // (one launch for everything the loop reduces; longer vectors go in pieces)
      const int64_t m = std::min<int64_t>(pp->cap, n - o);
      const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(m, CA_TB), 64));
      ca_p2p_args a;
      memset(&a, 0, sizeof(a));
      a.peers = pp->peers_dev; a.rank = h->opt.rank; a.world = h->opt.world; a.cap = pp->cap;
      a.seq = ++pp->seq; a.err = pp->err_dev; a.err_local = pp->err_local; a.timeout_ticks = pp->timeout_ticks;
      a.yw_index = -1;
      if (ride) {   // (only ever with n <= cap: one piece, p2p_ride_ok)
        a.gpart = ride->gpart; a.nslice = ride->nslice; a.fold_lo = ride->fold_lo; a.fold_n = ride->fold_n;
        a.yw_part = ride->yw_part; a.n_yw = ride->n_yw; a.yw_index = ride->yw_index;
      }
      LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_p2p_allreduce, dim3(nblk), dim3(CA_TB), 0, h->stream, buf + o, m, a));
    }
    return CA_OK;
  }
  if (h->host_ar) {
    if (n > h->host_ar_cap) {
      if (h->host_ar_buf) HIPCK(h, hipHostFree(h->host_ar_buf));
      HIPCK(h, hipHostMalloc((void**)&h->host_ar_buf, (size_t)n * sizeof(double)));
      h->host_ar_cap = n;
    }
    HIPCK(h, hipMemcpyAsync(h->host_ar_buf, buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    if (h->host_ar(h->host_ar_user, h->host_ar_buf, n) != 0) { h->err = "host all-reduce callback failed"; return CA_ERR_COMM; }
    HIPCK(h, hipMemcpyAsync(buf, h->host_ar_buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return CA_OK;
  }
  if (!h->comm) { h->err = "world > 1 but neither ca_comm_init() nor ca_set_host_allreduce() was called"; return CA_ERR_STATE; }
  int rc = g_rccl.AllReduce(buf, buf, (size_t)n, kNcclFloat64, kNcclSum, h->comm, h->stream);
  if (rc != 0) {
    h->err = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    return CA_ERR_COMM;
  }
  return CA_OK;
}

// the per-gene count totals are sums over ALL cells: reduced once, when the transport is set
int setup_global_sums(ca_engine* h) {
  if (h->sums_global) return CA_OK;
  // a transport that died between the two reductions leaves colsum reduced and YtX not: no second transport may reduce colsum again
  if (h->sums_started) { h->err = "an earlier transport failed inside the setup reductions; this engine cannot take another one -- destroy it"; return CA_ERR_STATE; }
  h->sums_started = true;
  CACK(allreduce(h, h->colsum, h->G));
  if (h->P > 0 && h->K > 0) CACK(allreduce(h, h->YtX, (int64_t)h->G * h->P));
  SYNC(h);
  if (!h->mu_part.empty()) {
    // loc0 = NULL on a shard (ABI 6): mu_guess_g = mean over ALL cells of y_ng / rowMeans(Y)_n (R/inference-tflow.R:220-235) -- the ranks' partial sums and
    // cell counts are added here, then loc0 = safe_inverse_softplus(mu_guess) (:262, :6-11) exactly as create_impl does it for one handle
    const int G = h->G;
    std::vector<double> pack(h->mu_part);
    pack.push_back((double)h->N);
    HIPCK(h, hipMemcpyAsync(h->red, pack.data(), pack.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CACK(allreduce(h, h->red, (int64_t)pack.size()));
    HIPCK(h, hipMemcpyAsync(pack.data(), h->red, pack.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    std::vector<float> l0((size_t)G);
    for (int g = 0; g < G; ++g) {
      const double mu = pack[(size_t)g] / pack[(size_t)G];
      l0[g] = (float)(std::log(1.0 - std::exp(-std::fabs(mu))) + std::max(mu, 0.0));
    }
    CACK(upload_f(h, h->loc, l0));
    HIPCK(h, hipMemcpyAsync(h->loc_init, h->loc, (size_t)G * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIPCK(h, hipMemsetAsync(h->red, 0, pack.size() * sizeof(double), h->stream));
    CACK(refresh_derived(h));
    SYNC(h);
    h->mu_part.clear();
  }
  h->sums_global = true;
  return CA_OK;
}

// Backward half of a train pass.  Needs: coef / dgl from the cell epilogue, mu of the same eps, red[0..3+C) cell sums.
// cell_sums_global: red[0..3+C) was already all-reduced by the monitor pass that produced it (fused path).

// arguments of the O(K + C) body (ca_final_small_body): monitor form (apply = 0, ELBO out) or train form
ca_small_args small_args(ca_engine* h, const double* gene_part, int apply, float lr_t, double* elbo_dst, bool reduce_cells) {
  ca_small_args a;
  a.enabled = 1;
  a.red = h->red; a.gene_part = gene_part; a.ngblk = h->ngblk;
  a.vchi = h->vchi; a.alpha_u = h->alpha_u; a.m_v = h->m_v; a.v_v = h->v_v; a.m_a = h->m_a; a.v_a = h->v_a; a.g_v = h->g_v; a.g_a = h->g_a;
  a.elbo_out = elbo_dst; a.terms_out = h->terms_dev;
  a.G = h->G; a.C = h->C; a.K = h->K; a.apply = apply;
  a.lr_t = lr_t; a.b1 = (float)h->opt.beta1; a.b2 = (float)h->opt.beta2; a.aeps = (float)h->opt.adam_eps;
  a.vmm_part = h->vmm_part; a.vmm = h->vmm; a.D = h->D;
  a.dir_const = h->dir_const;
  a.cell_part = reduce_cells ? h->cell_part : nullptr; a.ncblk = h->ncblk;
  a.host_out = nullptr; a.host_flag = nullptr; a.host_seq = 0; a.reduce_only = 0;
  a.yw_part = nullptr; a.n_yw = 0;
  a.ee_part = nullptr; a.n_ee = 0;
  a.vchi_out = nullptr; a.alpha_out = nullptr;
  if (!apply && elbo_dst && h->host_seq_next && h->host_dev) {   // monitor pass inside ca_run: mirror the ELBO to the host
    a.host_out = h->host_dev + 32;
    a.host_flag = reinterpret_cast<unsigned long long*>(h->host_dev + 33);
    a.host_seq = h->host_seq_next;
    h->host_seq_next = 0;
  }
  return a;
}
// a fused monitor pass leaves its ELBO assembly for the next backward sweep; if none is coming, run it now
inline bool is_sharded(const ca_engine* h) { return h->opt.world > 1 || h->comm || h->host_ar || (h->p2p && h->p2p->connected); }
// Reduce a pending monitor tail's cell partials (and psi.(YW) partials) for a sharded run: red[0 .. 3 + C) local sums,
// ready for the all-reduce.  The Y stream (side stream) must have delivered the psi.(YW) partials first.
int mon_tail_local_sums(ca_engine* h) {
  if (!h->mon_tail.enabled || !h->mon_tail.cell_part) return CA_OK;
  CACK(wait_y(h, false));
  ca_small_args r = h->mon_tail;
  r.reduce_only = 1; r.host_out = nullptr;
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_final_small, dim3(1), dim3(CA_TB), 0, h->stream, r));
  h->mon_tail.cell_part = nullptr; h->mon_tail.yw_part = nullptr;
  return CA_OK;
}
// a fused monitor pass leaves its ELBO assembly for the next train pass's per-gene kernel; if none is coming, run it now
int flush_mon_tail(ca_engine* h) {
  if (!h->mon_tail.enabled) return CA_OK;
  CACK(wait_y(h, false));
  if (h->mon_tail.yw_part) CACK(yfin_flush(h));
  if (is_sharded(h) && h->mon_tail.cell_part) {   // local sums, then the (3 + C)-double all-reduce of a monitor pass
    CACK(mon_tail_local_sums(h));
    CACK(allreduce(h, h->red, 3 + h->C));
  }
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_final_small, dim3(1), dim3(CA_TB), 0, h->stream, h->mon_tail));
  h->mon_tail.enabled = 0;
  return CA_OK;
}
inline ca_small_args no_small_args() { ca_small_args a; memset(&a, 0, sizeof(a)); return a; }

#ifndef CA_BWD_TL3_MAXN
#define CA_BWD_TL3_MAXN 18432   // cells up to which the backward sweep takes three gene tiles per wave (see create_impl)
#endif
// backward half of a train pass: the sweep, the column sums of its partials and the cross-shard reduction.  Changes no
// variable, so ca_run may issue it before it knows whether the loop goes on (train_bwd_speculative).
int train_bwd(ca_engine* h, const float* mu32, bool cell_sums_global) {
  const int W_ = h->S + h->D;
  bool merged = false, ride_ar = false;
  ca_ar_ride ride;
  if (h->poly_fresh) {   // the backward moments of this look-ahead half are in the workspace: per-gene sums straight into red (no slabs, no column sums)
    h->poly_fresh = false;
    h->fold_now = false;
    CACK(yfin_flush(h));
    // a pending monitor pass's tail (reduction of the cell partials into red -- which this pass's alpha step reads --, psi.(YW) sum, ELBO assembly) rides as
    // an extra block of the per-gene launch, as it does on the matrix-core way back; sharded it is a collective of its own
    ca_small_args bwd_tail = no_small_args();
    if (is_sharded(h)) CACK(flush_mon_tail(h));
    else if (h->mon_tail.enabled) { CACK(wait_y(h, false)); bwd_tail = h->mon_tail; h->mon_tail.enabled = 0; }
    CACK(prof_begin(h, CA_KERNEL_BWD));
    const hipError_t e = ca_poly_backward(h->stream, &h->pws, h->V, mu32, h->Lb, h->G, h->C, h->red + h->off_g, &bwd_tail);
    HIPCK(h, e);
    CACK(prof_end(h));
    h->poly_df = true;
  } else
  if (h->bwd_mfma) {
    h->poly_df = false;
    const int xb = cdiv(h->nwt, CA_TB / 64);
    // A pending monitor pass's tail rides on the sweep as one extra block (its fp64 chains hide under 130 us of sweep):
    // whole when unsharded; sharded, only the local sums of the cell / psi.(YW) partials -- ONE all-reduce per iteration
    // then carries them together with this pass's gene sums and the ELBO is assembled after it (k_final_gene's extra
    // block, or ca_run's flush).  Both need the Y stream's psi.(YW) partials: the side stream is awaited here, one
    // kernel later than the cell epilogue that used to need it.
    // The riding count-matrix stream's finishing sums are further extra blocks of this launch.  The pending monitor tail needs one of
    // them (the psi.(YW) partials), so it moves on to the per-gene kernel's extra block (train_update) when they ride here.
    ca_yfin_args yfin;
    memset(&yfin, 0, sizeof(yfin));
    // Sharded over the peer-to-peer transport (round 4): the stream's finisher rides here as it does unsharded, the psi.(YW) partial sum a
    // pending monitor pass needs is added to its cell sum INSIDE the all-reduce's launch, and so are the column sums of this sweep's
    // slabs -- fwd, bwd, all-reduce, update: four launches where there were six.  Other transports keep the launches.
    ride_ar = is_sharded(h) && p2p_ride_ok(h, h->red_n);
    if (is_sharded(h) && !ride_ar) CACK(yfin_flush(h));   // (the tail's local sums go into this pass's all-reduce, so they stay here)
    if (h->yfin_pending) { yfin = yfin_args(h); h->yfin_pending = false; }
    ca_small_args bwd_tail = no_small_args();
    bool split_tail = false;
    if (h->mon_tail.enabled && h->mon_tail.cell_part) {
      CACK(wait_y(h, false));   // the psi.(YW) partials; the column products may still be streaming beside this sweep
      bwd_tail = h->mon_tail;
      merged = is_sharded(h);
      if (merged) { bwd_tail.reduce_only = 1; bwd_tail.host_out = nullptr; }
      if (merged && ride_ar && yfin.nrow > 0 && bwd_tail.yw_part) {   // the partials are made by this very launch: their sum rides in the all-reduce
        ride.yw_part = bwd_tail.yw_part; ride.n_yw = bwd_tail.n_yw; ride.yw_index = 0;
        bwd_tail.yw_part = nullptr;
      }
      // the psi.(YW) partials are being made by this very launch: the cell partials are reduced here, the rest of the tail
      // (their sum, the assembly) follows on the per-gene kernel's extra block
      split_tail = !merged && yfin.nrow > 0 && bwd_tail.yw_part != nullptr;
      if (split_tail) { bwd_tail.reduce_only = 1; bwd_tail.host_out = nullptr; bwd_tail.yw_part = nullptr; }
    }
    const int nextra = (bwd_tail.enabled || yfin.nrow) ? 1 + cdiv(yfin.ncol, CA_TB / 64) + yfin.nrow : 0;
    const int yextra = cdiv(nextra, xb);   // extra block rows behind the sweep's
    ca_yfin_args no_yfin;
    memset(&no_yfin, 0, sizeof(no_yfin));
    // mc_samples = 2 (round 4): both samples in ONE sweep (k_bwd_mfma<.., S2>: one exp per (cell, gene) for the two of them)
    const bool s2b = h->s2f && h->S == 2 && !h->c16;
#define CA_BWDM(DDV) do { if (h->c16) CA_BWDM_(CA_BWD_TL, DDV, false, true, false); else if (s2b) { if (h->bwd_frac) CA_BWDM_(CA_BWD_TL, DDV, true, false, true); else CA_BWDM_(CA_BWD_TL, DDV, false, false, true); } \
                          else if (h->bwd_frac) CA_BWDM_(CA_BWD_TL, DDV, true, false, false); else if (h->bwd_tl == 3) CA_BWDM_(3, DDV, false, false, false); else CA_BWDM_(CA_BWD_TL, DDV, false, false, false); } while (0)
#define CA_BWDM_(TL, DDV, FRV, C16V, S2V)                                                                                   \
  LAUNCH(h, CA_KERNEL_BWD,                                                                                                 \
         hipLaunchKernelGGL((k_bwd_mfma<TL, DDV, FRV, C16V, S2V>), dim3(xb, h->csplit_m + (s == 0 ? yextra : 0)), dim3(CA_TB), \
                            (size_t)h->cchunk_m * 4 * DDV * sizeof(float) * (S2V ? 2 : 1), h->stream,                       \
                            h->coefq + (int64_t)s * h->N16 * 32, h->F, h->etamax2, h->Lb, mu32 + (int64_t)s * h->G, h->Vs,  \
                            h->V, h->gpart, h->dFpart, h->N, h->G, h->cchunk_m, h->S, s, 1, s == 0 ? 1 : 0,                 \
                            s == 0 ? bwd_tail : no_small_args(), h->csplit_m, s == 0 ? yfin : no_yfin,                      \
                            S2V ? h->coefq + h->N16 * 32 : nullptr, S2V ? mu32 + h->G : nullptr))
    for (int s = 0; s < (s2b ? 1 : h->S); ++s) {
      if (h->D == 1) CA_BWDM(1);
      else CA_BWDM(2);
    }
#undef CA_BWDM
#undef CA_BWDM_
    if (bwd_tail.enabled) {
      if (merged) { h->mon_tail.cell_part = nullptr; h->mon_tail.yw_part = nullptr; }   // local sums done; assembly still pending
      else if (split_tail) h->mon_tail.cell_part = nullptr;
      else h->mon_tail.enabled = 0;
    }
    // (summing the sweep's partials inside k_final_gene instead -- one thread per gene, csplit_m loads in a row -- was
    //  slower than this parallel launch at 100k cells: 2219 -> 2190 it/s; small unsharded problems fold it: fold_gsum)
    h->fold_now = h->fold_gsum && !is_sharded(h);
    if (ride_ar) { ride.gpart = h->gpart; ride.nslice = h->csplit_m; ride.fold_n = (int64_t)h->G * W_; }
    else if (!h->fold_now)
      LAUNCH(h, CA_KERNEL_OTHER,
             hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)h->G * W_, 64)), dim3(1024), 0, h->stream, h->gpart,
                                h->red + h->off_g, h->csplit_m, (int64_t)h->G * W_, h->G * W_));
  } else {
    h->fold_now = false;
    h->poly_df = false;
    CACK(yfin_flush(h));
    CACK(flush_mon_tail(h));
    for (int s = 0; s < h->S; ++s)
      for (int ch = 0; ch < h->nchunk; ++ch) {
        BwdArgs a;
        a.coef = h->coef + ((int64_t)s * h->nchunk + ch) * h->N * CA_CW;
        a.F = h->F; a.em = h->etamax2;
        a.Lb = h->Lb + (int64_t)ch * h->G * CA_CW;
        a.mu = mu32 + (int64_t)s * h->G;
        a.Vs = h->Vs; a.V = h->V; a.gpart = h->gpart; a.dFpart = h->dFpart;
        a.N = h->N; a.G = h->G; a.cchunk = h->cchunk; a.D = h->D; a.S = h->S; a.sidx = s;
        a.first_s = (ch == 0); a.first = (s == 0 && ch == 0);
        const int nc = std::min(CA_CW, h->C - ch * CA_CW);
        LAUNCH(h, CA_KERNEL_BWD, launch_bwd(h->RG, nc, dim3(cdiv(h->ntile, CA_TB / 64), h->csplit), h->stream, a));
      }
    LAUNCH(h, CA_KERNEL_OTHER,
           hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)h->G * W_, 64)), dim3(1024), 0, h->stream, h->gpart,
                              h->red + h->off_g, h->csplit, (int64_t)h->G * W_, h->G * W_));
  }
  // the Y stream's results (Y^T psi in red_y, the psi.(YW) partials) are first needed from here on
  CACK(wait_y(h, true));
  // sharded loop: a pending monitor pass's cell sums travel with this pass's gene sums -- ONE all-reduce per iteration;
  // the ELBO is assembled after it (k_final_gene's extra block, or ca_run's flush)
  if (cell_sums_global && !merged) { ride.fold_lo = 0; CACK(allreduce(h, h->red + h->off_g, h->red_n - h->off_g, ride_ar ? &ride : nullptr)); }
  else { ride.fold_lo = h->off_g; CACK(allreduce(h, h->red, h->red_n, ride_ar ? &ride : nullptr)); }
  if (is_sharded(h)) h->ycache_valid = false;   // red_y now holds the GLOBAL sum
  return CA_OK;
}

// update half: per-gene / per-cell gradients from the reduced sums, Adam when `apply`
// will the update half of a train pass take the one-launch form (k_update_merged)?  (the loop has announced the next eps pair, the fused forward
// sweep that follows can make the exponent bound itself, K >= 1)
inline bool update_merges(const ca_engine* h, int apply, const double* elbo_dst) {
  const int64_t mB = h->s2 ? 2 * h->hint_A + 1 : h->hint_B;
  return apply && h->upd_merge && !elbo_dst && h->pre_ok && h->hint_A >= 0 && mB >= 0 && h->fused_ok && h->gene_part_alt && h->fwd_cell && h->K > 0;
}
int train_update(ca_engine* h, const float* eps, int apply, double* elbo_dst) {
  const int N256 = cdiv(h->N, CA_TB);
  float lr_t = 0.f;
  if (apply) {
    // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), float32 like TF's _prepare()/_apply_dense
    lr_t = (float)h->opt.learning_rate * sqrtf(1.f - h->b2p) / (1.f - h->b1p);
  }
  // A pending monitor pass's ELBO is assembled by one extra block of the per-gene kernel (ca_final_small_body: the
  // reduction of the cell / psi.(YW) partials unless train_bwd did it for the all-reduce, then the assembly) BEFORE the
  // O(K + C) update, which rides on the per-cell kernel the same way.
  CACK(yfin_flush(h));   // (nothing pending when a backward sweep preceded)
  ca_small_args mon = no_small_args();
  if (h->mon_tail.enabled) { mon = h->mon_tail; h->mon_tail.enabled = 0; }
  // psi's own step: extra blocks of the per-gene kernel (it needs the sweep's dF partials and YW, nothing per-gene)
  ca_psi_args psi;
  memset(&psi, 0, sizeof(psi));
  if (h->K > 0) {
    psi.nblk = N256; psi.F = h->F; psi.YW = h->YW; psi.dFpart = h->dFpart; psi.m_psi = h->m_psi; psi.v_psi = h->v_psi; psi.g_psi = h->g_psi;
    psi.N = h->N; psi.D = h->D; psi.K = h->K; psi.ntile = h->poly_df ? 1 : h->bwd_mfma ? cdiv(h->nwt, CA_TB / 64) : h->ntile;
  }
  h->pre_valid = false;
  // Round 4: the whole update half in ONE launch (k_update_merged) whenever the loop has announced the next eps pair, the fused forward
  // sweep that follows can make the exponent bound itself (fwd_cell) and K >= 1 -- see the kernel.  Everything else (call-by-call API,
  // the last step of a run, the VALU / two-kernel forward paths) keeps the two launches below.
  {
    int64_t mA = h->hint_A, mB = h->hint_B;
    if (h->s2) { mA = 2 * h->hint_A; mB = mA + 1; }
    if (update_merges(h, apply, elbo_dst)) {
      ca_merge_args mg;
      memset(&mg, 0, sizeof(mg));
      h->gate_armed = false;
      if (h->gate_req && h->host_dev) {   // ca_run: wait on the device for the host's decision (every block but the monitor block)
        // (the gate word and the error word each in a cache line of their own: doubles 32 / 33 are the ELBO and its flag, which the host spins on)
        // (pinned doubles: 32 / 33 the ELBO and its flag, which the host spins on; 40 the fatal word; 48 the host's answer; 56 the relay's verdict)
        mg.gate = reinterpret_cast<const unsigned long long*>(h->host_dev + 48); mg.gate_seq = ++h->gate_seq;
        mg.gate_ack = reinterpret_cast<unsigned long long*>(h->host_dev + 56);
        mg.gate_err = reinterpret_cast<unsigned long long*>(h->host_dev + 40);
        mg.gate_local = h->gate_local;
        mg.gate_timeout = h->gate_ticks;   // the RELAY's patience (s_memrealtime ticks, 100 MHz): ~1 ms, then the launch stores nothing and the host re-queues it
        h->gate_armed = true;
        h->gate_t0 = std::chrono::steady_clock::now();   // (before the launch below: the relay's clock starts no earlier)
      }
      h->gate_req = false;
      ca_pre_args& pre = mg.pre;
      pre.nblk = h->ngblk;
      pre.loc = h->loc; pre.ls = h->ls; pre.epsA = h->eps_dev + mA * (int64_t)h->G; pre.epsB = h->eps_dev + mB * (int64_t)h->G;
      pre.colsum = h->colsum; pre.Lb = h->Lb; pre.V = h->V; pre.YtX = h->YtX; pre.muA = h->mu32; pre.muB = h->s2 ? h->mu32 + h->G : h->mu32B; pre.Mb = h->Mb2;
      pre.s2 = h->s2 ? 1 : 0;
      pre.gene_partA = h->gene_part_alt; pre.gene_partB = h->gene_partB_alt; pre.Mq = h->fwd_mfma ? h->Mq : nullptr;
      pre.G = h->G; pre.D = h->D; pre.K = h->K; pre.mrow = h->frow; pre.C = h->C;
      if (h->y_ys && !h->ys_quant_ready) {   // the int8 stream's images of the stepped W and psi (exponents from the lagged maxima, as on k_adam_cell)
        bool lagged = false;
        const ca_ysq_args a = ys_quant_args(h, h->ys_steps >= 0 ? h->ys_steps + 1 : -1, &lagged);
        if (lagged) {
          mg.ysq = a; mg.ysq.nblk = h->ngblk + N256;
          h->ys_namax[(h->ys_slot + 1) % 3] = h->ngblk + N256;   // one pair per gene block, then one per psi block
          h->ys_quant_ready = true;
        }
      }
      mg.tail = small_args(h, h->gene_part, 1, lr_t, nullptr, false);
      mg.tail.terms_out = nullptr;                 // (the pending monitor pass's block of this launch owns the ELBO terms)
      mg.tail.vmm_part = nullptr; mg.tail.vmm = nullptr;   // the range of V' is the next sweep's business
      mg.tail.vchi_out = h->vchi_alt; mg.tail.alpha_out = h->alpha_u_alt;
      mg.glogit = h->glogit; mg.dgl = h->dgl; mg.m_gl = h->m_gl; mg.v_gl = h->v_gl; mg.C = h->C; mg.ncell = N256;
      if (!h->vmm_at_ready) {   // first merged update of this engine: both range buffers at (+inf, -inf); from here on every launch resets the next one's
        int init[32];
        for (int i = 0; i < 32; ++i) init[i] = (i & 8) ? (int)0x807FFFFF /* ca_f2ord(-inf) */ : (int)0x7F800000 /* ca_f2ord(+inf) */;
        HIPCK(h, hipMemcpyAsync(h->vmm_at, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));   // (the source is on this stack frame)
        h->vmm_at_ready = true;
      }
      mg.vmm_at = h->vmm_at + 16 * h->vmm_at_idx; mg.vmm_at_next = h->vmm_at + 16 * (1 - h->vmm_at_idx);
      // the sweep-independent part of the per-gene gradient: left by the prologue that made this pass's eps draw (if it was a merged
      // update's, with the parameters still the ones it saw), and made for the next train pass's draw by this launch's prologue
      mg.aux_ld = h->G;
      const int64_t this_draw = (eps - h->eps_dev) / (int64_t)h->G;
      mg.aux_in = (h->S == 1 && h->gaux_slot >= 0 && h->gaux_slot == this_draw) ? h->gaux + (int64_t)h->gaux_idx * 5 * h->G : nullptr;
      mg.aux_out = h->S == 1 ? h->gaux + (int64_t)(1 - h->gaux_idx) * 5 * h->G : nullptr;
      LAUNCH(h, CA_KERNEL_OTHER,
             hipLaunchKernelGGL(k_update_merged, dim3(h->ngblk + (mon.enabled ? 1 : 0) + 1 + cdiv(psi.nblk, 4) + cdiv(N256, 4)), dim3(CA_UM_TB), 0, h->stream,
                                h->red + h->off_g, h->red + h->off_y, eps, h->colsum, h->YtX, h->vchi, h->loc, h->ls, h->V, h->m_loc, h->v_loc,
                                h->m_ls, h->v_ls, h->m_V, h->v_V, h->g_loc, h->g_ls, h->g_V, h->Vs, h->vmm_part, h->G, h->S, h->D, h->K, lr_t,
                                (float)h->opt.beta1, (float)h->opt.beta2, (float)h->opt.adam_eps, mon, h->ngblk, psi,
                                h->fold_now ? h->gpart : nullptr, h->csplit_m, mg));
      h->fold_now = false;
      if (h->async_y && !h->ride_ok && !h->ride_ys) {   // side-stream Y pass (2- / 4-byte storage, K != 1): psi is final from here, as below
        HIPCK(h, hipEventRecord(h->ev_params, h->stream));
        h->y_defer = true;
      }
      std::swap(h->vchi, h->vchi_alt);
      std::swap(h->alpha_u, h->alpha_u_alt);
      h->em_stale = true;
      h->vmm_at_idx = 1 - h->vmm_at_idx;   // (the buffer this launch filled is the other one from now on: vmm_at_cur below)
      h->gaux_idx = 1 - h->gaux_idx; h->gaux_slot = h->S == 1 ? mB : -1;
      h->pre_valid = true; h->pre_A = mA; h->pre_B = mB;
      h->hint_A = h->hint_B = -1;
      if (h->ys_steps >= 0) h->ys_steps += 1;
      h->b1p *= (float)h->opt.beta1;
      h->b2p *= (float)h->opt.beta2;
      h->adam_steps += 1;
      h->ycache_valid = false;
      h->yfin_pending = false;
      h->look_valid = false;
      if (h->poly && h->y_ys) h->poly_y_defer = true;   // series form: the next series pass sends the count-matrix products of the stepped state to the side stream
      return CA_OK;
    }
  }
  if (apply) h->gaux_slot = -1;   // the per-gene parameters change without a prologue that leaves the next step's sweep-independent part
  LAUNCH(h, CA_KERNEL_OTHER,
         hipLaunchKernelGGL(k_final_gene, dim3(h->ngblk + (mon.enabled ? 1 : 0) + psi.nblk), dim3(CA_TB), 0, h->stream, h->red + h->off_g,
                            h->red + h->off_y, eps, h->colsum, h->YtX, h->vchi, h->loc, h->ls, h->V, h->m_loc, h->v_loc, h->m_ls,
                            h->v_ls, h->m_V, h->v_V, h->g_loc, h->g_ls, h->g_V, h->Vs, h->vmm_part, h->G, h->S, h->D, h->K, apply,
                            lr_t, (float)h->opt.beta1, (float)h->opt.beta2, (float)h->opt.adam_eps, mon, h->ngblk, psi,
                            h->fold_now ? h->gpart : nullptr, h->csplit_m));
  h->fold_now = false;
  if (apply && h->async_y && h->K > 0 && !h->ride_ok && !h->ride_ys) {
    // psi is final: the Y pass for the new parameters goes to the side stream from HERE (its launches are issued by the
    // next pass, so the main stream is not left waiting for the host to get through them), and the per-cell kernel below
    // (q(z) logits, exponent bound, the O(K + C) update and the next pass's per-gene prologue: 12-16 us) is its head
    // start over the next forward sweep.  The Y stream needs one: letting the sweep and the Y kernel start together
    // cost 18 % (2219 -> 1825 it/s; the sweep's blocks take the CUs first).
    HIPCK(h, hipEventRecord(h->ev_params, h->stream));
    h->y_defer = true;
  }
  // the next fused pass's per-gene prologue, when the loop has announced its eps slots: extra blocks of the per-cell kernel
  ca_pre_args pre;
  memset(&pre, 0, sizeof(pre));
  // (the hints are PASS slots; two samples per pass for mc_samples = 2)
  int64_t hA = h->hint_A, hB = h->hint_B;
  if (h->s2) { hA = 2 * h->hint_A; hB = hA + 1; }
  if (apply && h->pre_ok && h->hint_A >= 0 && hB >= 0 && h->fused_ok && h->gene_part_alt) {
    pre.nblk = h->ngblk;
    pre.loc = h->loc; pre.ls = h->ls; pre.epsA = h->eps_dev + hA * (int64_t)h->G; pre.epsB = h->eps_dev + hB * (int64_t)h->G;
    pre.colsum = h->colsum; pre.Lb = h->Lb; pre.V = h->V; pre.YtX = h->YtX; pre.muA = h->mu32; pre.muB = h->s2 ? h->mu32 + h->G : h->mu32B; pre.Mb = h->Mb2;
    pre.s2 = h->s2 ? 1 : 0;
    pre.gene_partA = h->gene_part_alt; pre.gene_partB = h->gene_partB_alt; pre.Mq = h->fwd_mfma ? h->Mq : nullptr;
    pre.G = h->G; pre.D = h->D; pre.K = h->K; pre.mrow = h->frow; pre.C = h->C;
  }
  // the int8 count-matrix stream's quantiser for the state this step produces: extra blocks of the same launch (W and psi are
  // final since k_final_gene), exponents bounded from the maxima of the state before (`ys_steps` steps ago, + this one)
  ca_ysq_args ysq;
  memset(&ysq, 0, sizeof(ysq));
  if (apply && h->y_ys && h->K > 0 && !h->ys_quant_ready) {
    bool lagged = false;
    const ca_ysq_args a = ys_quant_args(h, h->ys_steps >= 0 ? h->ys_steps + 1 : -1, &lagged);
    if (lagged) { ysq = a; h->ys_quant_ready = true; }
  }
  LAUNCH(h, CA_KERNEL_OTHER,
         hipLaunchKernelGGL(k_adam_cell, dim3(N256 + 1 + pre.nblk + ysq.nblk), dim3(CA_TB), 0, h->stream, h->F, h->glogit, h->dgl, h->m_gl, h->v_gl,
                            h->N, h->C, h->D, apply, lr_t, (float)h->opt.beta1, (float)h->opt.beta2, (float)h->opt.adam_eps,
                            h->vmm_part, h->ngblk, h->etamax2, small_args(h, h->gene_part, apply ? 1 : 0, lr_t, elbo_dst, false),
                            N256, pre, ysq));
  if (pre.nblk) { h->pre_valid = true; h->pre_A = hA; h->pre_B = hB; }
  h->hint_A = h->hint_B = -1;
  if (apply) {
    if (h->ys_steps >= 0) h->ys_steps += 1;
    h->b1p *= (float)h->opt.beta1;
    h->b2p *= (float)h->opt.beta2;
    h->adam_steps += 1;
    h->ycache_valid = false;   // V', its range and etamax2 were refreshed inside the step's own kernels
    h->yfin_pending = false;
    h->look_valid = false;
    if (h->poly && h->y_ys && h->K > 0) h->poly_y_defer = true;   // (as in the one-launch form above)
  }
  return CA_OK;
}

int train_tail(ca_engine* h, const float* eps, const float* mu32, int apply, double* elbo_dst, bool cell_sums_global) {
  CACK(train_bwd(h, mu32, cell_sums_global));
  return train_update(h, eps, apply, elbo_dst);
}

// One evaluation of the model for the eps of device slot `eps_slot`.
//   mode CA_MODE_ELBO : forward only, ELBO -> elbo_dst        (`sess$run(elbo)`)
//   mode CA_MODE_GINIT: forward only, overwrite the q(z) logits (`gamma_init`)
//   mode CA_MODE_TRAIN: forward + backward (+ Adam when apply)  (`sess$run(train)`)
int run_pass(ca_engine* h, int64_t eps_slot, int mode, int apply, double* elbo_dst) {
  const float* eps = h->eps_dev + eps_slot * (int64_t)h->S * h->G;
  h->poly_fresh = false;   // (a plain pass makes its own forward half: backward moments left by an unused look-ahead are nobody's)
  h->poly_y_defer = false;   // (... and runs its count-matrix pass in line, as ever: the deferral was for a series pass)
  CACK(flush_mon_tail(h));
  CACK(ensure_etamax(h));
  if (mode != CA_MODE_TRAIN) h->hint_A = h->hint_B = -1;
  h->pre_valid = false;   // this pass rewrites mu32 and the current per-gene partials
  LAUNCH(h, CA_KERNEL_OTHER,
         hipLaunchKernelGGL(k_gene_pre, dim3(h->ngblk), dim3(CA_TB), 0, h->stream, h->loc, h->ls, eps, h->colsum, h->Lb, h->V,
                            h->D, h->K, h->YtX, h->mu32, h->Mb, h->gene_part, h->G, h->S, h->nchunk, CA_CW, 0, CA_CW));
  CACK(ensure_ycache(h));
  for (int s = 0; s < h->S; ++s)
    for (int ch = 0; ch < h->nchunk; ++ch) {
      const int nc = std::min(CA_CW, h->C - ch * CA_CW);
      const float* M = h->Mb + ((int64_t)s * h->nchunk + ch) * h->G * CA_CW;
      // Zpart [S][gsplit][nchunk][N][8]: the kernel strides its split index by N*8, so pass per-(s,ch) bases
      // laid out as [s][ch][split][N][8]
      float* Zp = h->Zpart + (((int64_t)s * h->nchunk + ch) * h->gsplit) * h->N * CA_CW;
      LAUNCH(h, CA_KERNEL_FWD, launch_fwd(nc, h->D, dim3(cdiv(h->N, CA_TB * kFwdR), h->gsplit), h->stream, h->F, h->etamax2, h->Vs, M, Zp, h->N, h->G, h->gchunk));
    }
  CACK(wait_y(h, true));   // the cell epilogue is the first consumer of YW / Y^T psi
  // (matrix-core products arrive as finished row sums: one "strip", already in YW)
  const bool yw_done = h->y_mfma || h->y_ys;
  const float* ywp = yw_done ? h->YW : h->YWpart;
  const int ywseg = yw_done ? 1 : h->nseg + (h->n_ovf > 0 ? 1 : 0);
  if (h->C <= 64) {
    int CP = 1;
    while (CP < h->C) CP <<= 1;
    dim3 grid(h->ncblk);
#define CA_CELL(CPV)                                                                                                          \
  LAUNCH(h, CA_KERNEL_CELL,                                                                                                   \
         hipLaunchKernelGGL((k_cell_par<CPV>), grid, dim3(CA_TB), 0, h->stream, h->Zpart, h->A, h->cn, h->s64, h->etamax2,    \
                            h->glogit, h->alpha_u, h->F, ywp, h->YW, h->coef, h->dgl, h->cell_part, h->N, h->C, h->S,       \
                            h->D, h->K, h->gsplit, h->nchunk, ywseg, mode, h->bwd_mfma ? h->coefq : nullptr, h->N16))
    switch (CP) {
      case 1: CA_CELL(1); break;
      case 2: CA_CELL(2); break;
      case 4: CA_CELL(4); break;
      case 8: CA_CELL(8); break;
      case 16: CA_CELL(16); break;
      case 32: CA_CELL(32); break;
      default: CA_CELL(64); break;
    }
#undef CA_CELL
  } else {
    LAUNCH(h, CA_KERNEL_CELL,
           hipLaunchKernelGGL(k_cell, dim3(h->ncblk), dim3(CA_TB), 0, h->stream, h->Zpart, h->A, h->cn, h->s64, h->etamax2, h->glogit,
                              h->alpha_u, h->F, ywp, h->YW, h->coef, h->dgl, h->scratch, h->cell_part, h->N, h->C, h->S, h->D,
                              h->K, h->gsplit, h->nchunk, ywseg, mode));
  }
  if (mode == CA_MODE_GINIT) return CA_OK;
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_reduce_part, dim3(3 + h->C), dim3(CA_TB), 0, h->stream, h->cell_part, h->red, h->ncblk, 3 + h->C));
  if (mode == CA_MODE_TRAIN) return train_tail(h, eps, h->mu32, apply, elbo_dst, false);
  CACK(allreduce(h, h->red, 3 + h->C));
  LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_final_small, dim3(1), dim3(CA_TB), 0, h->stream,
                                                small_args(h, h->gene_part, 0, 0.f, elbo_dst, false)));
  return CA_OK;
}

// Series form or sweeps for the pass about to be queued?  See ca_engine::poly_ring.  *use_series = the decision; *mirror / *seq = where and under which
// number this pass leaves its own ranges (the series form's first launch does; a pass that takes the sweeps queues the two tiny range launches itself,
// poly_ranges_for_sweeps).
#define CA_POLY_LAG 4
int poly_wait_entry(ca_engine* h, uint64_t want, double out[3]) {
  volatile double* e = h->poly_ring + (want % 16) * 4;
  unsigned spins = 0;
  auto t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(20);
  while (e[0] != (double)want) {
    if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() >= t_next) {
      t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
      const hipError_t q = hipStreamQuery(h->stream);
      if (q != hipSuccess && q != hipErrorNotReady) HIPCK(h, q);
      if ((q == hipSuccess && e[0] != (double)want) || std::chrono::steady_clock::now() >= t_end) {
        h->err = "internal: the ranges of an earlier pass never reached the host (series form's look ahead)";
        return CA_ERR_STATE;
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  out[0] = e[1]; out[1] = e[2]; out[2] = e[3];
  return CA_OK;
}
int poly_guard(ca_engine* h, bool* use_series, double** mirror, double* seq, bool* ranges_queued) {
  const uint64_t s = h->poly_seq + 1;
  *ranges_queued = false;
  *mirror = h->poly_ring_dev + (s % 16) * 4;
  *seq = (double)s;
  double r[3];
  int steps;
  if (s < h->poly_seq_base + CA_POLY_LAG) {
    // the first passes after the ranges became unknown (a new engine, a restart, a set_param): this state's own ranges, now -- the one wait for the device
    // the look ahead ever makes; the passes behind it count their steps from the base entry
    if (s == h->poly_seq_base) {
      HIPCK(h, ca_poly_ranges(h->stream, &h->pws, h->V, h->F, h->G, h->N, *mirror, *seq));
      *ranges_queued = true;
      h->poly_steps_at[s % 16] = h->adam_steps;
    }
    CACK(poly_wait_entry(h, h->poly_seq_base, r));
    steps = (int)(h->adam_steps - h->poly_steps_at[h->poly_seq_base % 16]);
  } else {
    CACK(poly_wait_entry(h, s - CA_POLY_LAG, r));
    steps = (int)(h->adam_steps - h->poly_steps_at[(s - CA_POLY_LAG) % 16]);
  }
  const double lr = h->opt.learning_rate;
  const double bound = std::max(lr, lr * (1.0 - h->opt.beta1) / std::sqrt(1.0 - h->opt.beta2)) * 1.0001;   // no Adam step moves a variable further (TF1 form, lr_t <= lr / sqrt(1 - beta2) ...)
  *use_series = ca_poly_covers(r[0], r[1], r[2], steps, bound);
  h->poly_seq = s;
  h->poly_steps_at[s % 16] = h->adam_steps;
  return CA_OK;
}

// Monitor pass for eps slot A fused with the forward half of the NEXT train pass (eps slot B): one sweep,
// one exp per (cell, gene) for both (same parameters, R/inference-tflow.R:401,403 of consecutive iterations).
int fused_pass(ca_engine* h, int64_t slotA, int64_t slotB, double* elbo_dst, double* elbo_dstB = nullptr, int64_t trainA = -1) {
  // trainA >= 0 (mc_samples = 2, h->s2f): slotA / slotB are the monitor pass's two samples, trainA / trainA + 1 the next train pass's -- four draws, one sweep
  const bool s2f = trainA >= 0;
  const float* epsA = h->eps_dev + slotA * (int64_t)h->G;
  const float* epsB = h->eps_dev + slotB * (int64_t)h->G;
  if (elbo_dstB) CACK(flush_mon_tail(h));   // pair sweeps come from outside the loop: nothing may be left pending
  if (h->pre_valid && h->pre_A == slotA && h->pre_B == slotB) {   // the train pass before this one already ran the prologue
    std::swap(h->gene_part, h->gene_part_alt);
    std::swap(h->gene_partB, h->gene_partB_alt);
  } else {
    LAUNCH(h, CA_KERNEL_OTHER,
           hipLaunchKernelGGL(k_gene_pre_fused, dim3(h->ngblk), dim3(CA_TB), 0, h->stream, h->loc, h->ls, epsA, epsB, h->colsum, h->Lb,
                              h->V, h->D, h->K, h->YtX, h->mu32, h->s2 ? h->mu32 + h->G : h->mu32B, h->Mb2, h->gene_part, h->gene_partB, h->G,
                              h->frow, h->C, h->fwd_mfma ? h->Mq : nullptr, h->s2 ? 1 : 0));
  }
  if (s2f) {   // the train pair's prologue: second operand image, mu of both samples where the backward sweeps read it (the monitor pair's mu is nobody's)
    const float* epsTA = h->eps_dev + trainA * (int64_t)h->G;
    LAUNCH(h, CA_KERNEL_OTHER,
           hipLaunchKernelGGL(k_gene_pre_fused, dim3(h->ngblk), dim3(CA_TB), 0, h->stream, h->loc, h->ls, epsTA, epsTA + h->G, h->colsum, h->Lb,
                              h->V, h->D, h->K, h->YtX, h->mu32, h->mu32 + h->G, h->Mb2, h->gene_partB, h->gene_partB, h->G,
                              h->frow, h->C, h->Mq + (int64_t)h->nk32 * 1024, 1));
  }
  h->pre_valid = false;
  // the Y products of this parameter state: riding on the sweep's own launch (below), or from the side stream / in line
  h->poly_fresh = false;   // (backward moments of an earlier, unused look-ahead half are nobody's from here on)
  bool series = h->poly && !s2f && !elbo_dstB && !h->fwd_gate && !is_sharded(h);   // (sharded: the ranks see different cells, they could decide differently)
  double* pl_mirror = nullptr; double pl_seq = 0.0;
  if (series) {
    bool queued = false;
    CACK(poly_guard(h, &series, &pl_mirror, &pl_seq, &queued));
    if (!series && !queued) HIPCK(h, ca_poly_ranges(h->stream, &h->pws, h->V, h->F, h->G, h->N, pl_mirror, pl_seq));   // (the sweeps take this pass: its ranges all the same)
    if (series) h->n_series += 1; else h->n_series_fallback += 1;
  }
  const bool ride = (h->ride_ok || h->ride_ys) && !h->ycache_valid && h->fwd_cell && !h->y_defer && !h->y_pending && !series;   // (the series form has no sweep launch to ride on)
  if (!ride && !series) CACK(ensure_ycache(h));   // (series: placed between its own launches, below)
  ca_cell_ptrs cp;
  cp.A = h->A; cp.cn = h->cn; cp.s64 = h->s64; cp.etamax2 = h->etamax2; cp.glogit = h->glogit; cp.F = h->F;
  cp.coef = h->coef; cp.dgl = h->dgl; cp.coefq = h->bwd_mfma ? h->coefq : nullptr;
  cp.vmm_at = nullptr; cp.etamax_w = nullptr;
  cp.gate = nullptr; cp.gate_go = 0ull;
  if (h->fwd_gate) {
    if (!(h->fwd_cell && ride && h->ride_ys) || !h->gate_local) { h->err = "internal: a forward sweep queued ahead of the host's decision must be the one-launch form"; return CA_ERR_STATE; }
    cp.gate = h->gate_local; cp.gate_go = (h->gate_seq << 1) | 1ull;
  }
  // (the series form needs no exponent bound: after a merged update it stays stale until a pass that wants it asks, ensure_etamax)
  if (series) {}
  else if (h->em_stale && h->fwd_cell && h->D > 0) { cp.vmm_at = h->vmm_at + 16 * (1 - h->vmm_at_idx); cp.etamax_w = h->etamax2; h->em_stale = false; }
  else CACK(ensure_etamax(h));
  cp.ee_partB = (elbo_dstB && h->fwd_cell) ? h->ee_partB : nullptr;
  cp.s2 = h->s2 ? 1 : 0; cp.N16 = h->N16;
  int CP = 1;
  while (CP < h->C) CP <<= 1;
  int cell_blocks = h->ncblk;
  if (series) {
    // Z of both draws from the moments of M over gene bins (ca_poly.hip), the same cell epilogue, d/dF and the backward moments in one pass over
    // the CELLS: no cells x genes sweep.  The count-matrix products of this state run as their own launch (in line).
    // Order: the three small moment launches FIRST, on an empty device (they are chains of memory latencies: beside the stream's blocks they took twice
    // as long, and the cell launch waits for them); THEN the count-matrix stream goes to the side stream (deferred since the update: poly_y_defer) and runs
    // beside the cell launch, which is arithmetic.
    cp.etamax2 = h->poly_zero; cp.coefq = nullptr; cp.vmm_at = nullptr; cp.etamax_w = nullptr;
    CACK(prof_begin(h, CA_KERNEL_FWD));
    hipError_t e = ca_poly_moments(h->stream, &h->pws, h->V, h->F, h->mu32, h->mu32B, h->Lb, h->G, h->N, h->C,
                                   h->host_dev ? reinterpret_cast<unsigned int*>(h->host_dev + 42) : nullptr, pl_mirror, pl_seq);
    HIPCK(h, e);
    CACK(prof_end(h));
    if (h->poly_side && h->poly_y_defer && !h->ycache_valid) {
      h->poly_y_defer = false;
      HIPCK(h, hipEventRecord(h->ev_params, h->stream));
      h->y_defer = true;
    }
    CACK(ensure_ycache(h));
    CACK(prof_begin(h, CA_KERNEL_FWD));
    e = ca_poly_cells(h->stream, &h->pws, h->N, h->C, h->K, &cp, h->alpha_u, h->cell_part, h->dFpart);
    HIPCK(h, e);
    CACK(prof_end(h));
    cell_blocks = h->pws.n_cell_blocks;
    h->poly_fresh = true;
  } else
  if (h->fwd_cell && ride && h->ride_ys) {   // the one-copy int8 matrix-core stream's blocks interleaved with the sweep's
    cell_blocks = h->ncblk_f;
    CACK(ys_quant(h));
    ca_ysride_args ya;
    memset(&ya, 0, sizeof(ya));
    ya.Ys = h->Ys; ya.io = ys_io(h); ya.F = h->F; ya.V = h->V; ya.Df = h->D;
    ya.Gp = h->Gp; ya.RS = h->ys_RS; ya.nb_main = h->ys_nrg * h->ys_nseg; ya.nb_y = ya.nb_main;
    ya.ovf = ys_ovf(h);
    ya.nb_y += ya.ovf.nb_rows + ya.ovf.nb_chunks;
    ya.pat_a = 2; ya.pat_b = 1;
    if (h->opt.ride_pattern > 0 && (h->opt.ride_pattern >> 8) > 0 && (h->opt.ride_pattern & 255) > 0) {
      ya.pat_a = h->opt.ride_pattern >> 8; ya.pat_b = h->opt.ride_pattern & 255;
    }
    if (h->opt.ride_pattern < 0) ya.pers = std::min(-h->opt.ride_pattern, ya.nb_main);
    else if (h->opt.ride_pattern == 0 && ya.nb_main >= 2 * h->n_cu) ya.pers = h->n_cu;
    const dim3 grid(ya.pers > 0 ? (unsigned)(ya.pers + h->ncblk_f + (ya.nb_y - ya.nb_main)) : (unsigned)(h->ncblk_f + ya.nb_y));
#define CA_FCYS(DV, TLBV, DPV, C16V, S2FV)                                                                                            \
  LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_mix_ys<DV, TLBV, 2, DPV, C16V, S2FV>), grid, dim3(CA_TB), 0, h->stream, h->F, h->etamax2, h->Vs, \
                                              h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig, h->ncblk_f, ya))
#define CA_FCYS_D(TLBV, DPV) do { if (h->c16) { if (h->D == 1) CA_FCYS(1, TLBV, DPV, true, false); else CA_FCYS(2, TLBV, DPV, true, false); }  \
                                  else if (s2f) { if (h->D == 1) CA_FCYS(1, TLBV, DPV, false, true); else CA_FCYS(2, TLBV, DPV, false, true); } \
                                  else { if (h->D == 1) CA_FCYS(1, TLBV, DPV, false, false); else CA_FCYS(2, TLBV, DPV, false, false); } } while (0)
    // (one piece in flight per wave: 128 VGPRs = four waves per SIMD like the vector stream's launch; two pieces, 162 VGPRs and
    //  three waves, measured 2824 against 2869 it/s at cfg-3 -- profiles/r03_ab_ystream.txt)
#ifndef CA_YS_RIDE_DEPTH
#define CA_YS_RIDE_DEPTH 1   // (lab: pieces in flight per stream wave)
#endif
    if (h->fwd_bal && !s2f) {   // small problems: one eight-wave sweep block per CU, left-over tiles spread gene-wise (ca_fwdbal.hip.h)
      ca_bal_args ba;
      memset(&ba, 0, sizeof(ba));
      ba.nb = h->n_cu; ba.r = h->bal_r; ba.nchunk = h->bal_nchunk; ba.xw = h->bal_xw;
      ba.extra = h->bal_nchunk == 0 ? h->bal_r : 0;   // left-over tiles exchanged in gene chunks (default), or as single-tile blocks of their own
      cell_blocks = h->n_cu + ba.extra;               // (a row of block partials per block that runs an epilogue)
      if (++h->bal_tag == 0u) h->bal_tag = 1u;
      ba.tag = h->bal_tag;
      ba.timeout_ticks = 50000000ull;   // 0.5 s: every chunk a block waits for was dispatched before it and is made first
      ba.err = reinterpret_cast<unsigned int*>(h->host_dev + 41);
      // stream units per stream block: one (four live waves beside the sweep block's eight) while the stream still ends inside the sweep, else two
      // (measured, 8192 ... 25 000 cells x 5000 genes: one unit per block is 2-4 us per iteration faster at every size, gpurun_out/r5/stair_units.txt;
      //  ride_pattern = 2 asks for two)
      ba.stream_units = h->opt.ride_pattern == 2 ? 2 : 1;
      const dim3 gridb((unsigned)(h->n_cu + ba.extra + (ba.stream_units == 2 ? (ya.nb_main + 1) / 2 : ya.nb_main) + (ya.nb_y - ya.nb_main)));   // sweep blocks, left-over tiles' blocks, stream blocks, the overflow list's
#define CA_FBAL(TLV) LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_bal_ys<1, TLV, CA_YS_RIDE_DEPTH>), gridb, dim3(CA_BAL_TB), 0, h->stream, h->F, h->etamax2, \
                                                                 h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, ba, ya))
      switch (h->bal_q) {
        case 1: CA_FBAL(1); break;
        case 2: CA_FBAL(2); break;
        case 3: CA_FBAL(3); break;
        case 4: CA_FBAL(4); break;
        case 5: CA_FBAL(5); break;
        default: CA_FBAL(6); break;
      }
#undef CA_FBAL
    } else
    if (h->fc_tl == 6) CA_FCYS_D(6, CA_YS_RIDE_DEPTH);
    else if (h->fc_tl == 1 && !h->c16) { if (h->D == 1) CA_FCYS(1, 1, CA_YS_RIDE_DEPTH, false, false); else CA_FCYS(2, 1, CA_YS_RIDE_DEPTH, false, false); }
    else CA_FCYS_D(2, CA_YS_RIDE_DEPTH);
#undef CA_FCYS_D
#undef CA_FCYS
    CACK(ys_finish(h, h->yfin_split && (!is_sharded(h) || p2p_ride_ok(h, h->red_n))));
  } else if (h->fwd_cell && ride) {   // ... and the Y stream's blocks interleaved with the sweep's in the same grid
    cell_blocks = h->ncblk_f;
    ca_yride_args ya;
    memset(&ya, 0, sizeof(ya));
    ya.Y = (const uint8_t*)h->Y; ya.F = h->F; ya.Dstride = h->D; ya.V = h->V; ya.YWpart = h->YWpart; ya.YTpart = h->YTpart;
    ya.G = h->G; ya.Gp = h->Gp; ya.nseg = h->nseg; ya.nrb = h->nrb; ya.TR = h->TR;
    ya.nb_main = h->nrg * h->nseg; ya.nb_y = ya.nb_main;
    if (h->n_ovf > 0) {
      ya.ovf.nb_rows = cdiv(h->N, CA_TB); ya.ovf.nb_chunks = cdiv(h->n_ovf_chunk, CA_TB / 64);
      ya.ovf.rowptr = h->ovf_rowptr; ya.ovf.col = h->ovf_col; ya.ovf.val = h->ovf_val; ya.ovf.YWextra = h->YWpart + (int64_t)h->nseg * h->N * h->K;
      ya.ovf.chunk_start = h->ovf_chunk_start; ya.ovf.row2 = h->ovf_row2; ya.ovf.val2 = h->ovf_val2; ya.ovf.csum = h->ovf_csum; ya.ovf.nchunk = h->n_ovf_chunk;
      ya.nb_y += ya.ovf.nb_rows + ya.ovf.nb_chunks;
    }
    // Interleave of the two kinds in dispatch order.  Blocks go round-robin over the 8 XCDs, so a period that divides 8 (the
    // obvious even / odd split) puts ALL sweep blocks on four XCDs and all stream blocks on the other four; two sweep blocks per
    // stream block mixes them on every CU: cfg-3 2795 -> 3008 it/s, 12.5k cells 10.9k -> 12.1k, cfg-2 16.3k -> 17.9k
    // (profiles/r02_ab_ystream.txt section 8).
    ya.pat_a = 2; ya.pat_b = 1;
    if (h->opt.ride_pattern > 0 && (h->opt.ride_pattern >> 8) > 0 && (h->opt.ride_pattern & 255) > 0) {
      ya.pat_a = h->opt.ride_pattern >> 8; ya.pat_b = h->opt.ride_pattern & 255;
    }
    // ride_seq: no separate stream blocks -- sweep block b also streams unit b (k_fwd_cell_seq_y); units past the sweep's block
    // count and the overflow list's blocks follow as stream-only blocks
    const bool seq = h->ride_seq;
    // Long-lived stream blocks lead the grid (ca_yride_args::pers) when there are at least two units of the matrix per CU: one
    // such block per CU measured best (cfg-3, with the non-temporal stream: 2:1 interleave 3090, 256 blocks 3147, 341 / 512
    // blocks 3008 / 2979, 192 / 128 blocks 2790 / 2320 it/s).  ride_pattern < 0 sets the number, > 0 asks for the interleave.
    if (!seq && h->opt.ride_pattern < 0) ya.pers = std::min(-h->opt.ride_pattern, ya.nb_main);
    else if (!seq && h->opt.ride_pattern == 0 && ya.nb_main >= 2 * h->n_cu) ya.pers = h->n_cu;
    const dim3 grid(seq ? (unsigned)(h->ncblk_f + std::max(0, ya.nb_main - h->ncblk_f) + (ya.nb_y - ya.nb_main))
                        : ya.pers > 0 ? (unsigned)(ya.pers + h->ncblk_f + (ya.nb_y - ya.nb_main)) : (unsigned)(h->ncblk_f + ya.nb_y));
#define CA_FCY(DV, TLBV)                                                                                                              \
  do {                                                                                                                                \
    if (seq)                                                                                                                          \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_seq_y<DV, TLBV, 2>), grid, dim3(CA_TB), 0, h->stream, h->F, h->etamax2, h->Vs, h->Mq, \
                                                  cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig, h->ncblk_f, ya)); \
    else                                                                                                                              \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_mix_y<DV, TLBV, 2>), grid, dim3(CA_TB), 0, h->stream, h->F, h->etamax2, h->Vs, h->Mq, \
                                                  cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig, h->ncblk_f, ya)); \
  } while (0)
    if (h->fc_tl == 8) { if (h->D == 1) CA_FCY(1, 8); else CA_FCY(2, 8); }
    else if (h->fc_tl == 6) { if (h->D == 1) CA_FCY(1, 6); else CA_FCY(2, 6); }
    else { if (h->D == 1) CA_FCY(1, 2); else CA_FCY(2, 2); }
#undef CA_FCY
    {   // the stream's finishers, in line behind the launch they rode on (one launch: column sums + row sums / psi.(YW) partials)
      const int nb_col = cdiv((int64_t)h->Gp * h->K, 64);
      LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_yfinish, dim3(nb_col + h->n_yw), dim3(1024), 0, h->stream, h->YTpart, h->red + h->off_y, h->nrg,
                                                    (int64_t)h->Gp * h->K, h->Gp * h->K, h->n_ovf > 0 ? h->ovf_col_chunk_ptr : nullptr,
                                                    h->n_ovf > 0 ? h->ovf_csum : nullptr, h->K, h->G, nb_col, h->YWpart,
                                                    h->nseg + (h->n_ovf > 0 ? 1 : 0), h->F, h->D, h->N, h->YW, h->yw_part));
    }
    h->ycache_valid = true;
  } else if (h->fwd_cell) {   // sweep + cell epilogue in one kernel: no Z partials, one launch
    cell_blocks = h->ncblk_f;
#define CA_FC(DV, TLV)                                                                                                       \
  LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell<DV, TLV>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, h->etamax2, \
                                              h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32))
#define CA_FCD(TLV) do { if (h->D == 1) CA_FC(1, TLV); else CA_FC(2, TLV); } while (0)
#define CA_FCM(DV, TLV)                                                                                                      \
  LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_mix<DV, TLV, 2>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, \
                                              h->etamax2, h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig))
#define CA_FCMD(TLV) do { if (h->D == 1) CA_FCM(1, TLV); else CA_FCM(2, TLV); } while (0)
    if (h->c16) {   // 9..16 clones: the two default block shapes
#define CA_FC16(DV) do { if (h->fc_nbig > 0) \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_mix<DV, 6, 2, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, \
                                                  h->etamax2, h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig)); \
    else if (h->fc_tl == 6) \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell<DV, 6, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, h->etamax2, \
                                                  h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32)); \
    else \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell<DV, 2, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, h->etamax2, \
                                                  h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32)); } while (0)
      if (h->D == 1) CA_FC16(1); else CA_FC16(2);
#undef CA_FC16
    } else if (s2f) {   // mc_samples = 2, four draws: the same two block shapes
#define CA_FCS2(DV) do { if (h->fc_nbig > 0) \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell_mix<DV, 6, 2, false, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, \
                                                  h->etamax2, h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32, h->fc_nbig)); \
    else if (h->fc_tl == 6) \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell<DV, 6, false, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, h->etamax2, \
                                                  h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32)); \
    else \
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_cell<DV, 2, false, true>), dim3(h->ncblk_f), dim3(CA_TB), 0, h->stream, h->F, h->etamax2, \
                                                  h->Vs, h->Mq, cp, h->alpha_u, h->cell_part, h->N, h->C, h->K, h->nk32)); } while (0)
      if (h->D == 1) CA_FCS2(1); else CA_FCS2(2);
#undef CA_FCS2
    } else if (h->fc_nbig > 0) {
      switch (h->fc_tl) {
        case 4: CA_FCMD(4); break;
        case 5: CA_FCMD(5); break;
        case 8: CA_FCMD(8); break;
        default: CA_FCMD(6); break;
      }
    } else
    switch (h->fc_tl) {
      case 1: CA_FCD(1); break;
      case 2: CA_FCD(2); break;
      case 4: CA_FCD(4); break;
      case 5: CA_FCD(5); break;
      case 6: CA_FCD(6); break;
      default: CA_FCD(8); break;
    }
#undef CA_FCMD
#undef CA_FCM
#undef CA_FCD
#undef CA_FC
  } else {
  if (h->fwd_mfma) {
    const dim3 grid(cdiv(h->N, (CA_TB / 64) * CA_FM_TL * 16), h->fsplit);
    if (h->D == 1)
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_mfma<1>), grid, dim3(CA_TB), 0, h->stream, h->F, h->etamax2, h->Vs, h->Mq,
                                                  h->Zpart2, h->N, h->G, h->fkchunk, h->nk32));
    else
      LAUNCH(h, CA_KERNEL_FWD, hipLaunchKernelGGL((k_fwd_mfma<2>), grid, dim3(CA_TB), 0, h->stream, h->F, h->etamax2, h->Vs, h->Mq,
                                                  h->Zpart2, h->N, h->G, h->fkchunk, h->nk32));
  } else {
    LAUNCH(h, CA_KERNEL_FWD, launch_fwd_fused(h->C, h->D, dim3(cdiv(h->N, CA_TB * kFwdR), h->gsplit), h->stream, h->F, h->etamax2,
                                              h->Vs, h->Mb2, h->Zpart2, h->N, h->G, h->gchunk));
  }
  {
    // (no wait for the side stream here: this epilogue does not touch the Y stream's products -- k_yw_dot does)
    dim3 grid(h->ncblk);
#define CA_CELLF(CPV)                                                                                                   \
  LAUNCH(h, CA_KERNEL_CELL,                                                                                             \
         hipLaunchKernelGGL((k_cell_fused<CPV>), grid, dim3(CA_TB), 0, h->stream, h->Zpart2, h->frow, cp, h->alpha_u,    \
                            h->cell_part, h->N, h->C, h->D, h->K, h->fwd_mfma ? h->fsplit : h->gsplit))
    switch (CP) {
      case 1: CA_CELLF(1); break;
      case 2: CA_CELLF(2); break;
      case 4: CA_CELLF(4); break;
      default: CA_CELLF(8); break;
    }
#undef CA_CELLF
  }
  }
  // The ELBO assembly (reduction of the cell partials and of k_yw_dot's psi.(YW) partials, then the O(K + C) body) is
  // left pending: it rides on the per-gene kernel of the train pass that completes this look-ahead (train_update), after
  // the single all-reduce of that pass when sharded (train_bwd); ca_run and the odd ends flush it (flush_mon_tail).
  h->mon_tail = small_args(h, h->gene_part, 0, 0.f, elbo_dst, true);
  h->mon_tail.ncblk = cell_blocks;
  if (h->K > 0) { h->mon_tail.yw_part = h->yw_part; h->mon_tail.n_yw = h->n_yw; }
  if (!h->tail_fuse) CACK(flush_mon_tail(h));
  if (cp.ee_partB) {
    // pair sweep (ca_final_elbo): both draws are monitor passes of the same parameters.  ELBO A as usual, then ELBO B from
    // the same red[1 ..] (prior, entropy and sum-gamma terms do not depend on the draw), its own cell sum of the expected
    // log-likelihood and its own per-gene partials.
    CACK(flush_mon_tail(h));
    ca_small_args b = small_args(h, h->gene_partB, 0, 0.f, elbo_dstB, false);
    b.ee_part = h->ee_partB; b.n_ee = cell_blocks;
    if (h->K > 0) { b.yw_part = h->yw_part; b.n_yw = h->n_yw; }
    LAUNCH(h, CA_KERNEL_OTHER, hipLaunchKernelGGL(k_final_small, dim3(1), dim3(CA_TB), 0, h->stream, b));
    h->look_valid = false;   // no train pass follows: the second draw's coef / d logits are by-products
    return CA_OK;
  }
  h->look_valid = true;
  h->look_slot = s2f ? trainA + 1 : slotB;
  h->bwd_ready = false;
  return CA_OK;
}

// Train pass whose forward half was already done by fused_pass(.., slotB): backward sweep + Adam.
int train_from_lookahead(ca_engine* h, int64_t slot) {
  // (s2: `slot` is the PASS, its two samples are draws 2 slot and 2 slot + 1 and the look-ahead bookkeeping is in draws)
  const int64_t want = h->s2 ? 2 * slot + 1 : slot;
  if (!h->look_valid || h->look_slot != want) { h->err = "internal: no look-ahead forward for this eps slot"; return CA_ERR_STATE; }
  h->look_valid = false;
  const bool have_bwd = h->bwd_ready && h->bwd_slot == want;
  h->bwd_ready = false;
  if (!have_bwd) CACK(train_bwd(h, h->s2 ? h->mu32 : h->mu32B, true));
  return train_update(h, h->eps_dev + slot * (int64_t)h->S * h->G, 1, nullptr);
}

// ca_run: issue the backward half of the NEXT train pass (its forward half came with the monitor pass just queued)
// before the host reads that monitor pass's ELBO, so the read-back and the stop test run beside the sweep instead of
// draining the GPU every iteration.  If the loop stops, the sweep's scratch results are simply never used.
int train_bwd_speculative(ca_engine* h) {
  if (!h->look_valid) return CA_OK;
  CACK(train_bwd(h, h->s2 ? h->mu32 : h->mu32B, true));
  h->bwd_ready = true;
  h->bwd_slot = h->look_slot;
  return CA_OK;
}
int read_doubles(ca_engine* h, const double* dev, double* out, int n);
// ca_run: the monitor pass's O(K + C) body mirrors its ELBO into pinned host memory and raises a sequence flag
// (ca_small_args::host_*); the host spins on the flag instead of draining a stream, so whatever was queued behind
// the monitor pass (the speculative backward sweep) keeps the GPU busy.  Falls back to a plain read-back when the
// stream runs dry without the flag (mirror unavailable) and surfaces stream errors.
int wait_host_elbo(ca_engine* h, unsigned long long seq, const double* dev, double* out) {
  if (!h->host_dev) return read_doubles(h, dev, out, 1);
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->host_pinned + 33);
  // (the stream is asked only when the flag is overdue: a hipStreamQuery behind freshly queued work makes the runtime put a
  //  marker packet into the queue, and the update half's first launch then started 5.8 us late in EVERY iteration of ca_run --
  //  the "unexplained gap" of round 1; host API trace in profiles/r02_ab_ystream.txt section 7)
  unsigned spins = 0;
  auto t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
  while (*flag != seq) {
    if ((++spins & 0x3FFu) == 0 && std::chrono::steady_clock::now() >= t_next) {
      t_next = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        if (*flag == seq) break;
        return read_doubles(h, dev, out, 1);
      }
      if (q != hipErrorNotReady) HIPCK(h, q);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  *out = *reinterpret_cast<volatile double*>(h->host_pinned + 32);
  return comm_check(h);
}

// monitor pass on eps slot m; with next >= 0 (and the fused path available) also the forward half of the train
// pass on slot `next`, which train_pass() then completes
int monitor_pass(ca_engine* h, int64_t m, int64_t next, double* elbo_dst) {
  if (h->fused_ok && h->s2) {   // mc_samples = 2: the pass's two samples in the two column halves
    // ... and the next train pass's two samples in a second operand set of the same sweep (round 4), or the sweep alone
    if (h->s2f && next >= 0 && next != m) return fused_pass(h, 2 * m, 2 * m + 1, elbo_dst, nullptr, 2 * next);
    CACK(fused_pass(h, 2 * m, 2 * m + 1, elbo_dst));
    h->look_valid = false;
    return CA_OK;
  }
  if (h->fused_ok && next >= 0) return fused_pass(h, m, next, elbo_dst);
  if (h->fused_ok && h->c16) {   // (9..16 clones have no plain matrix-core pass: the fused sweep with this draw in both roles)
    CACK(fused_pass(h, m, m, elbo_dst));
    h->look_valid = false;
    return CA_OK;
  }
  return run_pass(h, m, CA_MODE_ELBO, 0, elbo_dst);
}
int train_pass(ca_engine* h, int64_t slot) {
  if (h->look_valid && h->look_slot == (h->s2 ? 2 * slot + 1 : slot)) return train_from_lookahead(h, slot);
  if (h->fused_ok && (h->c16 || h->s2) && (h->bwd_mfma || !is_sharded(h))) {   // its forward half: the same sweep with this pass's draw(s)
    if (h->c16) CACK(flush_mon_tail(h));   // (9..16 clones: a pending tail is a real monitor pass's)
    h->mon_tail.enabled = 0;   // (mc_samples = 2: a tail still pending belongs to a ca_iterate pass whose ELBO nobody reads; ca_run has flushed its own)
    CACK(h->s2 ? fused_pass(h, 2 * slot, 2 * slot + 1, h->terms_dev + 3) : fused_pass(h, slot, slot, h->terms_dev + 3));   // (the monitor role's ELBO is scratch)
    return train_from_lookahead(h, slot);
  }
  return run_pass(h, slot, CA_MODE_TRAIN, 1, nullptr);
}

int ensure_eps_cap(ca_engine* h, int64_t draws) {
  if (draws <= h->eps_cap) return CA_OK;
  float* p = nullptr;
  HIPCK(h, hipMalloc((void**)&p, (size_t)draws * h->S * h->G * sizeof(float)));
  if (h->eps_dev) {
    SYNC(h);
    HIPCK(h, hipFree(h->eps_dev));
    h->dev_bytes -= h->eps_cap * (int64_t)h->S * h->G * 4;
  }
  h->eps_dev = p;
  h->eps_cap = draws;
  h->dev_bytes += draws * (int64_t)h->S * h->G * 4;
  return CA_OK;
}
int ensure_elbo_cap(ca_engine* h, int64_t n) {
  if (n <= h->elbo_cap) return CA_OK;
  double* p = nullptr;
  HIPCK(h, hipMalloc((void**)&p, (size_t)n * sizeof(double)));
  if (h->elbo_dev) {
    SYNC(h);
    HIPCK(h, hipFree(h->elbo_dev));
  }
  h->elbo_dev = p;
  h->elbo_cap = n;
  return CA_OK;
}

// put `n_draws` draws on the device: from the caller's stream, or generated (built-in Philox stream)
int stage_eps(ca_engine* h, const float* eps_stream, int64_t have, int64_t need) {
  const int64_t per = (int64_t)h->S * h->G;
  h->look_valid = false;   // the staged eps slots are about to change
  h->gaux_slot = -1;
  CACK(ensure_eps_cap(h, std::max<int64_t>(need, 1)));
  if (eps_stream) {
    if (have < need) {
      h->err = "eps stream too short: need " + std::to_string(need) + " draws, got " + std::to_string(have);
      return CA_ERR_INVALID;
    }
    // Through the engine's pinned staging buffer, stream-ordered, WITHOUT draining the stream: a copy from the caller's pageable
    // memory followed by a synchronisation left the GPU idle for ~250 us at the start of every call (kernel trace of the
    // driver's 20-step command: 4 % of its time).  The buffer is reused only after the copy that last read it has completed.
    const size_t bytes = (size_t)need * per * sizeof(float);
    if (bytes > h->eps_stage_bytes) {
      if (h->ev_stage) HIPCK(h, hipEventSynchronize(h->ev_stage));
      if (h->eps_stage) HIPCK(h, hipHostFree(h->eps_stage));
      h->eps_stage = nullptr; h->eps_stage_bytes = 0;
      HIPCK(h, hipHostMalloc((void**)&h->eps_stage, bytes));
      h->eps_stage_bytes = bytes;
    }
    if (!h->ev_stage) HIPCK(h, hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
    else HIPCK(h, hipEventSynchronize(h->ev_stage));
    memcpy(h->eps_stage, eps_stream, bytes);
    HIPCK(h, hipMemcpyAsync(h->eps_dev, h->eps_stage, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipEventRecord(h->ev_stage, h->stream));
  } else {
    // counter-based stream: draws are independent, so a long run's worth (2 + 2 max_iter draws) is generated by several
    // host threads -- same values whatever the thread count -- straight into a pinned staging buffer the engine keeps
    // (pageable memory cost 1.5 ms of copy for 8 MB; with 16 threads the 402 draws of a default fit took 3.3 ms in all)
    const size_t bytes = (size_t)need * per * sizeof(float);
    if (h->ev_stage) HIPCK(h, hipEventSynchronize(h->ev_stage));   // the copy that last read the buffer is done BEFORE it is freed or refilled
    if (bytes > h->eps_stage_bytes) {
      if (h->eps_stage) HIPCK(h, hipHostFree(h->eps_stage));
      h->eps_stage = nullptr; h->eps_stage_bytes = 0;
      HIPCK(h, hipHostMalloc((void**)&h->eps_stage, bytes));
      h->eps_stage_bytes = bytes;
    }
    float* out = h->eps_stage;
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(need / 4, 48), (int64_t)std::thread::hardware_concurrency() / 2));
    if (nt > 1 && need * per >= (1 << 16)) {
      std::vector<std::thread> pool;
      const uint64_t seed = h->opt.seed, base = h->draw;
      for (int64_t t = 0; t < nt; ++t)
        pool.emplace_back([=]() { for (int64_t d = t; d < need; d += nt) ca_philox::normal_draw(seed, base + d, per, out + d * per); });
      for (auto& th : pool) th.join();
    } else {
      for (int64_t d = 0; d < need; ++d) ca_philox::normal_draw(h->opt.seed, h->draw + d, per, out + d * per);
    }
    h->draw += need;
    HIPCK(h, hipMemcpyAsync(h->eps_dev, out, bytes, hipMemcpyHostToDevice, h->stream));
    if (!h->ev_stage) HIPCK(h, hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
    HIPCK(h, hipEventRecord(h->ev_stage, h->stream));
  }
  return CA_OK;
}

int read_doubles(ca_engine* h, const double* dev, double* out, int n) {
  HIPCK(h, hipMemcpyAsync(h->host_pinned, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  for (int i = 0; i < n; ++i) out[i] = h->host_pinned[i];
  return CA_OK;
}

template <typename ST>
int scan_and_convert(ca_engine* h, const ST* src_dev, int64_t sn, int64_t sg) {
  double* maxv = nullptr;
  int* flags = nullptr;
  HIPCK(h, hipMalloc((void**)&maxv, 32));
  flags = (int*)(maxv + 1);
  unsigned long long* cnt = (unsigned long long*)(maxv + 2);
  HIPCK(h, hipMemsetAsync(maxv, 0, 32, h->stream));
  const int64_t total = h->N * (int64_t)h->G;
  hipLaunchKernelGGL((k_scan_y<ST>), dim3(std::min<int64_t>(4096, cdiv(total, CA_TB))), dim3(CA_TB), 0, h->stream, src_dev, total, maxv, flags, cnt);
  HIPCK(h, hipGetLastError());
  double hm[4];
  HIPCK(h, hipMemcpyAsync(hm, maxv, 32, hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  const double mx = hm[0];
  int fl;
  memcpy(&fl, &hm[1], sizeof(int));
  unsigned long long n255;
  memcpy(&n255, &hm[2], sizeof(n255));
  if (fl & 2) { hipFree(maxv); h->err = "count matrix has negative or NaN entries"; return CA_ERR_INVALID; }
  int store = h->opt.y_storage;
  const bool integral = !(fl & 1);
  // u8 also serves matrices with a few entries above 255 (at most 1 in 64, each < 2^24): those keep 255 in
  // the dense byte and their excess in a sorted overflow list
  const bool u8_ok = integral && mx < 16777216.0 && (int64_t)n255 * 64 <= total;
  if (store == CA_YSTORE_AUTO) {
    if (!integral) store = CA_YSTORE_F32;
    else if (u8_ok) store = CA_YSTORE_U8;
    else if (mx <= 65535.0) store = CA_YSTORE_U16;
    else store = CA_YSTORE_F32;
  }
  if ((store == CA_YSTORE_U8 && !(integral && mx < 16777216.0)) || (store == CA_YSTORE_U16 && (!integral || mx > 65535.0))) {
    hipFree(maxv);
    h->err = "requested y_storage cannot hold the counts (max " + std::to_string(mx) + ")";
    return CA_ERR_INVALID;
  }
  h->ystore = store;
  h->ybytes = store == CA_YSTORE_U8 ? 1 : store == CA_YSTORE_U16 ? 2 : 4;
  h->VEC = 16 / h->ybytes;
  const int segw = 64 * h->VEC;
  h->nseg = cdiv(h->G, segw);
  h->Gp = h->nseg * segw;
  h->y_dev_bytes = h->N * (int64_t)h->Gp * h->ybytes;
  // Kernels that take one thread per 16 BYTES of the resident matrix (k_bias_y, the tilers) use one-dimensional grids: 2^32 work-items = 64 GiB of matrix,
  // more than a quarter of this device's memory in ONE matrix (there are two copies).  Refuse instead of wrapping.
  if ((h->N * (int64_t)h->Gp) / 16 >= ((int64_t)1 << 32) || h->N >= ((int64_t)1 << 31)) {
    h->err = "count matrix too large for this build (cells x padded genes >= 2^36 bytes, or 2^31 cells)";
    return CA_ERR_INVALID;
  }
  uint8_t* yb = nullptr;
  CACK(dalloc(h, &yb, h->y_dev_bytes));
  h->Y = yb;
  HIPCK(h, hipMemsetAsync(maxv, 0, 32, h->stream));
  const int64_t tot = h->N * (int64_t)h->Gp;
  const dim3 grid = ca_grid_flat(tot, CA_TB);
  if (store == CA_YSTORE_U8 && n255 > 0) {
    h->n_ovf = (int64_t)n255;
    int *orow = nullptr, *ocol = nullptr; float* oval = nullptr;
    HIPCK(h, hipMalloc((void**)&orow, n255 * sizeof(int)));
    HIPCK(h, hipMalloc((void**)&ocol, n255 * sizeof(int)));
    HIPCK(h, hipMalloc((void**)&oval, n255 * sizeof(float)));
    hipLaunchKernelGGL((k_convert_y_u8ovf<ST>), grid, dim3(CA_TB), 0, h->stream, src_dev, (uint8_t*)h->Y, h->N, h->G, h->Gp, sn, sg, cnt, orow, ocol, oval);
    HIPCK(h, hipGetLastError());
    h->h_orow.resize(n255); h->h_ocol.resize(n255); h->h_oval.resize(n255);
    HIPCK(h, hipMemcpyAsync(h->h_orow.data(), orow, n255 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->h_ocol.data(), ocol, n255 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipMemcpyAsync(h->h_oval.data(), oval, n255 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    hipFree(orow); hipFree(ocol); hipFree(oval); hipFree(maxv);
    // fixed order: sort by (cell, gene) for the CSR copy and by (gene, cell) for the CSC copy
    const int64_t nz = (int64_t)n255;
    std::vector<int64_t> idx(nz);
    for (int64_t i = 0; i < nz; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
      return h->h_orow[a] != h->h_orow[b] ? h->h_orow[a] < h->h_orow[b] : h->h_ocol[a] < h->h_ocol[b]; });
    std::vector<int> r1(nz), c1(nz); std::vector<float> v1(nz);
    for (int64_t i = 0; i < nz; ++i) { r1[i] = h->h_orow[idx[i]]; c1[i] = h->h_ocol[idx[i]]; v1[i] = h->h_oval[idx[i]]; }
    h->h_orow = r1; h->h_ocol = c1; h->h_oval = v1;
    std::vector<int64_t> rowptr(h->N + 1, 0), colptr(h->G + 1, 0);
    for (int64_t i = 0; i < nz; ++i) { rowptr[r1[i] + 1]++; colptr[c1[i] + 1]++; }
    for (int64_t i = 0; i < h->N; ++i) rowptr[i + 1] += rowptr[i];
    for (int g = 0; g < h->G; ++g) colptr[g + 1] += colptr[g];
    std::vector<int> r2(nz); std::vector<float> v2(nz);
    {
      std::vector<int64_t> fill(colptr.begin(), colptr.end() - 1);
      for (int64_t i = 0; i < nz; ++i) { const int64_t q = fill[c1[i]]++; r2[q] = r1[i]; v2[q] = v1[i]; }   // rows ascending within a gene
    }
    // chunks of <= 256 entries inside each gene's CSC range
    std::vector<int64_t> chunk_start; std::vector<int> col_chunk_ptr(h->G + 1, 0);
    for (int g = 0; g < h->G; ++g) {
      col_chunk_ptr[g] = (int)chunk_start.size();
      for (int64_t e = colptr[g]; e < colptr[g + 1]; e += 256) chunk_start.push_back(e);
    }
    col_chunk_ptr[h->G] = (int)chunk_start.size();
    h->n_ovf_chunk = (int)chunk_start.size();
    chunk_start.push_back(nz);
    CACK(dalloc(h, &h->ovf_rowptr, h->N + 1));
    CACK(dalloc(h, &h->ovf_chunk_start, (int64_t)chunk_start.size()));
    CACK(dalloc(h, &h->ovf_col_chunk_ptr, h->G + 1));
    CACK(dalloc(h, &h->ovf_csum, (int64_t)h->n_ovf_chunk * std::max(h->K, 1)));
    HIPCK(h, hipMemcpyAsync(h->ovf_chunk_start, chunk_start.data(), chunk_start.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->ovf_col_chunk_ptr, col_chunk_ptr.data(), col_chunk_ptr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    CACK(dalloc(h, &h->ovf_col, nz)); CACK(dalloc(h, &h->ovf_row2, nz));
    CACK(dalloc(h, &h->ovf_val, nz)); CACK(dalloc(h, &h->ovf_val2, nz));
    HIPCK(h, hipMemcpyAsync(h->ovf_rowptr, rowptr.data(), rowptr.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->ovf_col, c1.data(), nz * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->ovf_row2, r2.data(), nz * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->ovf_val, v1.data(), nz * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(h->ovf_val2, v2.data(), nz * sizeof(float), hipMemcpyHostToDevice, h->stream));
    SYNC(h);   // (the sources are this function's vectors; and every copy above is ordered on the engine's stream, behind the buffers' zeroing)
    return CA_OK;
  }
  if (store == CA_YSTORE_U8) hipLaunchKernelGGL((k_convert_y<ST, uint8_t>), grid, dim3(CA_TB), 0, h->stream, src_dev, (uint8_t*)h->Y, h->N, h->G, h->Gp, sn, sg, flags);
  else if (store == CA_YSTORE_U16) hipLaunchKernelGGL((k_convert_y<ST, uint16_t>), grid, dim3(CA_TB), 0, h->stream, src_dev, (uint16_t*)h->Y, h->N, h->G, h->Gp, sn, sg, flags);
  else hipLaunchKernelGGL((k_convert_y<ST, float>), grid, dim3(CA_TB), 0, h->stream, src_dev, (float*)h->Y, h->N, h->G, h->Gp, sn, sg, flags);
  HIPCK(h, hipGetLastError());
  HIPCK(h, hipMemcpyAsync(hm, maxv, 16, hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  memcpy(&fl, &hm[1], sizeof(int));
  hipFree(maxv);
  if (fl & 1) { h->err = "counts are not exactly representable in the on-device storage type"; return CA_ERR_INVALID; }
  return CA_OK;
}

template <typename ST>
int gather_y(ca_engine* h, const void* src, void** dst, int64_t sn, int64_t sg, const int64_t* ci_dev, const int32_t* gi_dev) {
  const int64_t total = h->N * (int64_t)h->G;
  HIPCK(h, hipMalloc(dst, (size_t)total * sizeof(ST)));
  hipLaunchKernelGGL((k_gather_y<ST>), ca_grid_flat(total, CA_TB), dim3(CA_TB), 0, h->stream, (const ST*)src, (ST*)*dst, h->N, h->G, sn, sg,
                     ci_dev, gi_dev);
  HIPCK(h, hipGetLastError());
  SYNC(h);
  return CA_OK;
}

// ---- host -> device ingestion (round 5; profiles/r05_ingest.txt) --------------------------------------------------------------
// The caller's matrix arrives in pageable host memory: from R an N x G column-major DOUBLE matrix (R/clonealign.R:212-222), 4 GB at
// 100k x 5k.  Until round 4 it went up as ONE pageable hipMemcpy into an N*G*8-byte device staging buffer (measured: 70 ms = the PCIe
// time of 4 GB, the whole fit that follows takes 60 ms).  Doubles carry nothing the engine can store -- its widest storage is float32
// and a count that is not exactly a float is an error either way -- so the bytes are narrowed on the HOST, on their way into the pinned
// buffers a copy has to pass through anyway: a few worker threads convert chunk c + 1 (float64 -> float32, every value checked for
// exactness) into one pinned buffer while the DMA engine moves chunk c out of the other.  Half the PCIe bytes, half the device
// staging, and the device sees exactly the values it saw before (the scan / conversion kernels and their results are unchanged).
// Other source types have nothing to narrow and keep the runtime's own pageable copy, which runs at 98 % of the pinned rate on these boxes
// (56.5 against 57.6 GB/s; the same bytes through this pipeline's memcpy threads measured SLOWER, 44 GB/s: profiles/r05_ingest.txt).
struct ingest_result { bool inexact = false; };
static int ingest_host_matrix(hipStream_t stream, std::string& err, const void* src, int64_t total, int src_dtype, void* dst_dev, ingest_result* res,
                              int64_t seg_len = 0, int64_t src_ld = 0) {
  // seg_len / src_ld (ABI 6, ca_problem.y_ld): the source is `total / seg_len` runs of seg_len contiguous elements, src_ld elements apart (a block of
  // rows of a column-major matrix); the device copy is compact.  0 = one dense run.
  if (seg_len <= 0 || src_ld == seg_len) { seg_len = total; src_ld = total; }
  const bool narrow = src_dtype == CA_F64;
  const size_t esz = src_dtype == CA_F64 ? 8 : (src_dtype == CA_F32 || src_dtype == CA_I32) ? 4 : src_dtype == CA_U16 ? 2 : 1;
  const size_t dsz = narrow ? 4 : esz;
  // 16 MiB of staged bytes per chunk, three pinned buffers, eight threads: measured on the MI355X boxes (256 host threads), tools/ingest_sweep.py --
  // 4 ... 24 threads are level (the loop runs at 85-90 % of the PCIe time of the narrowed bytes; 32+ threads and 64 MiB chunks lose to
  // pinning cost and scheduling noise, 4 MiB chunks to per-chunk overhead)
  const int64_t chunk_elems = (int64_t)(16u << 20) / (int64_t)dsz;
  const int64_t nchunks = (total + chunk_elems - 1) / chunk_elems;
  const int hw = (int)std::thread::hardware_concurrency();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(8, std::max(1, hw / 4)), (total + (1 << 18) - 1) >> 18));
  constexpr int NBMAX = 4;
  const int NB = (int)std::min<int64_t>(3, std::max<int64_t>(nchunks, 2));
  void* pin[NBMAX] = {};
  hipEvent_t ev[NBMAX] = {};
  auto fail = [&](const char* what, hipError_t e) { err = std::string(what) + ": " + hipGetErrorString(e); return CA_ERR_HIP; };
  hipError_t e = hipSuccess;
  {   // ONE pinned allocation for all buffers (pinning costs ~0.1 ms per MiB plus a fixed cost per call)
    void* base = nullptr;
    const size_t one = (size_t)std::min<int64_t>(chunk_elems, total) * dsz;
    e = hipHostMalloc(&base, one * NB);
    for (int b = 0; b < NB && e == hipSuccess; ++b) { pin[b] = reinterpret_cast<char*>(base) + one * b; e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming); }
  }
  std::atomic<int64_t> free_for[NBMAX];   // the chunk that may now be written into pinned buffer b
  for (int b = 0; b < NBMAX; ++b) free_for[b].store(b);
  std::vector<std::atomic<int>> filled((size_t)nchunks);
  for (auto& f : filled) f.store(0);
  std::atomic<int> inexact{0}, abort_all{0};
  auto fill_slice = [&](int64_t c, int t) {
    const int b = (int)(c % NB);
    const int64_t e0 = c * chunk_elems, n = std::min<int64_t>(chunk_elems, total - e0);
    // slice t of the chunk (multiples of 4096 elements: whole pages of the source, so the threads' pieces do not share lines)
    const int64_t per = ((n + T - 1) / T + 4095) & ~(int64_t)4095;
    const int64_t a = std::min<int64_t>(n, (int64_t)t * per), z = std::min<int64_t>(n, a + per);
    int bad = 0;
    for (int64_t i = a; i < z;) {   // run by run of the source (one run when it is dense)
      const int64_t lin = e0 + i, seg = lin / seg_len, off = lin - seg * seg_len, m = std::min<int64_t>(z - i, seg_len - off);
      const int64_t s0 = seg * src_ld + off;
      if (narrow) {
        const double* sp = reinterpret_cast<const double*>(src) + s0;
        float* dp = reinterpret_cast<float*>(pin[b]) + i;
        for (int64_t j = 0; j < m; ++j) { const double v = sp[j]; const float f = (float)v; dp[j] = f; bad |= ((double)f != v); }   // (NaN flags itself)
      } else {
        memcpy(reinterpret_cast<char*>(pin[b]) + (size_t)i * esz, reinterpret_cast<const char*>(src) + (size_t)s0 * esz, (size_t)m * esz);
      }
      i += m;
    }
    if (bad) inexact.store(1, std::memory_order_relaxed);
    filled[(size_t)c].fetch_add(1, std::memory_order_release);
  };
  auto worker = [&](int t) {
    for (int64_t c = 0; c < nchunks; ++c) {
      while (free_for[c % NB].load(std::memory_order_acquire) < c) { if (abort_all.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
      fill_slice(c, t);
    }
  };
  std::vector<std::thread> pool;
  if (e == hipSuccess) {
    try { for (int t = 1; t < T; ++t) pool.emplace_back(worker, t); }
    catch (...) {   // thread limit of a restricted container: a std::system_error must not cross the C ABI -- stop cleanly
      abort_all.store(1);
      for (auto& th : pool) th.join();
      for (int b = 0; b < NB; ++b) if (ev[b]) hipEventDestroy(ev[b]);
      if (pin[0]) hipHostFree(pin[0]);
      err = "upload of the count matrix: cannot start the host conversion threads";
      return CA_ERR_NOMEM;
    }
  }
  int rc = CA_OK;
  if (e != hipSuccess) rc = fail("pinned staging buffers of the count matrix", e);
  // this thread is worker 0 of every chunk and the one that queues the copies; the other workers run ahead into the free buffers
  for (int64_t c = 0; c < nchunks && rc == CA_OK; ++c) {
    const int b = (int)(c % NB);
    const int64_t e0 = c * chunk_elems, n = std::min<int64_t>(chunk_elems, total - e0);
    fill_slice(c, 0);   // (free_for[b] >= c holds: this thread released it, below)
    while (filled[(size_t)c].load(std::memory_order_acquire) < T) std::this_thread::yield();
    e = hipMemcpyAsync(reinterpret_cast<char*>(dst_dev) + (size_t)e0 * dsz, pin[b], (size_t)n * dsz, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(ev[b], stream);
    // the buffer that chunk c + 1 wants held chunk c + 1 - NB: once THAT copy is done it is free (the copies queued since run meanwhile)
    const int64_t prev = c + 1 - NB;
    if (e == hipSuccess && prev >= 0) e = hipEventSynchronize(ev[prev % NB]);
    if (e != hipSuccess) { rc = fail("upload of the count matrix", e); break; }
    if (prev >= 0) free_for[prev % NB].store(c + 1, std::memory_order_release);
  }
  if (rc != CA_OK) abort_all.store(1);
  for (auto& th : pool) th.join();
  if (rc == CA_OK) { e = hipStreamSynchronize(stream); if (e != hipSuccess) rc = fail("upload of the count matrix", e); }
  for (int b = 0; b < NB; ++b) if (ev[b]) hipEventDestroy(ev[b]);
  if (pin[0]) hipHostFree(pin[0]);
  if (res) res->inexact = inexact.load() != 0;
  return rc;
}

int upload_y(ca_engine* h, const ca_problem* p) {
  // the caller's matrix: N_src x G_src when a selection is given (ca_problem.cell_index / gene_index), else N x G
  const bool sel = p->cell_index || p->gene_index;
  const int64_t Ns = sel ? p->N_src : h->N;
  const int64_t Gs = sel ? (int64_t)p->G_src : (int64_t)h->G;
  const int64_t total = Ns * Gs;
  size_t esz = p->y_dtype == CA_F64 ? 8 : (p->y_dtype == CA_F32 || p->y_dtype == CA_I32) ? 4 : p->y_dtype == CA_U16 ? 2 : 1;
  // ABI 6: the source may be a block of a larger matrix (ca_problem.y_ld): `nrun` runs of `run` contiguous counts, `ld` counts apart
  const int64_t run = p->layout == CA_COL_MAJOR ? Ns : Gs, nrun = p->layout == CA_COL_MAJOR ? Gs : Ns;
  const int64_t ld = p->y_ld > 0 ? p->y_ld : run;
  if (ld < run) { h->err = "y_ld is smaller than the matrix it strides over"; return CA_ERR_INVALID; }
  const void* src = p->Y;
  void* staging = nullptr;
  void* cut = nullptr;
  int64_t* ci_dev = nullptr; int32_t* gi_dev = nullptr;
  auto cleanup = [&]() { if (staging) hipFree(staging); if (cut) hipFree(cut); if (ci_dev) hipFree(ci_dev); if (gi_dev) hipFree(gi_dev); };
  int y_dtype = p->y_dtype;   // of `src` as the kernels below see it
  if (!p->y_on_device) {
    // chunked through pinned double buffers, float64 narrowed to float32 on the way (ingest_host_matrix): the device staging is the
    // source's layout at <= 4 bytes per count
    if (y_dtype == CA_F64) esz = 4;
    {
      const hipError_t e = hipMalloc(&staging, (size_t)total * esz);
      if (e != hipSuccess) { h->err = std::string("hipMalloc of ") + std::to_string((size_t)total * esz) + " bytes (count matrix staging): " + hipGetErrorString(e); return CA_ERR_NOMEM; }
    }
    ingest_result ir;
    if (p->y_dtype == CA_F64) {
      int rci;
      try { rci = ingest_host_matrix(h->stream, h->err, p->Y, total, p->y_dtype, staging, &ir, run, ld); }
      catch (const std::exception& ex) { h->err = std::string("upload of the count matrix: ") + ex.what(); rci = CA_ERR_NOMEM; }   // (nothing unwinds through the C ABI)
      if (rci != CA_OK) { cleanup(); return rci; }
    } else if ((ld == run ? hipMemcpy(staging, p->Y, (size_t)total * esz, hipMemcpyHostToDevice)
                          : hipMemcpy2D(staging, (size_t)run * esz, p->Y, (size_t)ld * esz, (size_t)run * esz, (size_t)nrun, hipMemcpyHostToDevice)) != hipSuccess) {
      cleanup(); h->err = "upload of the count matrix failed"; return CA_ERR_HIP;
    }
    if (p->y_dtype == CA_F64) y_dtype = CA_F32;
    if (ir.inexact && sel) {
      // some double of the RAW matrix is not a float, but it may lie outside the selection (which is all the fit may judge): take the
      // matrix up again as it is, 8 bytes per count, and let the gather below cut it first -- the rare path
      hipFree(staging); staging = nullptr;
      const hipError_t e = hipMalloc(&staging, (size_t)total * 8);
      if (e != hipSuccess) { cleanup(); h->err = std::string("hipMalloc of the count matrix staging: ") + hipGetErrorString(e); return CA_ERR_NOMEM; }
      if ((ld == run ? hipMemcpy(staging, p->Y, (size_t)total * 8, hipMemcpyHostToDevice)
                     : hipMemcpy2D(staging, (size_t)run * 8, p->Y, (size_t)ld * 8, (size_t)run * 8, (size_t)nrun, hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup(); h->err = "upload of the count matrix failed"; return CA_ERR_HIP;
      }
      y_dtype = CA_F64;
    } else if (ir.inexact) {
      // a double that is not exactly a float can only be an error: NaN / negative takes precedence, as in the scan below
      double* maxv = nullptr;
      if (hipMalloc((void**)&maxv, 32) != hipSuccess) { cleanup(); h->err = "hipMalloc failed"; return CA_ERR_NOMEM; }
      double hm[4] = {0, 0, 0, 0};
      hipMemsetAsync(maxv, 0, 32, h->stream);
      hipLaunchKernelGGL((k_scan_y<float>), dim3(std::min<int64_t>(4096, cdiv(total, CA_TB))), dim3(CA_TB), 0, h->stream, (const float*)staging, total, maxv,
                         (int*)(maxv + 1), (unsigned long long*)(maxv + 2));
      hipMemcpyAsync(hm, maxv, 32, hipMemcpyDeviceToHost, h->stream);
      const hipError_t e = hipStreamSynchronize(h->stream);
      hipFree(maxv);
      cleanup();
      if (e != hipSuccess) { h->err = hipGetErrorString(e); return CA_ERR_HIP; }
      int fl;
      memcpy(&fl, &hm[1], sizeof(int));
      h->err = (fl & 2) ? "count matrix has negative or NaN entries" : "counts are not exactly representable in the on-device storage type";
      return CA_ERR_INVALID;
    }
    src = staging;
  }
  // strides of `src` as the kernels below see it: the staging copy of a host matrix is compact, a device matrix keeps its leading dimension
  const int64_t ldd = p->y_on_device ? ld : run;
  int64_t sn = p->layout == CA_COL_MAJOR ? 1 : ldd;
  int64_t sg = p->layout == CA_COL_MAJOR ? ldd : 1;
  int rc = CA_OK;
  if (sel || (p->y_on_device && ld != run)) {   // (a strided device block is compacted the same way: the scan below walks a dense matrix)
    // selection lists: validated on the host (strictly increasing, in range), applied by a device gather into a compact
    // row-major copy of the source type -- the storage scan / conversion below then sees only the selected counts
    auto bad = [&](const char* m) { cleanup(); h->err = m; return CA_ERR_INVALID; };
    if (p->cell_index) {
      for (int64_t n = 0; n < h->N; ++n)
        if (p->cell_index[n] < 0 || p->cell_index[n] >= Ns || (n > 0 && p->cell_index[n] <= p->cell_index[n - 1]))
          return bad("cell_index must be strictly increasing and within [0, N_src)");
      if (hipMalloc((void**)&ci_dev, (size_t)h->N * sizeof(int64_t)) != hipSuccess ||
          hipMemcpy(ci_dev, p->cell_index, (size_t)h->N * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) return bad("cell_index upload failed");
    } else if (h->N != Ns) return bad("N must equal N_src when cell_index is NULL");
    if (p->gene_index) {
      for (int g = 0; g < h->G; ++g)
        if (p->gene_index[g] < 0 || p->gene_index[g] >= Gs || (g > 0 && p->gene_index[g] <= p->gene_index[g - 1]))
          return bad("gene_index must be strictly increasing and within [0, G_src)");
      if (hipMalloc((void**)&gi_dev, (size_t)h->G * sizeof(int32_t)) != hipSuccess ||
          hipMemcpy(gi_dev, p->gene_index, (size_t)h->G * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return bad("gene_index upload failed");
    } else if (h->G != Gs) return bad("G must equal G_src when gene_index is NULL");
    switch (y_dtype) {
      case CA_F64: rc = gather_y<double>(h, src, &cut, sn, sg, ci_dev, gi_dev); break;
      case CA_F32: rc = gather_y<float>(h, src, &cut, sn, sg, ci_dev, gi_dev); break;
      case CA_I32: rc = gather_y<int32_t>(h, src, &cut, sn, sg, ci_dev, gi_dev); break;
      case CA_U16: rc = gather_y<uint16_t>(h, src, &cut, sn, sg, ci_dev, gi_dev); break;
      case CA_U8: rc = gather_y<uint8_t>(h, src, &cut, sn, sg, ci_dev, gi_dev); break;
      default: h->err = "unknown y_dtype"; rc = CA_ERR_INVALID;
    }
    if (rc != CA_OK) { cleanup(); return rc; }
    if (staging) { hipFree(staging); staging = nullptr; }
    src = cut; sn = h->G; sg = 1;
  }
  switch (y_dtype) {
    case CA_F64: rc = scan_and_convert<double>(h, (const double*)src, sn, sg); break;
    case CA_F32: rc = scan_and_convert<float>(h, (const float*)src, sn, sg); break;
    case CA_I32: rc = scan_and_convert<int32_t>(h, (const int32_t*)src, sn, sg); break;
    case CA_U16: rc = scan_and_convert<uint16_t>(h, (const uint16_t*)src, sn, sg); break;
    case CA_U8: rc = scan_and_convert<uint8_t>(h, (const uint8_t*)src, sn, sg); break;
    default: h->err = "unknown y_dtype"; rc = CA_ERR_INVALID;
  }
  cleanup();
  return rc;
}

template <typename YT>
void launch_prep(ca_engine* h, const double* logL, const double* extra) {
  hipLaunchKernelGGL((k_prep_cells<YT>), dim3((unsigned)h->N), dim3(CA_TB), 0, h->stream, (const YT*)h->Y, logL, extra, h->A, h->cn,
                     h->s64, h->s32, h->N, h->G, h->Gp, h->C, h->ovf_rowptr, h->ovf_col, h->ovf_val);
}

template <>
void launch_prep<uint8_t>(ca_engine* h, const double* logL, const double* extra) {
  if (!variant_on(h, CA_VAR_PREP_FAST, "CA_PREP_FAST")) {
    hipLaunchKernelGGL((k_prep_cells<uint8_t>), dim3((unsigned)h->N), dim3(CA_TB), 0, h->stream, (const uint8_t*)h->Y, logL, extra, h->A,
                       h->cn, h->s64, h->s32, h->N, h->G, h->Gp, h->C, h->ovf_rowptr, h->ovf_col, h->ovf_val);
    return;
  }
  const int grid = (int)std::min<int64_t>(cdiv(h->N, CA_TB / 64), (int64_t)h->n_cu * 8);
  hipLaunchKernelGGL(k_prep_cells_u8, dim3(grid), dim3(CA_TB), 0, h->stream, (const uint8_t*)h->Y, logL, extra, h->A, h->cn, h->s64,
                     h->s32, h->N, h->G, h->Gp, h->C, h->ovf_rowptr, h->ovf_col, h->ovf_val);
}

int create_impl(ca_engine* h, const ca_problem* p) {
  const int N = (int)h->N; (void)N;
  HIPCK(h, hipSetDevice(h->device));
  hipDeviceProp_t prop;
  HIPCK(h, hipGetDeviceProperties(&prop, h->device));
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  // (a low-priority side stream, or s_setprio 3 in the sweep, only moves the Y stream's time out from under the sweep:
  //  measured 2090 -> 2100 and 2090 -> 1730 it/s; the cell epilogue waits for it either way)
  HIPCK(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  HIPCK(h, hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
  HIPCK(h, hipEventCreateWithFlags(&h->ev_ydone, hipEventDisableTiming));
  HIPCK(h, hipEventCreateWithFlags(&h->ev_ywdone, hipEventDisableTiming));
  // side stream for the count-matrix products: pays from ~4e7 counts up (12.5k x 5k: 9543 it/s against 9207 in line;
  // 10k x 2k: 10917 against 13017 -- two cross-stream events per iteration cost more than the overlap returns)
  h->async_y = variant_on(h, CA_VAR_ASYNC_Y, "CA_ASYNC_Y") &&
               ((double)h->N * (double)h->G >= 4e7 || variantx_on(h, CA_VARX_ASYNC_SMALL, "CA_ASYNC_SMALL"));
  HIPCK(h, hipHostMalloc((void**)&h->host_pinned, 64 * sizeof(double)));
  memset(h->host_pinned, 0, 64 * sizeof(double));
  if (hipHostGetDevicePointer((void**)&h->host_dev, h->host_pinned, 0) != hipSuccess) { h->host_dev = nullptr; (void)hipGetLastError(); }
  h->mon_tail = no_small_args();
  h->tail_fuse = variant_on(h, CA_VAR_TAIL_FUSE, "CA_TAIL_FUSE");
  h->pre_ok = variant_on(h, CA_VAR_PRE, "CA_PRE");
  h->upd_merge = h->tail_fuse && h->pre_ok && variant_on(h, CA_VAR_UPDATE_MERGE, "CA_UPDATE_MERGE");
  h->p2p_ride = h->tail_fuse && variant_on(h, CA_VAR_P2P_RIDE, "CA_P2P_RIDE");
  h->run_gate = h->upd_merge && variant_on(h, CA_VAR_RUN_GATE, "CA_RUN_GATE");
  {   // the device's wall clock (s_memrealtime) in ticks per microsecond, asked of the runtime instead of assumed
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) == hipSuccess && khz >= 1000) h->ticks_per_us = (double)khz / 1000.0;
    else (void)hipGetLastError();
  }
  h->gate_ticks = (unsigned long long)(h->ticks_per_us * (double)(h->opt.gate_timeout_us > 0 ? h->opt.gate_timeout_us : 1000));
  h->run_fwd = h->run_gate && variantx_on(h, CA_VARX_RUN_FWD, "CA_RUN_FWD");   // opt-in: see the header (no runtime call may block between a gated launch and its answer)
  h->pair_elbo = variant_on(h, CA_VAR_PAIR_ELBO, "CA_PAIR_ELBO");
  CACK(upload_y(h, p));
  const int G = h->G, C = h->C, K = h->K, P = h->P, S = h->S, D = h->D;
  const int64_t Nn = h->N;
  h->nchunk = cdiv(C, CA_CW);
  h->ngblk = cdiv(G, CA_TB);
  {
    int CP = 1;
    while (CP < C) CP <<= 1;
    h->ncblk = (C <= 64) ? std::min(cdiv(Nn, CA_TB / CP), 16 * 256) : cdiv(Nn, CA_TB);   // k_cell_par: grid-stride over groups of CA_TB / CP cells
  }
  // ---- sweep decomposition
  const int target_blocks = 8 * h->n_cu;
  // one full round of resident blocks (8 x 256 threads per CU) when the cell blocks alone do not fill the chip
  const int nfb = cdiv(Nn, CA_TB * kFwdR);
  // enough blocks for ~one resident round, but never more than 16 gene slices: every slice adds an N x 16 float
  // partial that the cell epilogue has to read back (at 12.5k cells 78 slices cost 2x the sweep itself)
  h->gsplit = std::max(1, std::min(std::min(target_blocks / std::max(nfb, 1), 16), std::max(1, G / 64)));
  h->gsplit = std::max(h->gsplit, cdiv(G, 1024));      // LDS slice: at most 1024 genes x (8 + D) floats = 64 KB
  if (const int t = tune_val(h, CA_TUNE_GSPLIT, "CA_GSPLIT")) h->gsplit = std::max(std::max(1, t), cdiv(G, 1024));   // tuning override
  h->gchunk = cdiv(G, h->gsplit);
  h->gsplit = cdiv(G, h->gchunk);
  // genes per lane of the backward sweep: more genes amortise the per-cell wave reduction (tools/bwd_lab2.hip)
  h->RG = G >= 1024 ? 4 : 1;   // RG = 8 measured equal in the engine (272 vs 276 us): kept selectable, not default
  if (const int r = tune_val(h, CA_TUNE_RG, "CA_RG")) { if (r == 1 || r == 4 || r == 8) h->RG = r; }
  h->ntile = cdiv(G, 64 * h->RG);
  const int gblocks = cdiv(h->ntile, CA_TB / 64);
  h->csplit = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(target_blocks, gblocks), std::max<int64_t>(1, Nn / 64)));
  if (const int t = tune_val(h, CA_TUNE_CSPLIT, "CA_CSPLIT")) h->csplit = std::max(1, t);   // tuning override
  h->cchunk = (Nn + h->csplit - 1) / h->csplit;
  h->csplit = cdiv(Nn, h->cchunk);
  h->TR = 128;
  const int tr_set = tune_val(h, CA_TUNE_TR, "CA_TR");
  if (tr_set) h->TR = std::min(128, std::max(1, tr_set));   // tuning override (k_ypass keeps psi of <= 128 rows)
  h->nrb = cdiv(Nn, h->TR);
  while (!tr_set && (int64_t)h->nrb * h->nseg < 4 * h->n_cu && h->TR > 32) { h->TR /= 2; h->nrb = cdiv(Nn, h->TR); }
  h->nrg = cdiv(h->nrb, CA_TB / 64);
  // ---- constants
  std::vector<double> Lrm((size_t)G * C), logL((size_t)G * C);
  std::vector<float> Lb((size_t)h->nchunk * G * CA_CW, 0.f);
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < C; ++c) {
      const double v = p->L[hidx(p->layout, g, c, G, C)];
      Lrm[(size_t)g * C + c] = v;
      logL[(size_t)g * C + c] = std::log(v);
      Lb[((size_t)(c / CA_CW) * G + g) * CA_CW + (c % CA_CW)] = (float)v;
    }
  CACK(dalloc(h, &h->Lb, (int64_t)Lb.size()));
  CACK(upload_f(h, h->Lb, Lb));
  double* logL_dev = nullptr; double* extra_dev = nullptr;
  HIPCK(h, hipMalloc((void**)&logL_dev, logL.size() * sizeof(double)));
  HIPCK(h, hipMemcpy(logL_dev, logL.data(), logL.size() * sizeof(double), hipMemcpyHostToDevice));
  if (p->extra_loglik) {
    std::vector<double> ex((size_t)Nn * C);
    for (int64_t n = 0; n < Nn; ++n)
      for (int c = 0; c < C; ++c) ex[(size_t)n * C + c] = p->extra_loglik[hidx(p->layout, n, c, Nn, C)];
    HIPCK(h, hipMalloc((void**)&extra_dev, ex.size() * sizeof(double)));
    HIPCK(h, hipMemcpy(extra_dev, ex.data(), ex.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  CACK(dalloc(h, &h->A, Nn * C));
  CACK(dalloc(h, &h->cn, Nn));
  CACK(dalloc(h, &h->s64, Nn));
  CACK(dalloc(h, &h->s32, Nn));
  CACK(dalloc(h, &h->colsum, G));
  if (h->ystore == CA_YSTORE_U8) launch_prep<uint8_t>(h, logL_dev, extra_dev);
  else if (h->ystore == CA_YSTORE_U16) launch_prep<uint16_t>(h, logL_dev, extra_dev);
  else launch_prep<float>(h, logL_dev, extra_dev);
  HIPCK(h, hipGetLastError());
  SYNC(h);
  hipFree(logL_dev);
  if (extra_dev) hipFree(extra_dev);
  // ---- variables (:240-272)
  // matrix-core backward sweep: needs D == 1, one clone chunk and bf16-exact copy numbers (integers up to 256)
  {
    bool exact = true;
    for (double v : Lrm) {
      const float f = (float)v;
      uint32_t u; memcpy(&u, &f, 4);
      if ((double)f != v || (u & 0xFFFFu) != 0) { exact = false; break; }
    }
    h->bwd_frac = !exact;   // copy numbers that are not bf16-exact: the sweep's two-part form (k_bwd_mfma<.., FRAC>)
    // 9..16 clones (two clone chunks): the sweep's C16 form, integer copy numbers only (fractional ones would need 48 operand slots)
    h->bwd_mfma = (D == 1 || D == 2) && (h->nchunk == 1 || (h->nchunk == 2 && exact && S == 1)) && variant_on(h, CA_VAR_BWD_MFMA, "CA_BWD_MFMA");
    h->N16 = (Nn + 15) / 16 * 16;
    if (h->bwd_mfma) {
      // Gene tiles per wave.  Four amortise the coef operand best (cfg-3: 111.6 us against 122.5 with three, although three fit four waves per SIMD);
      // a small problem has too few wave jobs at four (cfg-2: 32 of them, eight blocks across) and gains from three: cfg-2 39.3 -> 37.5 us per
      // iteration, 12.5k x 5k 63.7 -> 62.6; 25k cells 91.2 -> 95.1 (profiles/r05_small_shapes.txt section 9).  Main variant only.
      h->bwd_tl = (Nn <= CA_BWD_TL3_MAXN && !h->c16 && exact && S == 1 && h->nchunk == 1 && variant_on(h, CA_VAR_BWD_TL3, "CA_BWD_TL3")) ? 3 : CA_BWD_TL;
      h->nwt = cdiv(G, h->bwd_tl * 16);
      const int xb = cdiv(h->nwt, CA_TB / 64);
      {
        // resident blocks per CU from the compiler's register count (LDS, 16 B per cell of the slice, is not the limit
        // at the slice lengths this produces); one or two full rounds instead of "about 8 blocks per CU"
        int per_cu = 4;
        const void* bfn = h->bwd_tl == 3 ? (D == 1 ? (const void*)k_bwd_mfma<3, 1, false> : (const void*)k_bwd_mfma<3, 2, false>)
                                         : (D == 1 ? (const void*)k_bwd_mfma<CA_BWD_TL, 1, false> : (const void*)k_bwd_mfma<CA_BWD_TL, 2, false>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bfn, CA_TB, 16 * 1024) != hipSuccess || per_cu < 1)
          per_cu = 4;
        (void)hipGetLastError();
        const int smax = (int)std::max<int64_t>(1, std::min<int64_t>(Nn / 256, cdiv(2 * (int64_t)per_cu * h->n_cu, xb)));
        h->csplit_m = pick_split(xb, (int64_t)per_cu * h->n_cu, smax, 1e-4);
      }
      if (const int t = tune_val(h, CA_TUNE_CSPLIT_M, "CA_CSPLIT_M")) h->csplit_m = std::max(1, t);
      // mc_samples = 2: the two-sample sweep keeps a second set of d/dF slices in LDS; where that would leave fewer than three blocks per CU
      // (more than 48 KB a block) the cell slices are halved -- two whole rounds where there was one -- with or without the variant
      if (S == 2 && (int64_t)2 * 4 * (((Nn + h->csplit_m - 1) / h->csplit_m + 15) / 16 * 16) * D * 4 > 48 * 1024) h->csplit_m *= 2;
      h->csplit_m = (int)std::max<int64_t>(h->csplit_m, (Nn * D * (S == 2 ? 2 : 1) + 4079) / 4080);   // LDS: [samples x] 4 waves x cchunk x D floats <= 64 KB
      h->cchunk_m = ((Nn + h->csplit_m - 1) / h->csplit_m + 15) / 16 * 16;
      h->csplit_m = cdiv(Nn, h->cchunk_m);
      // fold the column sums of the sweep's partials into the per-gene kernel: a launch and its gap less.  Up to 32k cells in round 2
      // (the fold's loads went out eight at a time, one column after the other: slower than the k_colsum launch above that);
      // with both columns from one load and twenty slices in flight it is level or ahead wherever the slice count is the one
      // resident round (cfg-3 +0.3 %, 50k cells +0.6 %, 200k +0.2 %); long shards with many more slices keep the launch
      h->fold_gsum = (Nn <= 32768 || (h->csplit_m <= 64 && S + D == 2) || variantx_on(h, CA_VARX_FOLD_ALWAYS, "CA_FOLD_ALWAYS")) && S + D <= 12 && h->tail_fuse &&
                     variant_on(h, CA_VAR_FOLD_GSUM, "CA_FOLD_GSUM");
      CACK(dalloc(h, &h->coefq, (int64_t)S * h->N16 * 32));
    }
  }
  CACK(dalloc(h, &h->F, h->N16 * std::max(D, 1)));
  CACK(dalloc(h, &h->m_psi, Nn * std::max(K, 1)));
  CACK(dalloc(h, &h->v_psi, Nn * std::max(K, 1)));
  CACK(dalloc(h, &h->g_psi, Nn * std::max(K, 1)));
  CACK(dalloc(h, &h->glogit, Nn * C));
  CACK(dalloc(h, &h->m_gl, Nn * C));
  CACK(dalloc(h, &h->v_gl, Nn * C));
  CACK(dalloc(h, &h->dgl, Nn * C));
  CACK(dalloc(h, &h->V, (int64_t)G * std::max(D, 1)));
  CACK(dalloc(h, &h->m_V, (int64_t)G * std::max(D, 1)));
  CACK(dalloc(h, &h->v_V, (int64_t)G * std::max(D, 1)));
  CACK(dalloc(h, &h->g_V, (int64_t)G * std::max(D, 1)));
  CACK(dalloc(h, &h->Vs, (int64_t)cdiv(G, 32) * 32 * std::max(D, 1)));   // padded to whole 32-gene k-steps (last gene replicated)
  CACK(dalloc(h, &h->loc, G)); CACK(dalloc(h, &h->ls, G));
  CACK(dalloc(h, &h->m_loc, G)); CACK(dalloc(h, &h->v_loc, G));
  CACK(dalloc(h, &h->m_ls, G)); CACK(dalloc(h, &h->v_ls, G));
  CACK(dalloc(h, &h->g_loc, G)); CACK(dalloc(h, &h->g_ls, G));
  CACK(dalloc(h, &h->vchi, std::max(K, 1))); CACK(dalloc(h, &h->m_v, std::max(K, 1))); CACK(dalloc(h, &h->v_v, std::max(K, 1)));
  CACK(dalloc(h, &h->g_v, std::max(K, 1)));
  CACK(dalloc(h, &h->alpha_u, C)); CACK(dalloc(h, &h->m_a, C)); CACK(dalloc(h, &h->v_a, C)); CACK(dalloc(h, &h->g_a, C));
  CACK(dalloc(h, &h->vchi_alt, std::max(K, 1))); CACK(dalloc(h, &h->alpha_u_alt, C));
  CACK(dalloc(h, &h->vmm_at, 32));
  CACK(dalloc(h, &h->gate_local, 2));
  CACK(dalloc(h, &h->gaux, (int64_t)2 * 5 * G));
  {
    std::vector<float> Fh((size_t)Nn * std::max(D, 1), 0.f);
    if (D > 0) {
      for (int64_t n = 0; n < Nn; ++n) {
        for (int k = 0; k < K; ++k) Fh[(size_t)n * D + k] = (float)p->psi0[hidx(p->layout, n, k, Nn, K)];
        for (int q = 0; q < P; ++q) Fh[(size_t)n * D + K + q] = (float)p->X[hidx(p->layout, n, q, Nn, P)];
      }
    }
    CACK(upload_f(h, h->F, Fh));
    std::vector<float> l0((size_t)G);
    if (p->loc0)   // else: filled below from the data (device form of mu_guess, R/inference-tflow.R:220-235)
      for (int g = 0; g < G; ++g) l0[g] = (float)p->loc0[g];
    CACK(upload_f(h, h->loc, l0));
  }
  // ---- pass buffers
  CACK(dalloc(h, &h->mu32, (int64_t)S * G));
  CACK(dalloc(h, &h->Mb, (int64_t)S * h->nchunk * G * CA_CW));
  // 9..16 clones (c16): the fused machinery with ONE draw per sweep in all sixteen operand columns -- only in its default form
  // (matrix-core sweeps, sweep + cell epilogue in one kernel); otherwise such problems take the plain passes as before
  h->c16 = S == 1 && C > CA_CW && C <= 2 * CA_CW && (D == 1 || D == 2) && h->bwd_mfma && h->tail_fuse &&
           variant_on(h, CA_VAR_FWD_MFMA, "CA_FWD_MFMA") && variant_on(h, CA_VAR_FWD_CELL, "CA_FWD_CELL");
  if (h->nchunk == 2 && !h->c16) h->bwd_mfma = false;
  // mc_samples = 2 (s2): the two column halves carry the two SAMPLES of one pass; same conditions, up to eight clones
  h->s2 = S == 2 && C <= CA_CW && (D == 1 || D == 2) && h->bwd_mfma && h->tail_fuse &&
          variant_on(h, CA_VAR_FWD_MFMA, "CA_FWD_MFMA") && variant_on(h, CA_VAR_FWD_CELL, "CA_FWD_CELL");
  h->fused_ok = ((S == 1 && (C <= CA_CW || h->c16)) || h->s2) && variant_on(h, CA_VAR_FUSED, "CA_FUSED");
  if (!h->fused_ok) { if (h->nchunk == 2) h->bwd_mfma = false; h->c16 = false; h->s2 = false; }
  if (h->s2) h->pair_elbo = false;   // (two draws per sweep need two column halves of their own)
  if (h->fused_ok) {
    h->frow = (2 * C <= 8) ? 8 : 16;
    // matrix-core forward sweep (k_fwd_mfma): D in {1, 2}; few gene slices, streamed through LDS
    h->fwd_mfma = (D == 1 || D == 2) && variant_on(h, CA_VAR_FWD_MFMA, "CA_FWD_MFMA");
    h->fwd_cell = h->fwd_mfma && h->tail_fuse && variant_on(h, CA_VAR_FWD_CELL, "CA_FWD_CELL");
    {
      // cells per block of k_fwd_cell: 16 * TL -- measured, not derived.  96-cell blocks (fewest re-reads of the B operand
      // from L2) as soon as there is one full round of them, the launch then carries a second, 32-cell block size for the
      // cells that do not fill a whole round (k_fwd_cell_mix, below).  With that, TL = 6 is best or within 1 % of the best of
      // {4, 5, 6} from 35k to 400k cells (35k: 6310 / 6207 / 6046 it/s for 6 / 5 / 4; 50k: 5012 / 4916 / 4951; 70k: 3499 /
      // 3459 / 3523; 100k: 2691 / 2594 / 2625); 128-cell blocks never won (200k: 1346 vs 1363, 400k: 609 vs 638 it/s).
      // Small shards: 32-cell blocks, or CUs are left with one block or none (25k: 7619 / 7443 / 6701 for TL = 2 / 4 / 6
      // without the second block size, 7535 for 6 with it).
      // (16-cell blocks, fc_tl = 1, were tried for shards below 16k cells -- twice the waves per SIMD -- and lost: 12.5k cells
      //  39 us against 35, every block re-reads the B operand)
      h->fc_tl = (h->fwd_cell && cdiv(Nn, 64) < 2 * h->n_cu) ? 2 : 6;
      // fewer 32-cell blocks than CUs: 16-cell blocks, where the int8 stream rides (round 3, with this round's kernels: 6250 cells
      // 17.2k -> 18.8k it/s; at 12.5k cells and at 10k x 2k they still lose, 13.7k vs 14.2k and 21.7k vs 22.1k)
      if (h->fwd_cell && h->fc_tl == 2 && cdiv(Nn, 32) < h->n_cu && h->ystore == CA_YSTORE_U8 && K == 1 && !h->c16 && h->fused_ok &&
          variant_on(h, CA_VAR_Y_MFMA1, "CA_Y_MFMA1") && variant_on(h, CA_VAR_Y_RIDE, "CA_Y_RIDE") && !variantx_on(h, CA_VARX_Y_MFMA2, "CA_Y_MFMA2"))
        h->fc_tl = 1;   // (only where the int8 stream will ride: the other streams' merged kernels exist for 32- and 96-cell blocks)
      if (const int t = tune_val(h, CA_TUNE_FC_TL, "CA_FC_TL")) { if (t == 1 || t == 2 || t == 4 || t == 5 || t == 6 || t == 8) h->fc_tl = t; }
      if ((h->c16 || h->s2) && h->fc_tl != 2) h->fc_tl = 6;   // (the two-operand-set kernels -- sixteen clones, four draws of mc_samples = 2 -- exist for the two default block shapes)
      if (h->s2 && !h->fwd_cell) { h->s2 = false; h->fused_ok = false; }
      h->ncblk_f = cdiv(Nn, 16 * h->fc_tl);
      // two block sizes in one launch (k_fwd_cell_mix)
      if (h->fwd_cell && (h->fc_tl == 4 || h->fc_tl == 5 || h->fc_tl == 6 || h->fc_tl == 8) && (D == 1 || D == 2)) {
        // measured at 100k cells (1042 blocks of 96): 1024 big + 53 small 2706 it/s, 768 + 821: 2682, 512 + 1589: 2669, all big
        // 2642 -- what pays is every CU getting the same number of big blocks, so: whole multiples of the CU count in big
        // blocks, the remainder (less than one big block per CU) in small ones
        int nbig = (h->ncblk_f / h->n_cu) * h->n_cu;
        if (const int t = tune_val(h, CA_TUNE_FC_NBIG, "CA_FC_NBIG")) nbig = t < 0 ? 0 : t;
        if (nbig > 0 && h->ncblk_f > nbig) {
          h->fc_nbig = nbig;
          h->ncblk_f = nbig + cdiv(Nn - (int64_t)nbig * 16 * h->fc_tl, 32);
        }
      }
    }
    int zsplit = h->gsplit;
    if (h->fwd_mfma) {
      h->frow = 16;
      h->nk32 = cdiv(G, 32);
      const int cblocks = cdiv(Nn, (CA_TB / 64) * CA_FM_TL * 16);
      {
        int per_cu = 4;
        const void* fn = (D == 1) ? (const void*)k_fwd_mfma<1> : (const void*)k_fwd_mfma<2>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, CA_TB, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        (void)hipGetLastError();
        // every gene slice adds an N x 16 float partial (written here, re-read by the cell epilogue): ~2 % of the sweep
        h->fsplit = pick_split(cblocks, (int64_t)per_cu * h->n_cu, std::max(1, std::min(16, h->nk32 / CA_FM_KC)), 0.02);
      }
      if (const int t = tune_val(h, CA_TUNE_FSPLIT, "CA_FSPLIT")) h->fsplit = std::max(1, std::min(t, h->nk32));
      h->fkchunk = cdiv(h->nk32, h->fsplit);
      h->fsplit = cdiv(h->nk32, h->fkchunk);
      zsplit = h->fsplit;
      CACK(dalloc(h, &h->Mq, (int64_t)h->nk32 * 2 * 64 * 8 * ((h->c16 || h->s2) ? 2 : 1)));   // zero-filled: padding genes and columns stay 0 (9..16 clones: one image per draw)
    } else {
      CACK(dalloc(h, &h->Mb2, (int64_t)G * h->frow));
    }
    CACK(dalloc(h, &h->mu32B, G));
    CACK(dalloc(h, &h->gene_partB, (int64_t)h->ngblk * (3 + K)));
    CACK(dalloc(h, &h->gene_part_alt, (int64_t)h->ngblk * (3 + K)));
    CACK(dalloc(h, &h->gene_partB_alt, (int64_t)h->ngblk * (3 + K)));
    if (!h->fwd_cell) CACK(dalloc(h, &h->Zpart2, (int64_t)zsplit * Nn * h->frow));   // k_fwd_cell keeps Z in the block
  }
  // The side stream accompanies the cell-kernel path only.  Round 5, found by a parity run at 150k cells x 18 clones: with the plain passes (more than sixteen
  // clones, D >= 3, K = 0 ...) the loop's pipelining (deferred monitor tail, prologue hand-over, gated update) and the side stream's deferred start are not ordered
  // against each other -- ca_run's ELBOs from the second iteration on were wrong by 1e-4 ... 1e-1 and differed from run to run, silently, from 4e7 counts up
  // (below that the Y pass runs in line; the suite's side-stream cases all take the cell kernel).  Those shapes run the Y pass in line at every size.
  if (!h->fwd_cell) h->async_y = false;
  if (verbose(h))
    fprintf(stderr, "[clonealign_hip] N=%lld G=%d C=%d D=%d n_cu=%d gsplit=%d gchunk=%d csplit=%d fused=%d fwd_mfma=%d fsplit=%d fkchunk=%d "
            "bwd_mfma=%d csplit_m=%d cchunk_m=%lld nwt=%d fwd_cell=%d fc_tl=%d fc_nbig=%d ncblk_f=%d ystore=%d async_y=%d\n", (long long)Nn, G, C, D, h->n_cu, h->gsplit,
            h->gchunk, h->csplit, (int)h->fused_ok, (int)h->fwd_mfma, h->fsplit, h->fkchunk, (int)h->bwd_mfma, h->csplit_m,
            (long long)h->cchunk_m, h->nwt, (int)h->fwd_cell, h->fc_tl, h->fc_nbig, h->ncblk_f, h->ystore, (int)h->async_y);
  CACK(dalloc(h, &h->vmm, 2 * std::max(D, 1)));
  CACK(dalloc(h, &h->vmm_part, (int64_t)h->ngblk * 2 * std::max(D, 1)));
  CACK(dalloc(h, &h->etamax2, h->N16));
  CACK(dalloc(h, &h->gene_part, (int64_t)h->ngblk * (3 + K)));
  CACK(dalloc(h, &h->Zpart, (int64_t)S * h->gsplit * h->nchunk * Nn * CA_CW));
  CACK(dalloc(h, &h->coef, (int64_t)S * h->nchunk * Nn * CA_CW));
  CACK(dalloc(h, &h->scratch, Nn * C));
  const int64_t n_cpart = std::max(std::max(h->ncblk, h->ncblk_f), 2 * h->n_cu);   // (balanced sweep: n_cu blocks + up to n_cu - 1 left-over tiles' blocks)
  CACK(dalloc(h, &h->cell_part, n_cpart * (3 + C)));
  // The series form of the contraction (ca_poly.hip) where it is measured faster than the sweeps: the count-matrix stream then runs as a launch of its own
  // (in line) and the rest is O(N + G) work in half a dozen small launches -- a gain once the sweeps' N G C dominates (100k x 5k x 8: 191 against 253 us per
  // iteration; 50k x 3k x 6: 101 against 107), a loss below (25k x 5k: 112 against 91; 10k x 2k: 76 against 38: launch latencies).  CA_VAR_SERIES switches the
  // pick off, CA_VARX_SERIES forces the form at any size (tests; profiles/r06_series.txt).
#ifdef CA_SERIES_ALWAYS
  const bool series_auto = true;   // (shake-out builds: every eligible shape through the series form, whole GPU suite)
#else
  const bool series_auto = (double)Nn * (double)G >= 1.4e8 && Nn >= 32768;
#endif
  h->poly = ca_poly_ok(D, S, C) && h->fused_ok && !h->c16 && !h->s2 && K == 1 &&
            ((series_auto && variant_on(h, CA_VAR_SERIES, "CA_SERIES")) || variantx_on(h, CA_VARX_SERIES, "CA_SERIES_ON"));
  if (h->poly) {
    int CPp = 1;
    while (CPp < C) CPp <<= 1;
    const int per_cu = (h->opt.reserved[0] >= 1 && h->opt.reserved[0] <= 8) ? h->opt.reserved[0] : 2;   // (ca_options.reserved[0]: cell blocks per CU of the series form, lab)
    const int ncb = (int)std::min<int64_t>(cdiv(Nn, CA_TB / CPp), (int64_t)per_cu * h->n_cu);   // (<= ncblk: cell_part has a row for each)
    h->poly_side = h->opt.reserved[1] == 1;   // (ca_options.reserved[1] = 1: the count-matrix stream on the side stream beside the cell launch; default in line)
    const size_t wb = ca_poly_workspace_bytes(G, ncb);
    CACK(dalloc(h, &h->poly_mem, (int64_t)wb));
    HIPCK(h, hipMemsetAsync(h->poly_mem, 0, wb, h->stream));
    ca_poly_bind(&h->pws, h->poly_mem, G, ncb);
    HIPCK(h, hipHostMalloc((void**)&h->poly_ring, 16 * 4 * sizeof(double), hipHostMallocMapped));
    memset(h->poly_ring, 0, 16 * 4 * sizeof(double));
    if (hipHostGetDevicePointer((void**)&h->poly_ring_dev, h->poly_ring, 0) != hipSuccess) { (void)hipGetLastError(); h->poly = false; }
    CACK(dalloc(h, &h->poly_zero, Nn));
    HIPCK(h, hipMemsetAsync(h->poly_zero, 0, (size_t)Nn * sizeof(float), h->stream));
  }
  CACK(dalloc(h, &h->ee_partB, n_cpart));
  CACK(dalloc(h, &h->gpart, (int64_t)std::max(h->csplit, h->csplit_m) * G * (S + D)));
  CACK(dalloc(h, &h->dFpart, (int64_t)std::max(h->ntile, h->nwt) * Nn * std::max(D, 1)));
  CACK(dalloc(h, &h->YWpart, (int64_t)(h->nseg + 1) * Nn * std::max(K, 1)));
  CACK(dalloc(h, &h->YTpart, (int64_t)(h->nrb + 1) * h->Gp * std::max(K, 1)));
  CACK(dalloc(h, &h->YW, Nn * std::max(K, 1)));
  h->n_yw = cdiv(Nn, CA_TB);
  CACK(dalloc(h, &h->yw_part, h->n_yw));
  CACK(dalloc(h, &h->ytpsi, (int64_t)h->Gp * std::max(K, 1)));
  // ---- count-matrix products on the int8 matrix cores: two tiled copies of the 1-byte matrix (cell-tiled for Y.W,
  //      gene-tiled for Y^T.psi), each in the operand layout of v_mfma_i32_16x16x64_i8 (ca_ymfma.hip.h)
  if (h->ystore == CA_YSTORE_U8 && K >= 1 && K <= 4 && variantx_on(h, CA_VARX_Y_MFMA2, "CA_Y_MFMA2")) {
    h->ym_NT = cdiv(Nn, 16); h->ym_NS = cdiv(Nn, 64); h->ym_GS = cdiv(G, 64); h->ym_GT = cdiv(G, 16);
    uint8_t *yf = nullptr, *yb = nullptr;
    CACK(dalloc(h, &yf, h->ym_NT * h->ym_GS * 1024));
    CACK(dalloc(h, &yb, (int64_t)h->ym_GT * h->ym_NS * 1024));
    h->Yf = (uint4*)yf; h->Yb = (uint4*)yb;
    uint8_t *wq = nullptr, *pq = nullptr;
    CACK(dalloc(h, &wq, (int64_t)h->ym_GS * 1024));
    CACK(dalloc(h, &pq, h->ym_NS * 1024));
    h->Wq = (uint4*)wq; h->Pq = (uint4*)pq;
    CACK(dalloc(h, &h->ym_amax, 2));
    // waves: rows 64 TL cells per block, at least ~4 waves per CU; columns 8 gene tiles per block x cell slices
    h->ym_tl = (h->ym_NT / 4 >= 4 * h->n_cu) ? 4 : (h->ym_NT / 2 >= 4 * h->n_cu) ? 2 : 1;
    h->n_yw = cdiv(h->ym_NT, 4 * h->ym_tl);   // blocks of k_yw_mfma = partials of sum_n psi_n.(YW)_n
    CACK(dalloc(h, &h->yw_part, h->n_yw));
    const int gblocks = cdiv(h->ym_GT, 8);
    h->ym_csplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, cdiv(8 * (int64_t)h->n_cu, 4 * gblocks)), h->ym_NS / 4));
    h->ym_schunk = cdiv(h->ym_NS, h->ym_csplit);
    h->ym_csplit = cdiv(h->ym_NS, h->ym_schunk);
    CACK(dalloc(h, &h->ym_out, (int64_t)h->ym_csplit * h->ym_GT * 256));
    hipLaunchKernelGGL(k_tile_yf, dim3(cdiv(h->ym_NT * h->ym_GS * 64, CA_YM_TB)), dim3(CA_YM_TB), 0, h->stream, (const uint8_t*)h->Y, h->Yf, Nn, h->Gp,
                       h->ym_NT, h->ym_GS);
    hipLaunchKernelGGL(k_tile_yb, dim3((unsigned)h->ym_NS, cdiv(h->ym_GT, 4)), dim3(CA_YM_TB), 0, h->stream, (const uint8_t*)h->Y, h->Yb, Nn, h->Gp,
                       h->ym_GT, h->ym_NS);
    HIPCK(h, hipGetLastError());
    h->y_mfma = true;
    h->y_dev_bytes += (h->ym_NT * h->ym_GS + (int64_t)h->ym_GT * h->ym_NS) * 1024;
  }
  // ---- both products from ONE tiled copy (k_ys_mfma): K = 1, 1-byte storage
  // (the stream's buffer loads carry 32-bit byte offsets inside a strip: RS <= 512 cells x Gp bytes, and 16 N bytes of psi's image -- far inside 2^31 below these bounds)
  if (h->ystore == CA_YSTORE_U8 && K == 1 && !h->y_mfma && h->Gp < (1 << 21) && Nn < ((int64_t)1 << 26) && variant_on(h, CA_VAR_Y_MFMA1, "CA_Y_MFMA1")) {
    h->ys_N64 = (Nn + 63) / 64 * 64;
    h->ys_nseg = h->Gp / CA_YS_GW;                  // Gp is a multiple of 1024
    // strips of RS cells per wave: about one resident round of blocks (3 per CU), at least 64 cells
    h->ys_RS = 64;
    while (h->ys_RS < 512 && cdiv(Nn, 4 * h->ys_RS) * h->ys_nseg > 4 * h->n_cu) h->ys_RS *= 2;
    h->ys_nrg = cdiv(Nn, 4 * h->ys_RS);
    CACK(dalloc(h, &h->Ys, h->ys_N64 * h->Gp));
    uint8_t *wr = nullptr, *pr = nullptr;
    CACK(dalloc(h, &wr, (int64_t)(h->Gp / 64) * 1024));
    CACK(dalloc(h, &pr, (h->ys_N64 / 64) * 1024));
    h->Wr = (uint4*)wr; h->Pr = (uint4*)pr;
    CACK(dalloc(h, &h->Wsum, (int64_t)(h->Gp / 64) * 4));
    CACK(dalloc(h, &h->Psum, (h->ys_N64 / 64) * 4));
    CACK(dalloc(h, &h->YWpart, (int64_t)(std::max(h->nseg, h->ys_nseg) + 1) * Nn));      // (replaces the vector stream's smaller slabs)
    CACK(dalloc(h, &h->YTpart, (int64_t)(std::max(h->nrb, h->ys_nrg) + 1) * h->Gp));
    CACK(dalloc(h, &h->ys_exps, 6));
    CACK(dalloc(h, &h->ys_amax, 2));
    h->ys_nq = cdiv((int64_t)(h->Gp / 64) + h->ys_N64 / 64, CA_YM_TB / 64);
    h->ys_ncap = std::max(h->ys_nq, h->ngblk + cdiv(Nn, CA_TB)) + 2;
    CACK(dalloc(h, &h->ys_amaxp, (int64_t)3 * h->ys_ncap * 2));
    hipLaunchKernelGGL(k_bias_y, dim3(cdiv(h->ys_N64 * (h->Gp / 16), CA_YM_TB)), dim3(CA_YM_TB), 0, h->stream, (const uint8_t*)h->Y, (uint4*)h->Ys, Nn,
                       h->ys_N64, h->Gp);
    HIPCK(h, hipGetLastError());
    {
      // what ONE TF1-Adam step can add to a magnitude: lr_t |m| / sqrt(v) <= lr_t (1 - b1) / sqrt((1 - b2)(1 - b1^2 / b2)) (Cauchy-
      // Schwarz on the two moving averages), lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) maximised over t
      const double b1 = h->opt.beta1, b2 = h->opt.beta2, lr = h->opt.learning_rate;
      if (b1 >= 0 && b2 > 0 && b1 * b1 < b2 && b2 < 1 && lr > 0) {
        double sup = 0.0, p1 = 1.0, p2 = 1.0;
        for (int t = 1; t <= 200000; ++t) { p1 *= b1; p2 *= b2; sup = std::max(sup, std::sqrt(1.0 - p2) / (1.0 - p1)); }
        h->ys_step_bound = (float)(1.02 * lr * sup * (1.0 - b1) / std::sqrt((1.0 - b2) * (1.0 - b1 * b1 / b2)));
      }
    }
    h->y_ys = true;
    h->y_dev_bytes += h->ys_N64 * h->Gp;
  }
  // the Y stream rides on the forward sweep's launch: 1-byte storage, K = 1, the fused sweep with its default block shapes
  h->ride_ok = h->ystore == CA_YSTORE_U8 && K == 1 && h->fused_ok && !h->c16 && h->fwd_cell && (h->fc_tl == 6 || h->fc_tl == 8 || (h->fc_tl == 2 && h->fc_nbig == 0)) &&
               !h->y_mfma && !h->y_ys && variant_on(h, CA_VAR_Y_RIDE, "CA_Y_RIDE");
  constexpr bool kRideSeqDefault = false;
  h->ride_seq = h->ride_ok && variant_on(h, CA_VAR_RIDE_SEQ, "CA_RIDE_SEQ") && (kRideSeqDefault || variantx_on(h, CA_VARX_RIDE_SEQ, "CA_RIDE_SEQ_ON"));
  const bool tl1_ok = h->fc_tl == 1 && h->fc_nbig == 0 && !h->c16;
  h->ride_ys = h->y_ys && h->fused_ok && h->fwd_cell && (h->fc_tl == 6 || tl1_ok || (h->fc_tl == 2 && h->fc_nbig == 0)) &&
               variant_on(h, CA_VAR_Y_RIDE, "CA_Y_RIDE");
  h->yfin_split = h->ride_ys && h->bwd_mfma && h->tail_fuse && variant_on(h, CA_VAR_YFIN_RIDE, "CA_YFIN_RIDE");
  // The int8 one-copy stream either rides or runs in line: as a launch of its own on the SIDE stream (variant y_ride off at 4e7 counts and more) its parameter
  // images -- rewritten by the update launch -- are not ordered against that stream (tools/fuzz_large.py --wide: 22 165 x 4734 x 3, trace 6e-5 off).  Not a
  // default path (1-byte storage with K = 1 rides); the side stream stays for the vector streams of 2- / 4-byte storage and K != 1, which the large runs confirm.
  if (h->y_ys && !h->ride_ys) h->async_y = false;
  {   // balanced forward sweep (ca_fwdbal.hip.h): one to six whole tiles per CU, the int8 stream riding, eight clones at most, one latent dimension
    const int tiles = cdiv(Nn, 16), qb = tiles / std::max(h->n_cu, 1), rb = tiles - qb * h->n_cu;
    // ... and at least 96 k-steps of 32 genes: with eight waves per block a wave of cfg-2 (2000 genes, 63 k-steps) has eight k-steps, the two
    // pipeline fills of a block that also sweeps a chunk cost more than the balance returns (10k x 2k x 4: 42.7 against 39.3 us per iteration)
    h->fwd_bal = h->ride_ys && !h->c16 && !h->s2 && D == 1 && C <= 8 && qb >= 1 && qb <= 6 && h->nk32 >= 96 && h->host_dev && variant_on(h, CA_VAR_FWD_BAL, "CA_FWD_BAL");
    if (h->fwd_bal) {
      // left-over tiles: gene chunks swept by the sweep blocks beside their own tiles (the default), or -- opt-in CA_VARX_BAL_TILES, bal_nchunk = 0 -- single-tile
      // blocks of their own with no exchange (measured, us per iteration at 12 500 / 25 000 / 10 240 / 14 336 cells, i.e. 14 / 27 / 128 / 128 left-over tiles:
      // chunks 63.6 / 95.1 / 57.3 / 68.0, tile blocks 63.2 / 96.7 / 60.6 / 72.0, four-wave sweep 65.0 / 98.7 / 58.5 / 69.3: level at few left-over tiles --
      // the stream's blocks go to the CUs without a tile block -- and slower at many; profiles/r05_small_shapes.txt)
      h->bal_q = qb; h->bal_r = rb;
      h->bal_nchunk = (rb > 0 && !variantx_on(h, CA_VARX_BAL_TILES, "CA_BAL_TILES")) ? std::min(CA_BAL_MAXCHUNK, h->n_cu / rb) : 0;
      CACK(dalloc(h, &h->bal_xw, std::max<int64_t>(1, (int64_t)rb * h->bal_nchunk * 512)));
    }
  }
  // mc_samples = 2, four draws per sweep: where the sweep is the cell kernel and the stream either rides as the int8 stream or not at all
  h->s2f = h->s2 && h->fused_ok && h->fwd_cell && (h->fc_tl == 2 || h->fc_tl == 6) && (h->ride_ys || !h->ride_ok) && variant_on(h, CA_VAR_S2_FUSE, "CA_S2_FUSE");
  h->off_g = 3 + C;
  h->off_y = h->off_g + (int64_t)G * (S + D);
  h->red_n = h->off_y + (int64_t)G * K;
  CACK(dalloc(h, &h->red, h->off_y + (int64_t)h->Gp * std::max(K, 1)));   // Y^T psi lands here directly ([Gp][K] rows, first G reduced)
  CACK(dalloc(h, &h->terms_dev, 4));
  CACK(ensure_elbo_cap(h, 64));
  CACK(ensure_eps_cap(h, 1));
  // ---- column sums and Y^T X via the streaming kernel with all-ones factors (exact for integer counts:
  //      every per-strip partial is an integer < 2^24, the cross-strip sum is fp64)
  {
    // colsum: run ypass with K' = 1, psi == 1, W == 0 on temporary factor buffers
    const int cols = 1 + ((K > 0) ? P : 0);
    // loc0 == NULL: one more column with factor 1 / rowMeans(Y) gives mu_guess_g = mean_n(y_ng / mean_g' y_ng') (:220-235,
    // data_init_mu = TRUE), and loc0 = safe_inverse_softplus(mu_guess) (:262, :6-11).  The weights go through float32 and
    // the strip sums are fp32 (1e-7 relative on an initial value).
    const int cols_all = cols + (p->loc0 ? 0 : 1);
    std::vector<double> srow;
    if (!p->loc0) {
      srow.resize((size_t)Nn);
      HIPCK(h, hipMemcpyAsync(srow.data(), h->s64, (size_t)Nn * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      SYNC(h);
    }
    std::vector<double> cs((size_t)G, 0.0), ytx((size_t)G * std::max(P, 1), 0.0);
    float *Ft = nullptr, *Vt = nullptr, *YWp = nullptr, *YTp = nullptr; double* yt = nullptr;
    HIPCK(h, hipMalloc((void**)&Ft, (size_t)Nn * sizeof(float)));
    HIPCK(h, hipMalloc((void**)&Vt, (size_t)G * sizeof(float)));
    HIPCK(h, hipMalloc((void**)&YWp, (size_t)h->nseg * Nn * sizeof(float)));
    HIPCK(h, hipMalloc((void**)&YTp, (size_t)h->nrb * h->Gp * sizeof(float)));
    HIPCK(h, hipMalloc((void**)&yt, (size_t)h->Gp * sizeof(double)));
    HIPCK(h, hipMemsetAsync(Vt, 0, (size_t)G * sizeof(float), h->stream));
    std::vector<float> col((size_t)Nn);
    for (int j = 0; j < cols_all; ++j) {
      if (j >= cols) for (int64_t n = 0; n < Nn; ++n) col[n] = (float)((double)G / srow[(size_t)n]);
      else for (int64_t n = 0; n < Nn; ++n) col[n] = j == 0 ? 1.f : (float)p->X[hidx(p->layout, n, j - 1, Nn, P)];
      HIPCK(h, hipMemcpyAsync(Ft, col.data(), (size_t)Nn * sizeof(float), hipMemcpyHostToDevice, h->stream));
      dim3 grid((unsigned)((int64_t)h->nrg * h->nseg));
      ca_ovf_args no_ovf;
      memset(&no_ovf, 0, sizeof(no_ovf));
      if (h->ystore == CA_YSTORE_U8)
        hipLaunchKernelGGL((k_ypass<uint8_t, 1>), grid, dim3(CA_TB), 0, h->stream, (const uint8_t*)h->Y, Ft, 1, Vt, 0, YWp, YTp, Nn, G, h->Gp, h->nseg, h->nrb, h->TR, 1, no_ovf, (int)grid.x);
      else if (h->ystore == CA_YSTORE_U16)
        hipLaunchKernelGGL((k_ypass<uint16_t, 1>), grid, dim3(CA_TB), 0, h->stream, (const uint16_t*)h->Y, Ft, 1, Vt, 0, YWp, YTp, Nn, G, h->Gp, h->nseg, h->nrb, h->TR, 1, no_ovf, (int)grid.x);
      else
        hipLaunchKernelGGL((k_ypass<float, 1>), grid, dim3(CA_TB), 0, h->stream, (const float*)h->Y, Ft, 1, Vt, 0, YWp, YTp, Nn, G, h->Gp, h->nseg, h->nrb, h->TR, 1, no_ovf, (int)grid.x);
      HIPCK(h, hipGetLastError());
      hipLaunchKernelGGL(k_colsum, dim3(cdiv(h->Gp, 64)), dim3(1024), 0, h->stream, YTp, yt, h->nrg, (int64_t)h->Gp, h->Gp);
      std::vector<double> tmp((size_t)G);
      HIPCK(h, hipMemcpyAsync(tmp.data(), yt, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      SYNC(h);
      for (int64_t e = 0; e < h->n_ovf; ++e)   // overflow list, fixed (cell, gene) order
        tmp[h->h_ocol[e]] += (double)h->h_oval[e] * (double)col[h->h_orow[e]];
      if (j >= cols) {
        if (h->opt.world > 1) h->mu_part = tmp;   // a shard: the guess over ALL cells is completed when the transport is set (setup_global_sums)
        std::vector<float> l0((size_t)G);
        for (int g = 0; g < G; ++g) {
          const double mu = tmp[g] / (double)Nn;
          l0[g] = (float)(std::log(1.0 - std::exp(-std::fabs(mu))) + std::max(mu, 0.0));
        }
        CACK(upload_f(h, h->loc, l0));
      } else if (j == 0) cs = tmp;
      else for (int g = 0; g < G; ++g) ytx[(size_t)g * P + (j - 1)] = tmp[g];
    }
    hipFree(Ft); hipFree(Vt); hipFree(YWp); hipFree(YTp); hipFree(yt);
    CACK(upload_d(h, h->colsum, cs));
    CACK(dalloc(h, &h->YtX, (int64_t)G * std::max(P, 1)));
    CACK(upload_d(h, h->YtX, ytx));
  }
  CACK(dalloc(h, &h->loc_init, G));
  HIPCK(h, hipMemcpyAsync(h->loc_init, h->loc, (size_t)G * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  h->dir_const = -((double)C * std::lgamma(1.0 / (double)C) - std::lgamma(1.0));
  h->b1p = (float)h->opt.beta1;
  h->b2p = (float)h->opt.beta2;
  CACK(refresh_derived(h));
  SYNC(h);
  return CA_OK;
}

struct ParamRef {
  float* f = nullptr; double* d = nullptr;
  int64_t rows = 0, cols = 0, stride = 0, off = 0;  // logical rows x cols; device element (r,c) at r*stride + off + c
  int xform = 0;  // 0 none, 1 softplus, 2 row softmax, 3 exp, 4 softmax (vector)
  bool matrix = false;
};

bool find_param(ca_engine* h, const std::string& n, bool grad, ParamRef& r) {
  const int64_t N = h->N; const int G = h->G, C = h->C, K = h->K, P = h->P, D = h->D;
  if (n == "loc" || n == "mu") { r.f = grad ? h->g_loc : h->loc; r.rows = G; r.cols = 1; r.stride = 1; r.xform = (n == "mu") ? 1 : 0; return !(grad && n == "mu"); }
  if (n == "ls") { r.f = grad ? h->g_ls : h->ls; r.rows = G; r.cols = 1; r.stride = 1; return true; }
  if (n == "gamma_logits" || n == "clone_probs") { r.f = grad ? h->dgl : h->glogit; r.rows = N; r.cols = C; r.stride = C; r.matrix = true; r.xform = (n == "clone_probs") ? 2 : 0; return !(grad && n == "clone_probs"); }
  if (n == "alpha_unconstr" || n == "alpha") { r.f = grad ? h->g_a : h->alpha_u; r.rows = C; r.cols = 1; r.stride = 1; r.xform = (n == "alpha") ? 4 : 0; return !(grad && n == "alpha"); }
  if (n == "v" || n == "chi") { r.f = grad ? h->g_v : h->vchi; r.rows = K; r.cols = 1; r.stride = 1; r.xform = (n == "chi") ? 3 : 0; return !(grad && n == "chi"); }
  if (n == "psi") { r.f = grad ? h->g_psi : h->F; r.rows = N; r.cols = K; r.stride = grad ? K : D; r.matrix = true; return true; }
  if (n == "W") { r.f = grad ? h->g_V : h->V; r.rows = G; r.cols = K; r.stride = D; r.matrix = true; return true; }
  if (n == "beta") { r.f = grad ? h->g_V : h->V; r.rows = G; r.cols = (D > 0) ? P : 0; r.stride = D; r.off = K; r.matrix = true; return true; }
  if (n == "s" && !grad) { r.d = h->s64; r.rows = N; r.cols = 1; r.stride = 1; return true; }
  return false;
}

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
  float *vchi, *vchi_alt, *alpha_u, *alpha_u_alt;
  void take(const ca_engine* h) {
    look_valid = h->look_valid; bwd_ready = h->bwd_ready; pre_valid = h->pre_valid; em_stale = h->em_stale; y_defer = h->y_defer;
    ys_quant_ready = h->ys_quant_ready; ycache_valid = h->ycache_valid; yfin_pending = h->yfin_pending; fold_now = h->fold_now;
    pre_A = h->pre_A; pre_B = h->pre_B; hint_A = h->hint_A; hint_B = h->hint_B; gaux_slot = h->gaux_slot;
    vmm_at_idx = h->vmm_at_idx; gaux_idx = h->gaux_idx; ys_steps = h->ys_steps;
    for (int i = 0; i < 3; ++i) ys_namax[i] = h->ys_namax[i];
    b1p = h->b1p; b2p = h->b2p; adam_steps = h->adam_steps;
    vchi = h->vchi; vchi_alt = h->vchi_alt; alpha_u = h->alpha_u; alpha_u_alt = h->alpha_u_alt;
  }
  void restore(ca_engine* h) const {
    h->look_valid = look_valid; h->bwd_ready = bwd_ready; h->pre_valid = pre_valid; h->em_stale = em_stale; h->y_defer = y_defer;
    h->ys_quant_ready = ys_quant_ready; h->ycache_valid = ycache_valid; h->yfin_pending = yfin_pending; h->fold_now = fold_now;
    h->pre_A = pre_A; h->pre_B = pre_B; h->hint_A = -1; h->hint_B = -1; h->gaux_slot = gaux_slot;
    h->vmm_at_idx = vmm_at_idx; h->gaux_idx = gaux_idx; h->ys_steps = ys_steps;
    for (int i = 0; i < 3; ++i) h->ys_namax[i] = ys_namax[i];
    h->b1p = b1p; h->b2p = b2p; h->adam_steps = adam_steps;
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
  i->fwd_mfma = (h->fused_ok && h->fwd_mfma) ? 1 : 0; i->bwd_mfma = h->bwd_mfma ? 1 : 0; i->fsplit = h->fsplit; i->fwd_cell = (h->fused_ok && h->fwd_cell) ? 1 : 0;
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

int ca_comm_unique_id(char id[128]) {
  if (!g_rccl.load()) { g_last_error = g_rccl.err; return CA_ERR_COMM; }
  ca_nccl_uid u;
  int rc = g_rccl.GetUniqueId(&u);
  if (rc != 0) { g_last_error = "ncclGetUniqueId failed"; return CA_ERR_COMM; }
  memcpy(id, u.internal, 128);
  return CA_OK;
}

int ca_comm_init(ca_handle h, const char id[128]) {
  if (!h || !id) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  CACK(comm_check(h));   // a peer-to-peer transport that timed out leaves the engine dead: no falling back on the same handle
  if (!g_rccl.load()) { h->err = g_rccl.err; return CA_ERR_COMM; }
  HIPCK(h, hipSetDevice(h->device));
  ca_nccl_uid u;
  memcpy(u.internal, id, 128);
  int rc = g_rccl.CommInitRank(&h->comm, h->opt.world, u, h->opt.rank);
  if (rc != 0) {
    h->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    h->comm = nullptr;
    return CA_ERR_COMM;
  }
  return setup_global_sums(h);
}

int ca_p2p_export(ca_handle h, char handle[CA_P2P_HANDLE_BYTES]) {
  if (!h || !handle) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  if (h->opt.world > CA_TB) { h->err = "peer-to-peer transport: at most " + std::to_string(CA_TB) + " ranks (one flag lane per rank)"; return CA_ERR_COMM; }
  if (!h->p2p) {
    ca_p2p* pp = new ca_p2p();
    const int W = h->opt.world;
    // room for everything one call reduces: the train pass's summands, the setup sums, the PCA / correlation packs
    pp->cap = std::max<int64_t>(std::max<int64_t>(h->red_n, (int64_t)h->G * (h->C + 2) + 64), 4096);
    pp->slab_bytes = (size_t)2 * W * pp->cap * 16;   // [parity 2][source W][cap] entries of 16 bytes: two halves of a double, each with the call's tag (k_p2p_allreduce)
    // Fine-grained memory or nothing: the slab is written by remote peers over xGMI and polled here, which ordinary
    // (coarse-grained) device memory does not keep coherent -- a stale flag would be a hang or a wrong sum.  The caller moves
    // on to RCCL when this fails.
    if (hipExtMallocWithFlags((void**)&pp->slab, pp->slab_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      delete pp;
      h->err = "peer-to-peer transport: fine-grained device memory unavailable (hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed)";
      return CA_ERR_COMM;
    }
    auto fail = [&](const std::string& m) { if (pp->err_host) hipHostFree(pp->err_host); if (pp->err_local) hipFree(pp->err_local); if (pp->peers_dev) hipFree(pp->peers_dev);
                                            hipFree(pp->slab); delete pp; h->err = m; return CA_ERR_HIP; };
    if (hipMemset(pp->slab, 0, pp->slab_bytes) != hipSuccess) return fail("hipMemset of the p2p slab failed");
    if (hipMalloc((void**)&pp->peers_dev, (size_t)W * sizeof(double*)) != hipSuccess) return fail("hipMalloc (p2p peer table) failed");
    if (hipMalloc((void**)&pp->err_local, sizeof(unsigned int)) != hipSuccess) return fail("hipMalloc (p2p error flag) failed");
    if (hipMemset(pp->err_local, 0, sizeof(unsigned int)) != hipSuccess) return fail("hipMemset (p2p error flag) failed");
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize (p2p setup) failed");   // (NULL-stream memsets are not ordered against the engine's non-blocking stream)
    if (hipHostMalloc((void**)&pp->err_host, sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return fail("hipHostMalloc (p2p error word) failed");
    *pp->err_host = 0ull;
    if (hipHostGetDevicePointer((void**)&pp->err_dev, pp->err_host, 0) != hipSuccess) return fail("hipHostGetDevicePointer (p2p error word) failed");
    const int ms = h->opt.comm_timeout_ms > 0 ? h->opt.comm_timeout_ms : 10000;
    pp->timeout_ticks = (unsigned long long)((double)ms * 1000.0 * h->ticks_per_us);   // s_memrealtime ticks
    h->p2p = pp;
  }
  ca_p2p_wire w;
  memset(&w, 0, sizeof(w));
  HIPCK(h, hipIpcGetMemHandle(&w.mem, h->p2p->slab));
  w.cap = h->p2p->cap; w.rank = h->opt.rank; w.world = h->opt.world; w.device = h->device; w.pid = (int32_t)getpid();
  w.local_ptr = (uint64_t)(uintptr_t)h->p2p->slab;
  memset(handle, 0, CA_P2P_HANDLE_BYTES);
  memcpy(handle, &w, sizeof(w));
  return CA_OK;
}

static void p2p_unmap(ca_p2p* pp) {
  for (void*& q : pp->opened) if (q) { hipIpcCloseMemHandle(q); q = nullptr; }
  (void)hipGetLastError();
  pp->mapped = false;
}

// Phase 1: map every peer's slab.  Touches no peer and launches nothing, so a rank whose peer failed does not end up waiting for
// it: the caller agrees on every rank's result over its control plane and then calls ca_p2p_commit on all ranks.
int ca_p2p_connect(ca_handle h, const char* handles) {
  if (!h || !handles) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  if (!h->p2p) { h->err = "ca_p2p_connect before ca_p2p_export"; return CA_ERR_STATE; }
  if (!variant_on(h, CA_VAR_P2P, "CA_P2P")) { h->err = "peer-to-peer transport switched off (CA_VAR_P2P)"; return CA_ERR_COMM; }
  HIPCK(h, hipSetDevice(h->device));
  ca_p2p* pp = h->p2p;
  if (pp->connected) { h->err = "ca_p2p_connect: the transport is already committed"; return CA_ERR_STATE; }
  const int W = h->opt.world;
  std::vector<double*> peers((size_t)W, nullptr);
  p2p_unmap(pp);
  pp->opened.assign((size_t)W, nullptr);
  auto fail = [&](const std::string& m) { p2p_unmap(pp); h->err = m; return CA_ERR_COMM; };
  for (int r = 0; r < W; ++r) {
    ca_p2p_wire w;
    memcpy(&w, handles + (size_t)r * CA_P2P_HANDLE_BYTES, sizeof(w));
    if (w.rank != r || w.world != W || w.cap != pp->cap)
      return fail("p2p handle " + std::to_string(r) + " does not match this problem (rank / world / payload size)");
    if (r == h->opt.rank) { peers[r] = pp->slab; continue; }
    if (w.device != h->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, h->device, w.device) != hipSuccess || !can) {
        (void)hipGetLastError();
        return fail("no peer access from device " + std::to_string(h->device) + " to device " + std::to_string(w.device));
      }
      const hipError_t e = hipDeviceEnablePeerAccess(w.device, 0);
      (void)hipGetLastError();
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
    }
    if (w.pid == (int32_t)getpid()) {   // a handle of THIS process (one R session driving several devices): the slab's own address
      // ... but not two ranks of one process on ONE device: the runtime's device-wide synchronising calls (hipFree, hipMalloc of
      // a grown eps buffer, ...) made for one handle wait for every kernel on the device, also the other handle's all-reduce
      // kernel -- which waits for this rank.  Measured: the second all-reduce of such a pair ran into the device-side time
      // limit (tests/test_gpu_sharding.py).  Separate processes sharing a device are fine (their runtimes do not see each other).
      if (w.device == h->device && !(h->opt.variant_on & CA_VARX_P2P_SAME_DEVICE))
        return fail("peer-to-peer transport: ranks " + std::to_string(h->opt.rank) + " and " + std::to_string(r) + " are handles of one process on one "
                    "device; use one rank per device (or one process per rank)");
      peers[r] = (double*)(uintptr_t)w.local_ptr;
      continue;
    }
    void* q = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&q, w.mem, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
    }
    pp->opened[r] = q;
    peers[r] = (double*)q;
  }
  if (hipMemcpy(pp->peers_dev, peers.data(), (size_t)W * sizeof(double*), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy (p2p peer table) failed");
  pp->mapped = true;
  return CA_OK;
}

// Phase 2, collective: all_ranks_ok = 1 only if ca_p2p_connect returned CA_OK on EVERY rank (the caller's control plane says
// so).  Then the transport becomes the engine's all-reduce and the setup sums are reduced -- the first call that waits for
// peers.  all_ranks_ok = 0: the mappings are dropped and the engine is left without a transport (next: ca_comm_init).
int ca_p2p_commit(ca_handle h, int32_t all_ranks_ok) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  if (!h->p2p) { h->err = "ca_p2p_commit before ca_p2p_export"; return CA_ERR_STATE; }
  HIPCK(h, hipSetDevice(h->device));
  ca_p2p* pp = h->p2p;
  if (!all_ranks_ok) { p2p_unmap(pp); pp->connected = false; return CA_OK; }
  if (!pp->mapped) { h->err = "ca_p2p_commit(1) without a successful ca_p2p_connect on this rank"; return CA_ERR_STATE; }
  pp->connected = true;
  return setup_global_sums(h);
}

// Times n_calls all-reduces of n_doubles doubles on one of the engine's device transports, back to back on the engine's stream
// (HIP events around the batch).  Collective: every rank calls it with the same arguments.  The buffer is scratch.
int ca_comm_benchmark(ca_handle h, int32_t transport, int32_t n_calls, int64_t n_doubles, double* us_per_call) {
  if (!h || !us_per_call || n_calls < 1 || n_doubles < 1) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  const bool want_p2p = transport == CA_TRANSPORT_P2P;
  if (want_p2p && !(h->p2p && h->p2p->connected)) { h->err = "ca_comm_benchmark: no committed peer-to-peer transport"; return CA_ERR_STATE; }
  if (transport == CA_TRANSPORT_RCCL && !h->comm) { h->err = "ca_comm_benchmark: no RCCL communicator (ca_comm_init)"; return CA_ERR_STATE; }
  if (!want_p2p && transport != CA_TRANSPORT_RCCL) { h->err = "ca_comm_benchmark: transport must be CA_TRANSPORT_P2P or CA_TRANSPORT_RCCL"; return CA_ERR_INVALID; }
  double* buf = nullptr;
  HIPCK(h, hipMalloc((void**)&buf, (size_t)n_doubles * sizeof(double)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto cleanup = [&]() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); hipFree(buf); };
  int rc = CA_OK;
  // the RCCL leg runs with the peer-to-peer transport hidden from allreduce()
  const bool was = h->p2p && h->p2p->connected;
  if (!want_p2p && was) h->p2p->connected = false;
  auto run = [&]() -> int {
    HIPCK(h, hipMemsetAsync(buf, 0, (size_t)n_doubles * sizeof(double), h->stream));
    HIPCK(h, hipEventCreate(&e0)); HIPCK(h, hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CACK(allreduce(h, buf, n_doubles));   // warm-up (RCCL builds its channels on first use)
    HIPCK(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < n_calls; ++i) CACK(allreduce(h, buf, n_doubles));
    HIPCK(h, hipEventRecord(e1, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    CACK(comm_check(h));
    float ms = 0.f;
    HIPCK(h, hipEventElapsedTime(&ms, e0, e1));
    *us_per_call = (double)ms * 1e3 / n_calls;
    return CA_OK;
  };
  rc = run();
  if (!want_p2p && was) h->p2p->connected = true;
  cleanup();
  return rc;
}

int ca_comm_selftest(ca_handle h, int32_t n_rounds, int64_t n_doubles, int64_t* n_bad) {
  if (!h || !n_bad || n_rounds < 1 || n_doubles < 1) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  *n_bad = 0;
  // Round 5: the transport is asked what the loop asks of it, not six equal calls.  Call sizes ALTERNATE -- the train pass's
  // payload, a monitor pass's 3 + C doubles, and a vector longer than the peer-to-peer inbox (several pieces per call) -- in BURSTS of
  // back-to-back calls with no host synchronisation between them (both inbox parities and the sequence tags under the loop's own timing:
  // a fast rank one call ahead of a slow one), every sum checked; and, on the peer-to-peer transport, the RIDE form of the call (the
  // backward sweep's slabs folded and a block-partial sum added inside the all-reduce's launch, ca_p2p_args) against the same sums made
  // on the host.  Summands are small integers and halves times (rank + 1): every partial sum is exact in a double in any order.
  const double W = (double)std::max(h->opt.world, 1), tri = W * (W + 1.0) / 2.0;
  const bool p2p = h->p2p && h->p2p->connected;
  const int64_t big = p2p ? 2 * h->p2p->cap + 17 : 2 * n_doubles + 17;
  const int64_t sizes[3] = {n_doubles, std::min<int64_t>(n_doubles, 11), big};
  constexpr int BURST = 8;
  int64_t slot = 0;
  for (int64_t sz : sizes) slot = std::max(slot, sz);
  double* buf = nullptr;
  HIPCK(h, hipMalloc((void**)&buf, (size_t)(slot * BURST) * sizeof(double)));
  std::vector<double> host((size_t)(slot * BURST));
  auto pattern = [](int64_t i, int r) { return (double)((i * 7 + (int64_t)r * 13) % 251 + 1); };
  int64_t checked = 0;
  auto run = [&]() -> int {
    for (int r0 = 0; r0 < n_rounds; r0 += BURST) {
      const int nb = std::min(BURST, n_rounds - r0);
      for (int j = 0; j < nb; ++j) {
        const int r = r0 + j;
        const int64_t sz = sizes[r % 3];
        for (int64_t i = 0; i < sz; ++i) host[(size_t)(j * slot + i)] = (double)(h->opt.rank + 1) * pattern(i, r) + 0.5 * r;
      }
      HIPCK(h, hipMemcpyAsync(buf, host.data(), (size_t)(slot * nb) * sizeof(double), hipMemcpyHostToDevice, h->stream));
      for (int j = 0; j < nb; ++j) CACK(allreduce(h, buf + j * slot, sizes[(r0 + j) % 3]));     // back to back, no host in between
      HIPCK(h, hipMemcpyAsync(host.data(), buf, (size_t)(slot * nb) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      SYNC(h);
      for (int j = 0; j < nb; ++j) {
        const int r = r0 + j;
        const int64_t sz = sizes[r % 3];
        checked += sz;
        for (int64_t i = 0; i < sz; ++i)
          if (host[(size_t)(j * slot + i)] != tri * pattern(i, r) + W * 0.5 * r) *n_bad += 1;
      }
    }
    if (p2p && p2p_ride_ok(h, n_doubles) && n_doubles >= 8) {
      // ride form: entries [lo, lo + fn) are column sums of `ns` float slabs made here, entry `yi` also gets the sum of `ny` block partials
      const int64_t n = n_doubles, lo = std::min<int64_t>(3, n - 4), fn = n - lo - 1, yi = 0;
      const int ns = 5, ny = 37;
      std::vector<float> slabs((size_t)(ns * fn));
      std::vector<double> ywp((size_t)ny), base((size_t)n), want((size_t)n);
      for (int rep = 0; rep < 4; ++rep) {
        for (int sl = 0; sl < ns; ++sl) for (int64_t i = 0; i < fn; ++i) slabs[(size_t)(sl * fn + i)] = (float)((h->opt.rank + 1) * ((i + 3 * sl + rep) % 17));
        for (int b2 = 0; b2 < ny; ++b2) ywp[(size_t)b2] = (double)((h->opt.rank + 1) * ((b2 + rep) % 5)) * 0.5;
        for (int64_t i = 0; i < n; ++i) base[(size_t)i] = (double)(h->opt.rank + 1) * pattern(i, 100 + rep);
        for (int64_t i = 0; i < n; ++i) {   // what every rank contributes, then summed over ranks: (rank + 1) factors out -> tri
          double mine = (i >= lo && i < lo + fn) ? 0.0 : pattern(i, 100 + rep);
          if (i >= lo && i < lo + fn) for (int sl = 0; sl < ns; ++sl) mine += (double)(((i - lo) + 3 * sl + rep) % 17);
          if (i == yi) for (int b2 = 0; b2 < ny; ++b2) mine += 0.5 * (double)((b2 + rep) % 5);
          want[(size_t)i] = tri * mine;
        }
        float* gdev = nullptr; double* ydev = nullptr;
        HIPCK(h, hipMalloc((void**)&gdev, slabs.size() * sizeof(float)));
        if (hipMalloc((void**)&ydev, ywp.size() * sizeof(double)) != hipSuccess) { hipFree(gdev); h->err = "hipMalloc failed"; return CA_ERR_NOMEM; }
        hipMemcpyAsync(gdev, slabs.data(), slabs.size() * sizeof(float), hipMemcpyHostToDevice, h->stream);
        hipMemcpyAsync(ydev, ywp.data(), ywp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
        hipMemcpyAsync(buf, base.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream);
        ca_ar_ride ride;
        ride.gpart = gdev; ride.nslice = ns; ride.fold_lo = lo; ride.fold_n = fn; ride.yw_part = ydev; ride.n_yw = ny; ride.yw_index = yi;
        int rc2 = allreduce(h, buf, n, &ride);
        if (rc2 == CA_OK && hipMemcpyAsync(host.data(), buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc2 = CA_ERR_HIP;
        const hipError_t e2 = hipStreamSynchronize(h->stream);
        hipFree(gdev); hipFree(ydev);
        if (rc2 != CA_OK) return rc2;
        if (e2 != hipSuccess) { h->err = hipGetErrorString(e2); return CA_ERR_HIP; }
        CACK(comm_check(h));
        checked += n;
        for (int64_t i = 0; i < n; ++i) if (host[(size_t)i] != want[(size_t)i]) *n_bad += 1;
      }
    }
    return CA_OK;
  };
  const int rc = run();
  hipFree(buf);
  if (rc == CA_OK && *n_bad) h->err = "all-reduce known-answer test: " + std::to_string(*n_bad) + " of " + std::to_string(checked) + " sums are wrong";
  return rc;
}

int ca_set_host_allreduce(ca_handle h, ca_host_allreduce_fn fn, void* user) {
  if (!h || !fn) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  CACK(comm_check(h));
  HIPCK(h, hipSetDevice(h->device));
  h->host_ar = fn;
  h->host_ar_user = user;
  return setup_global_sums(h);
}

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

// host-side all-reduce of a small double vector through the engine's transport (device scratch round trip)
static int allreduce_host_vec(ca_engine* h, std::vector<double>& v, double* dev_scratch) {
  if (!is_sharded(h)) return CA_OK;
  HIPCK(h, hipMemcpyAsync(dev_scratch, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  CACK(allreduce(h, dev_scratch, (int64_t)v.size()));
  HIPCK(h, hipMemcpyAsync(v.data(), dev_scratch, v.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  SYNC(h);
  return CA_OK;
}

int ca_init_psi_pca(ca_handle h, const double* noise, int32_t n_iter, uint64_t seed, double* pcs_out) {
  if (!h) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  if (h->K == 0) return CA_OK;
  HIPCK(h, hipSetDevice(h->device));
  CACK(wait_y(h, true));
  const int64_t N = h->N; const int G = h->G, Gp = h->Gp, K = h->K;
  const int q = std::min(std::min(K + 4, 12), G);
  if (n_iter <= 0) n_iter = 40;
  float *Fp = nullptr, *Vp = nullptr, *YWp = nullptr, *YTp = nullptr, *csum = nullptr; double *ytd = nullptr, *cdev = nullptr;
  auto cleanup = [&]() { hipFree(Fp); hipFree(Vp); hipFree(YWp); hipFree(YTp); hipFree(csum); hipFree(ytd); hipFree(cdev); };
#define PCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return CA_ERR_HIP; } } while (0)
  PCK(hipMalloc((void**)&Fp, (size_t)N * q * sizeof(float)));
  PCK(hipMalloc((void**)&Vp, (size_t)Gp * q * sizeof(float)));
  PCK(hipMalloc((void**)&YWp, (size_t)(h->nseg + 1) * N * q * sizeof(float)));
  PCK(hipMalloc((void**)&YTp, (size_t)(h->nrb + 1) * Gp * q * sizeof(float)));
  PCK(hipMalloc((void**)&csum, (size_t)std::max(h->n_ovf_chunk, 1) * q * sizeof(float)));
  PCK(hipMalloc((void**)&ytd, (size_t)std::max<int64_t>((int64_t)Gp * q, 2 * (int64_t)Gp + q * q + 4 * q + 8) * sizeof(double)));
  PCK(hipMalloc((void**)&cdev, (size_t)q * sizeof(double)));
  const int nrb_tot = h->nrg + (h->n_ovf > 0 ? 1 : 0), nseg_tot = h->nseg + (h->n_ovf > 0 ? 1 : 0);
  auto colsum_to_host = [&](int qq, std::vector<double>& out) -> int {
    hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)Gp * qq, 64)), dim3(1024), 0, h->stream, YTp, ytd, nrb_tot, (int64_t)Gp * qq, Gp * qq);
    out.resize((size_t)G * qq);
    HIPCK(h, hipMemcpyAsync(out.data(), ytd, (size_t)G * qq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    return CA_OK;
  };
  int rc = CA_OK;
  // ---- column means and standard deviations of x = log2(y + 1)
  std::vector<double> sx, sxx, ntot(1, (double)N);
  {
    // float64 sums of x and x^2 per gene over the resident matrix (k_col_logstats), slices summed in order on the host; entries held
    // as 255 + overflow excess are put right from the host copy of the list (x = log2(256 + excess) where the dense byte gave 8)
    const int nsl = (int)std::max<int64_t>(1, std::min<int64_t>(128, N / 256));
    const int64_t rows_per = (N + nsl - 1) / nsl;
    double* part = nullptr;
    PCK(hipMalloc((void**)&part, (size_t)nsl * 2 * G * sizeof(double)));
    const dim3 grid(cdiv(G, CA_TB), nsl);
    if (h->ystore == CA_YSTORE_U8) hipLaunchKernelGGL((k_col_logstats<uint8_t>), grid, dim3(CA_TB), 0, h->stream, (const uint8_t*)h->Y, N, G, Gp, rows_per, part);
    else if (h->ystore == CA_YSTORE_U16) hipLaunchKernelGGL((k_col_logstats<uint16_t>), grid, dim3(CA_TB), 0, h->stream, (const uint16_t*)h->Y, N, G, Gp, rows_per, part);
    else hipLaunchKernelGGL((k_col_logstats<float>), grid, dim3(CA_TB), 0, h->stream, (const float*)h->Y, N, G, Gp, rows_per, part);
    std::vector<double> hp((size_t)nsl * 2 * G);
    hipError_t e1 = hipGetLastError();
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(h->stream);
    hipFree(part);
    PCK(e1);
    sx.assign((size_t)G, 0.0); sxx.assign((size_t)G, 0.0);
    for (int sl = 0; sl < nsl; ++sl)
      for (int g = 0; g < G; ++g) { sx[g] += hp[((size_t)sl * 2 + 0) * G + g]; sxx[g] += hp[((size_t)sl * 2 + 1) * G + g]; }
    for (int64_t i = 0; i < h->n_ovf; ++i) {
      const double x = std::log2(256.0 + (double)h->h_oval[(size_t)i]);
      sx[h->h_ocol[(size_t)i]] += x - 8.0; sxx[h->h_ocol[(size_t)i]] += x * x - 64.0;
    }
    std::vector<double> pack((size_t)2 * G + 1);
    for (int g = 0; g < G; ++g) { pack[g] = sx[g]; pack[G + g] = sxx[g]; }
    pack[2 * G] = (double)N;
    if ((rc = allreduce_host_vec(h, pack, ytd)) != CA_OK) { cleanup(); return rc; }
    for (int g = 0; g < G; ++g) { sx[g] = pack[g]; sxx[g] = pack[G + g]; }
    ntot[0] = pack[2 * G];
  }
  const double Nt = ntot[0];
  std::vector<double> mean(G), sd(G);
  for (int g = 0; g < G; ++g) {
    mean[g] = sx[g] / Nt;
    const double var = (sxx[g] - Nt * mean[g] * mean[g]) / (Nt - 1.0);
    sd[g] = std::sqrt(var);
    if (!(var > 1e-12 * std::max(1.0, mean[g] * mean[g]))) {
      h->err = "cannot rescale a constant/zero column to unit variance";   // prcomp(scale. = TRUE) on a constant gene
      cleanup();
      return CA_ERR_INVALID;
    }
  }
  // ---- blocked subspace iteration on Xs^T Xs
  std::vector<double> Q((size_t)G * q);
  {
    std::vector<float> r((size_t)G * q);
    ca_philox::normal_draw(seed ^ 0x9E3779B97F4A7C15ull, 0, (int64_t)G * q, r.data());
    for (size_t i = 0; i < Q.size(); ++i) Q[i] = r[i];
    orthonormalize(Q, G, q);
  }
  std::vector<float> vp((size_t)Gp * q, 0.f);
  std::vector<double> c(q), B;
  auto rows_pass = [&]() -> int {   // A = Xs Q  ->  Fp
    for (int k = 0; k < q; ++k) c[k] = 0.0;
    for (int g = 0; g < G; ++g)
      for (int k = 0; k < q; ++k) {
        const double w = Q[(size_t)g * q + k] / sd[g];
        vp[(size_t)g * q + k] = (float)w;
        c[k] += mean[g] * w;
      }
    HIPCK(h, hipMemcpyAsync(Vp, vp.data(), vp.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCK(h, hipMemcpyAsync(cdev, c.data(), q * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CACK(ypass_tf<1>(h, Fp, Vp, q, YWp, YTp, csum));
    hipLaunchKernelGGL(k_pca_rows, dim3(cdiv(N * q, CA_TB)), dim3(CA_TB), 0, h->stream, YWp, cdev, Fp, N, q, nseg_tot);
    HIPCK(h, hipGetLastError());
    return CA_OK;
  };
  for (int it = 0; it < n_iter && rc == CA_OK; ++it) {
    if ((rc = rows_pass()) != CA_OK) break;
    if ((rc = ypass_tf<1>(h, Fp, Vp, q, YWp, YTp, csum)) != CA_OK) break;   // B = Xs^T A (column products)
    if ((rc = colsum_to_host(q, B)) != CA_OK) break;
    if ((rc = allreduce_host_vec(h, B, ytd)) != CA_OK) break;
    for (int g = 0; g < G; ++g)
      for (int k = 0; k < q; ++k) Q[(size_t)g * q + k] = B[(size_t)g * q + k] / sd[g];
    orthonormalize(Q, G, q);
  }
  if (rc != CA_OK) { cleanup(); return rc; }
  // ---- Rayleigh-Ritz on the converged subspace
  if ((rc = rows_pass()) != CA_OK) { cleanup(); return rc; }
  std::vector<float> Af((size_t)N * q);
  PCK(hipMemcpyAsync(Af.data(), Fp, Af.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  PCK(hipStreamSynchronize(h->stream));
  std::vector<double> T((size_t)q * q, 0.0);
  for (int64_t n = 0; n < N; ++n)
    for (int a = 0; a < q; ++a)
      for (int b = a; b < q; ++b) T[(size_t)a * q + b] += (double)Af[(size_t)n * q + a] * (double)Af[(size_t)n * q + b];
  for (int a = 0; a < q; ++a) for (int b = 0; b < a; ++b) T[(size_t)a * q + b] = T[(size_t)b * q + a];
  if ((rc = allreduce_host_vec(h, T, ytd)) != CA_OK) { cleanup(); return rc; }
  std::vector<double> lam, W;
  sym_eig(T, q, lam, W);
  std::vector<double> sgn(K, 1.0);
  for (int k = 0; k < K; ++k) {   // sign: loading of largest magnitude positive
    double best = 0.0, val = 1.0;
    for (int g = 0; g < G; ++g) {
      double vgk = 0.0;
      for (int j = 0; j < q; ++j) vgk += Q[(size_t)g * q + j] * W[(size_t)j * q + k];
      if (std::fabs(vgk) > best) { best = std::fabs(vgk); val = vgk; }
    }
    sgn[k] = val < 0 ? -1.0 : 1.0;
  }
  std::vector<double> sc((size_t)N * K), stat((size_t)2 * K, 0.0);
  for (int64_t n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      double v = 0.0;
      for (int j = 0; j < q; ++j) v += (double)Af[(size_t)n * q + j] * W[(size_t)j * q + k];
      v *= sgn[k];
      sc[(size_t)n * K + k] = v;
      stat[k] += v; stat[K + k] += v * v;
    }
  if ((rc = allreduce_host_vec(h, stat, ytd)) != CA_OK) { cleanup(); return rc; }
  std::vector<float> Fh;
  if ((rc = download_f(h, Fh, h->F, N * std::max(h->D, 1))) != CA_OK) { cleanup(); return rc; }
  for (int k = 0; k < K; ++k) {
    const double mu_ = stat[k] / Nt, sdk = std::sqrt((stat[K + k] - Nt * mu_ * mu_) / (Nt - 1.0));   // scale(pcs)
    for (int64_t n = 0; n < N; ++n) {
      double v = (sc[(size_t)n * K + k] - mu_) / sdk;
      if (noise) v += noise[hidx(h->layout, n, k, N, K)];
      if (pcs_out) pcs_out[hidx(h->layout, n, k, N, K)] = v;
      Fh[(size_t)n * h->D + k] = (float)v;
    }
  }
  rc = upload_f(h, h->F, Fh);
  cleanup();
  if (rc != CA_OK) return rc;
  CACK(refresh_derived(h));
  SYNC(h);
  return CA_OK;
#undef PCK
}

int ca_clone_gene_sums(ca_handle h, const int32_t* clone_of_cell, double* Tout, double* Syy) {
  if (!h || !clone_of_cell || !Tout || !Syy) return CA_ERR_INVALID;
  CA_NOT_IN_RUN(h);
  HIPCK(h, hipSetDevice(h->device));
  CACK(wait_y(h, true));
  const int64_t N = h->N; const int G = h->G, Gp = h->Gp, C = h->C;
  float *Fp = nullptr, *Vp = nullptr, *YWp = nullptr, *YTp = nullptr, *csum = nullptr; double* ytd = nullptr;
  auto cleanup = [&]() { hipFree(Fp); hipFree(Vp); hipFree(YWp); hipFree(YTp); hipFree(csum); hipFree(ytd); };
#define PCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return CA_ERR_HIP; } } while (0)
  PCK(hipMalloc((void**)&Fp, (size_t)N * C * sizeof(float)));
  PCK(hipMalloc((void**)&Vp, (size_t)Gp * C * sizeof(float)));
  PCK(hipMalloc((void**)&YWp, (size_t)(h->nseg + 1) * N * C * sizeof(float)));
  PCK(hipMalloc((void**)&YTp, (size_t)(h->nrb + 1) * Gp * C * sizeof(float)));
  PCK(hipMalloc((void**)&csum, (size_t)std::max(h->n_ovf_chunk, 1) * C * sizeof(float)));
  PCK(hipMalloc((void**)&ytd, (size_t)Gp * C * sizeof(double)));
  PCK(hipMemsetAsync(Vp, 0, (size_t)Gp * C * sizeof(float), h->stream));
  const int nrb_tot = h->nrg + (h->n_ovf > 0 ? 1 : 0);
  std::vector<float> ind((size_t)N * C, 0.f), asg((size_t)N, 0.f);
  for (int64_t n = 0; n < N; ++n) {
    const int c = clone_of_cell[n];
    if (c >= C) { h->err = "clone index out of range"; cleanup(); return CA_ERR_INVALID; }
    if (c >= 0) { ind[(size_t)n * C + c] = 1.f; asg[n] = 1.f; }
  }
  int rc;
  std::vector<double> out;
  // T = Y^T I  (strip partials are exact integers below 2^24 for integer counts; the cross-strip sum is fp64)
  PCK(hipMemcpyAsync(Fp, ind.data(), ind.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if ((rc = ypass_tf<0>(h, Fp, Vp, C, YWp, YTp, csum)) != CA_OK) { cleanup(); return rc; }
  hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)Gp * C, 64)), dim3(1024), 0, h->stream, YTp, ytd, nrb_tot, (int64_t)Gp * C, Gp * C);
  out.resize((size_t)G * C);
  PCK(hipMemcpyAsync(out.data(), ytd, out.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PCK(hipStreamSynchronize(h->stream));
  for (int g = 0; g < G; ++g)
    for (int c = 0; c < C; ++c) Tout[hidx(h->layout, g, c, G, C)] = out[(size_t)g * C + c];
  // Syy = (Y^2)^T 1_assigned
  PCK(hipMemcpyAsync(Fp, asg.data(), asg.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if ((rc = ypass_tf<3>(h, Fp, Vp, 1, YWp, YTp, csum)) != CA_OK) { cleanup(); return rc; }
  hipLaunchKernelGGL(k_colsum, dim3(cdiv((int64_t)Gp, 64)), dim3(1024), 0, h->stream, YTp, ytd, nrb_tot, (int64_t)Gp, Gp);
  PCK(hipMemcpyAsync(Syy, ytd, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PCK(hipStreamSynchronize(h->stream));
  cleanup();
  if (is_sharded(h)) {   // sharded: totals over all cells
    std::vector<double> pack((size_t)G * C + G);
    for (int g = 0; g < G; ++g) { for (int c = 0; c < C; ++c) pack[(size_t)g * C + c] = out[(size_t)g * C + c]; pack[(size_t)G * C + g] = Syy[g]; }
    double* scratch = nullptr;
    HIPCK(h, hipMalloc((void**)&scratch, pack.size() * sizeof(double)));
    rc = allreduce_host_vec(h, pack, scratch);
    hipFree(scratch);
    if (rc != CA_OK) return rc;
    for (int g = 0; g < G; ++g) { for (int c = 0; c < C; ++c) Tout[hidx(h->layout, g, c, G, C)] = pack[(size_t)g * C + c]; Syy[g] = pack[(size_t)G * C + g]; }
  }
  return CA_OK;
#undef PCK
}

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
