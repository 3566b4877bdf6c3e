// ca_group.cpp -- ONE fit cell-sharded over several devices of ONE host process (include/clonealign_hip.h, "ca_group_*", ABI 6).
//
// The reference's caller is a single R session: clonealign() calls inference_tflow() once (R/clonealign.R:262-280) and takes the whole
// fit back (R/inference-tflow.R:424-480).  SURVEY.md section 8b puts the multi-GPU communicator inside that call ("one process / 8
// devices, communicator created per fit"), section 8e says how the work divides (cells local, gene parameters replicated, one
// all-reduce per train pass).  This file is that layer and nothing else: it is written against the PUBLIC C ABI only -- a group is
// W engine handles, W - 1 worker threads (the calling thread drives rank 0, so that a poll hook runs where R's API may be used) and
// the glue that slices the inputs, brings a transport up, keeps the ranks' decisions equal and gathers the outputs.  No kernel,
// no HIP call beyond what the engine handles make.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "clonealign_hip.h"

namespace {

thread_local std::string g_group_error;

// all W threads meet here; `abort` (a rank failed elsewhere) or a time limit releases everybody with `false`
struct SpinBarrier {
  std::atomic<int> count{0};
  std::atomic<int> gen{0};
  int n = 1;
  std::atomic<int>* abort = nullptr;
  int64_t timeout_ms = 600000;
  bool wait() {
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      count.store(0, std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
      return abort->load(std::memory_order_acquire) == 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0; gen.load(std::memory_order_acquire) == g; ++spin) {
      if (abort->load(std::memory_order_acquire)) return false;
      if (spin > 2000) {
        std::this_thread::yield();
        if ((spin & 1023) == 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
          abort->store(1, std::memory_order_release);
          return false;
        }
      }
    }
    return abort->load(std::memory_order_acquire) == 0;
  }
};

struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int(int)> task;
  bool has = false, quit = false, done = true;
  int rc = CA_OK;
};

struct RankCtx { struct ::ca_group* g; int rank; };   // what a rank's host all-reduce callback is handed

struct Shard {
  int64_t lo = 0, hi = 0;
  ca_problem p;
  std::vector<double> psi0, X, extra;
  std::vector<int64_t> ci;
};

}  // namespace

struct ca_group {
  int W = 0;
  int64_t N = 0;
  int32_t G = 0, C = 0, K = 0, P = 0, S = 0, layout = 0;
  std::vector<int32_t> devices;
  std::vector<ca_handle> h;
  std::vector<Shard> shard;
  std::vector<Worker*> workers;   // [1, W)
  ca_options opt;
  ca_group_info info;
  std::string err;
  bool dead = false;
  // host transport: every rank leaves its summand pointer, all meet, every rank adds the W buffers in rank order into its own scratch
  std::atomic<int> abort{0};
  SpinBarrier bar;
  std::vector<double*> ar_buf;
  std::vector<std::vector<double>> ar_tmp;
  std::vector<RankCtx> ctx;
  // the poll hook's decision at iteration i, made on rank 0 and followed by the others (ca_poll_fn contract: same decision on every rank)
  std::atomic<int64_t> decided{-1};   // highest iteration whose decision is published
  std::atomic<int> decision[64];
  ca_poll_fn user_poll = nullptr;
  void* user_ptr = nullptr;
};

namespace {

int host_allreduce_cb(void* user, double* buf, int64_t n) {
  RankCtx* c = static_cast<RankCtx*>(user);
  ca_group* g = c->g;
  const int r = c->rank, W = g->W;
  g->ar_buf[(size_t)r] = buf;
  if (!g->bar.wait()) return 1;
  std::vector<double>& t = g->ar_tmp[(size_t)r];
  if ((int64_t)t.size() < n) t.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) {   // rank order 0 .. W-1 on every rank: the replicas stay bit-identical
    double s = g->ar_buf[0][i];
    for (int q = 1; q < W; ++q) s += g->ar_buf[(size_t)q][i];
    t[(size_t)i] = s;
  }
  if (!g->bar.wait()) return 1;       // nobody overwrites its summands before everybody has read them
  memcpy(buf, t.data(), (size_t)n * sizeof(double));
  return 0;
}

void worker_main(Worker* w, int rank) {
  for (;;) {
    std::function<int(int)> task;
    {
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [&] { return w->has || w->quit; });
      if (w->quit) return;
      task = w->task;
      w->has = false;
    }
    int rc;
    try { rc = task(rank); } catch (...) { rc = CA_ERR_NOMEM; }
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->rc = rc;
      w->done = true;
    }
    w->cv.notify_all();
  }
}

// fn(rank) on every rank at once: ranks 1.. on their threads, rank 0 here.  Returns the ranks' codes.
std::vector<int> dispatch(ca_group* g, const std::function<int(int)>& fn) {
  std::vector<int> rc((size_t)g->W, CA_OK);
  // a rank that leaves with a real error releases the ranks that wait for it in the host reduction / for a poll decision
  const std::function<int(int)> guarded = [g, &fn](int r) {
    int c;
    try { c = fn(r); } catch (...) { c = CA_ERR_NOMEM; }
    if (c != CA_OK && c != CA_INTERRUPTED && c != CA_ERR_NAN && c != CA_ERR_INVALID) g->abort.store(1, std::memory_order_release);
    return c;
  };
  for (int r = 1; r < g->W; ++r) {
    Worker* w = g->workers[(size_t)r];
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->task = guarded; w->has = true; w->done = false;
    }
    w->cv.notify_all();
  }
  rc[0] = guarded(0);
  for (int r = 1; r < g->W; ++r) {
    Worker* w = g->workers[(size_t)r];
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    rc[(size_t)r] = w->rc;
  }
  return rc;
}

// One code for the group: CA_OK when every rank says so; the ranks' common code when they agree (NaN, interrupted); else the first
// real error, and the group is dead (some rank is out of step with the others).
int settle(ca_group* g, const std::vector<int>& rc, const char* what) {
  bool same = true;
  for (int r = 1; r < g->W; ++r) same = same && rc[(size_t)r] == rc[0];
  if (same && (rc[0] == CA_OK || rc[0] == CA_ERR_NAN || rc[0] == CA_INTERRUPTED || rc[0] == CA_ERR_INVALID)) {
    g->abort.store(0, std::memory_order_release);
    if (rc[0] != CA_OK) g->err = ca_last_error(g->h[0]);   // the engine's own words ("Initial elbo is NA", ...): the callers' messages are the one-handle ones
    return rc[0];
  }
  int first = 0;
  for (int r = 0; r < g->W; ++r)
    if (rc[(size_t)r] != CA_OK && rc[(size_t)r] != CA_INTERRUPTED && rc[(size_t)r] != CA_ERR_COMM) { first = r; break; }
    else if (rc[(size_t)r] != CA_OK && rc[(size_t)first] == CA_OK) first = r;
  g->err = std::string(what) + " (rank " + std::to_string(first) + " on device " + std::to_string(g->devices[(size_t)first]) + "): " +
           (g->h[(size_t)first] ? ca_last_error(g->h[(size_t)first]) : "no engine");
  g->dead = true;
  return rc[(size_t)first] != CA_OK ? rc[(size_t)first] : CA_ERR_STATE;
}

inline int64_t hidx(int layout, int64_t r, int64_t c, int64_t rows, int64_t cols) { return layout == CA_COL_MAJOR ? r + c * rows : r * cols + c; }

// rows [lo, hi) of an N x cols matrix in the problem's layout, as a compact (hi - lo) x cols matrix in the same layout
void slice_rows(const double* src, int layout, int64_t N, int64_t cols, int64_t lo, int64_t hi, std::vector<double>& dst) {
  const int64_t n = hi - lo;
  dst.resize((size_t)(n * cols));
  for (int64_t c = 0; c < cols; ++c)
    for (int64_t i = 0; i < n; ++i) dst[(size_t)hidx(layout, i, c, n, cols)] = src[hidx(layout, lo + i, c, N, cols)];
}

void scatter_rows(const double* src, int layout, int64_t N, int64_t cols, int64_t lo, int64_t hi, double* dst) {
  const int64_t n = hi - lo;
  for (int64_t c = 0; c < cols; ++c)
    for (int64_t i = 0; i < n; ++i) dst[hidx(layout, lo + i, c, N, cols)] = src[hidx(layout, i, c, n, cols)];
}

size_t dtype_bytes(int dt) { return dt == CA_F64 ? 8 : (dt == CA_F32 || dt == CA_I32) ? 4 : dt == CA_U16 ? 2 : 1; }

// the shard's view of the caller's problem: its rows of Y in place (y_ld), compact copies of the small per-cell inputs
void make_shard(const ca_problem& p, int rank, int W, Shard& s) {
  s.lo = (p.N * rank) / W; s.hi = (p.N * (rank + 1)) / W;
  const int64_t n = s.hi - s.lo;
  s.p = p;
  s.p.N = n;
  const bool sel = p.cell_index || p.gene_index;
  const int64_t Ns = sel ? p.N_src : p.N, Gs = sel ? (int64_t)p.G_src : (int64_t)p.G;
  const int64_t run = p.layout == CA_COL_MAJOR ? Ns : Gs, ld = p.y_ld > 0 ? p.y_ld : run;
  // source rows the shard needs: [r0, r1]
  const int64_t r0 = p.cell_index ? p.cell_index[s.lo] : s.lo, r1 = p.cell_index ? p.cell_index[s.hi - 1] : s.hi - 1;
  const int64_t step = p.layout == CA_COL_MAJOR ? 1 : ld;      // elements from one source row to the next
  s.p.Y = static_cast<const char*>(p.Y) + (size_t)(r0 * step) * dtype_bytes(p.y_dtype);
  s.p.y_ld = ld;
  if (sel) {
    s.p.N_src = r1 - r0 + 1;
    if (p.cell_index) {
      s.ci.resize((size_t)n);
      for (int64_t i = 0; i < n; ++i) s.ci[(size_t)i] = p.cell_index[s.lo + i] - r0;
      s.p.cell_index = s.ci.data();
    }
  }
  if (p.K > 0 && p.psi0) { slice_rows(p.psi0, p.layout, p.N, p.K, s.lo, s.hi, s.psi0); s.p.psi0 = s.psi0.data(); }
  if (p.P > 0 && p.X) { slice_rows(p.X, p.layout, p.N, p.P, s.lo, s.hi, s.X); s.p.X = s.X.data(); }
  if (p.extra_loglik) { slice_rows(p.extra_loglik, p.layout, p.N, p.C, s.lo, s.hi, s.extra); s.p.extra_loglik = s.extra.data(); }
}

void note(ca_group* g, const std::string& m) {
  std::string cur(g->info.note);
  if (!cur.empty()) cur += "; ";
  cur += m;
  strncpy(g->info.note, cur.c_str(), sizeof(g->info.note) - 1);
  g->info.note[sizeof(g->info.note) - 1] = 0;
}

void destroy_handles(ca_group* g) {
  for (int r = 1; r < g->W; ++r) if (!g->workers[(size_t)r]) return;   // (a group whose threads never all started holds no engine)
  dispatch(g, [g](int r) { if (g->h[(size_t)r]) { ca_destroy(g->h[(size_t)r]); g->h[(size_t)r] = nullptr; } return CA_OK; });
  g->abort.store(0, std::memory_order_release);
}

int create_handles(ca_group* g) {
  std::vector<std::string> errs((size_t)g->W);
  std::vector<int> rc = dispatch(g, [g, &errs](int r) {
    ca_options o = g->opt;
    o.device = g->devices[(size_t)r]; o.rank = r; o.world = g->W;
    const int c = ca_create(&g->shard[(size_t)r].p, &o, &g->h[(size_t)r]);
    if (c != CA_OK) errs[(size_t)r] = ca_last_error(nullptr);
    return c;
  });
  g->abort.store(0, std::memory_order_release);
  for (int r = 0; r < g->W; ++r)
    if (rc[(size_t)r] != CA_OK) {
      g->err = "ca_create (rank " + std::to_string(r) + " on device " + std::to_string(g->devices[(size_t)r]) + "): " + errs[(size_t)r];
      destroy_handles(g);
      return rc[(size_t)r];
    }
  return CA_OK;
}

constexpr int SELFTEST_ROUNDS = 12;

// known-answer all-reduces on the transport every rank now has; 0 = it adds what it should
int selftest(ca_group* g, std::string& why) {
  std::vector<int64_t> bad((size_t)g->W, 0);
  ca_info i0;
  ca_get_info(g->h[0], &i0);
  const int64_t n = i0.red_n;
  std::vector<int> rc = dispatch(g, [g, &bad, n](int r) { return ca_comm_selftest(g->h[(size_t)r], SELFTEST_ROUNDS, n, &bad[(size_t)r]); });
  g->abort.store(0, std::memory_order_release);
  for (int r = 0; r < g->W; ++r)
    if (rc[(size_t)r] != CA_OK || bad[(size_t)r] != 0) {
      why = "rank " + std::to_string(r) + ": " + (rc[(size_t)r] != CA_OK ? std::string(ca_last_error(g->h[(size_t)r])) : std::to_string(bad[(size_t)r]) + " wrong sums");
      return 1;
    }
  return 0;
}

bool devices_distinct(const ca_group* g) {
  for (int a = 0; a < g->W; ++a)
    for (int b = a + 1; b < g->W; ++b)
      if (g->devices[(size_t)a] == g->devices[(size_t)b]) return false;
  return true;
}

// returns 1 in use, -1 set-up failed (engines untouched: nothing was committed), -2 committed and failed (engines must be re-created)
int try_p2p(ca_group* g) {
  if (!devices_distinct(g) && !(g->opt.variant_on & CA_VARX_P2P_SAME_DEVICE)) { note(g, "peer-to-peer skipped: a device holds more than one rank"); return -1; }
  const int W = g->W;
  std::vector<char> handles((size_t)W * CA_P2P_HANDLE_BYTES, 0);
  std::vector<int> rc = dispatch(g, [g, &handles](int r) { return ca_p2p_export(g->h[(size_t)r], handles.data() + (size_t)r * CA_P2P_HANDLE_BYTES); });
  bool ok = true;
  std::string why;
  for (int r = 0; r < W; ++r) if (rc[(size_t)r] != CA_OK) { ok = false; if (why.empty()) why = ca_last_error(g->h[(size_t)r]); }
  if (ok) {
    rc = dispatch(g, [g, &handles](int r) { return ca_p2p_connect(g->h[(size_t)r], handles.data()); });
    for (int r = 0; r < W; ++r) if (rc[(size_t)r] != CA_OK) { ok = false; if (why.empty()) why = ca_last_error(g->h[(size_t)r]); }
  }
  g->abort.store(0, std::memory_order_release);
  if (!ok) {
    dispatch(g, [g](int r) { ca_p2p_commit(g->h[(size_t)r], 0); return CA_OK; });   // drop whatever was mapped; nothing has waited on a device yet
    g->abort.store(0, std::memory_order_release);
    note(g, "peer-to-peer set-up failed: " + why);
    return -1;
  }
  rc = dispatch(g, [g](int r) { return ca_p2p_commit(g->h[(size_t)r], 1); });       // the first call that waits for peers (setup sums)
  g->abort.store(0, std::memory_order_release);
  for (int r = 0; r < W; ++r) if (rc[(size_t)r] != CA_OK) { note(g, std::string("peer-to-peer commit failed: ") + ca_last_error(g->h[(size_t)r])); return -2; }
  if (selftest(g, why)) { note(g, "peer-to-peer failed its known-answer test: " + why); return -2; }
  return 1;
}

int try_rccl(ca_group* g) {
  if (!devices_distinct(g)) { note(g, "RCCL skipped: a device holds more than one rank"); return -1; }
  char id[128];
  if (ca_comm_unique_id(id) != CA_OK) { note(g, std::string("RCCL unavailable: ") + ca_last_error(nullptr)); return -1; }
  std::vector<int> rc = dispatch(g, [g, &id](int r) { return ca_comm_init(g->h[(size_t)r], id); });
  g->abort.store(0, std::memory_order_release);
  for (int r = 0; r < g->W; ++r) if (rc[(size_t)r] != CA_OK) { note(g, std::string("RCCL communicator failed: ") + ca_last_error(g->h[(size_t)r])); return -2; }
  std::string why;
  if (selftest(g, why)) { note(g, "RCCL failed its known-answer test: " + why); return -2; }
  return 1;
}

}  // namespace

static int use_host(ca_group* g) {
  std::vector<int> rc = dispatch(g, [g](int r) { return ca_set_host_allreduce(g->h[(size_t)r], host_allreduce_cb, &g->ctx[(size_t)r]); });
  const int s = settle(g, rc, "host transport");
  if (s != CA_OK) return s;
  std::string why;
  if (selftest(g, why)) { g->err = "host transport failed its known-answer test: " + why; g->dead = true; return CA_ERR_COMM; }
  return CA_OK;
}

extern "C" {

const char* ca_group_last_error(ca_group_handle g) { return g ? g->err.c_str() : g_group_error.c_str(); }

int ca_group_destroy(ca_group_handle g) {
  if (!g) return CA_OK;
  g->abort.store(0, std::memory_order_release);
  destroy_handles(g);
  for (int r = 1; r < g->W; ++r) {
    Worker* w = g->workers[(size_t)r];
    if (!w) continue;
    { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  delete g;
  return CA_OK;
}

int ca_group_create(const ca_problem* p, const ca_options* o, const int32_t* devices, int32_t n_devices, int32_t transport, ca_group_handle* out) {
  g_group_error.clear();
  auto bad = [&](const std::string& m, int code = CA_ERR_INVALID) { g_group_error = m; return code; };
  if (!p || !out || !devices) return bad("null argument");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return bad("1 to 64 devices");
  if (p->N < n_devices) return bad("fewer cells than devices");
  if (transport != 0 && transport != CA_TRANSPORT_RCCL && transport != CA_TRANSPORT_HOST && transport != CA_TRANSPORT_P2P) return bad("transport must be 0 (automatic) or a ca_transport");
  if (!p->Y || !p->L) return bad("Y and L are required");
  if (p->K > 0 && !p->psi0) return bad("psi0 is required when K > 0");
  if (p->P > 0 && !p->X) return bad("X is required when P > 0");
  if (p->cell_index) for (int64_t n = 1; n < p->N; ++n) if (p->cell_index[n] <= p->cell_index[n - 1]) return bad("cell_index must be strictly increasing and within [0, N_src)");
  if (p->y_on_device) for (int d = 1; d < n_devices; ++d) if (devices[d] != devices[0]) return bad("a device-resident count matrix can only be sharded over ranks of its own device; hand the matrix over from host memory");
  ca_group* g = nullptr;
  try {
    g = new ca_group();
    g->W = n_devices; g->N = p->N; g->G = p->G; g->C = p->C; g->K = p->K; g->P = p->P; g->S = p->S; g->layout = p->layout;
    g->devices.assign(devices, devices + n_devices);
    if (o) g->opt = *o; else ca_default_options(&g->opt);
    memset(&g->info, 0, sizeof(g->info));
    g->info.world = g->W; g->info.N = p->N;
    g->h.assign((size_t)g->W, nullptr);
    g->shard.resize((size_t)g->W);
    g->ar_buf.assign((size_t)g->W, nullptr);
    g->ar_tmp.resize((size_t)g->W);
    for (int r = 0; r < g->W; ++r) g->ctx.push_back(RankCtx{g, r});
    g->bar.n = g->W; g->bar.abort = &g->abort;
    for (auto& d : g->decision) d.store(0);
    g->workers.assign((size_t)g->W, nullptr);
    for (int r = 1; r < g->W; ++r) {
      g->workers[(size_t)r] = new Worker();
      g->workers[(size_t)r]->th = std::thread(worker_main, g->workers[(size_t)r], r);
    }
    for (int r = 0; r < g->W; ++r) make_shard(*p, r, g->W, g->shard[(size_t)r]);
  } catch (const std::exception& ex) {
    const std::string m = std::string("ca_group_create: ") + ex.what();
    if (g) ca_group_destroy(g);
    return bad(m, CA_ERR_NOMEM);
  }
  auto fail = [&](int code) { g_group_error = g->err; ca_group_destroy(g); return code; };
  int rc = create_handles(g);
  if (rc != CA_OK) return fail(rc);
  if (g->W == 1) { g->info.transport = CA_TRANSPORT_NONE; *out = g; return CA_OK; }
  // ---- transport chain: peer-to-peer by address -> RCCL -> host reduction; each must pass the known-answer test on every rank
  const bool want_p2p = transport == 0 || transport == CA_TRANSPORT_P2P, want_rccl = transport == 0 || transport == CA_TRANSPORT_RCCL;
  bool up = false;
  if (want_p2p) {
    const int st = try_p2p(g);
    g->info.p2p_status = st;
    if (st == 1) { g->info.transport = CA_TRANSPORT_P2P; up = true; }
    else if (transport == CA_TRANSPORT_P2P) { g->err = std::string("peer-to-peer transport: ") + g->info.note; return fail(CA_ERR_COMM); }
    else if (st == -2) { destroy_handles(g); g->info.rebuilds += 1; if ((rc = create_handles(g)) != CA_OK) return fail(rc); }
  }
  if (!up && want_rccl) {
    const int st = try_rccl(g);
    g->info.rccl_status = st;
    if (st == 1) { g->info.transport = CA_TRANSPORT_RCCL; up = true; }
    else if (transport == CA_TRANSPORT_RCCL) { g->err = std::string("RCCL transport: ") + g->info.note; return fail(CA_ERR_COMM); }
    else if (st == -2) { destroy_handles(g); g->info.rebuilds += 1; if ((rc = create_handles(g)) != CA_OK) return fail(rc); }
  }
  if (!up) {
    if ((rc = use_host(g)) != CA_OK) return fail(rc);
    g->info.transport = CA_TRANSPORT_HOST;
  }
  g->info.selftest_rounds = SELFTEST_ROUNDS;
  *out = g;
  return CA_OK;
}

int ca_group_get_info(ca_group_handle g, ca_group_info* info) {
  if (!g || !info) return CA_ERR_INVALID;
  *info = g->info;
  return CA_OK;
}

int ca_group_rank_handle(ca_group_handle g, int32_t rank, ca_handle* h) {
  if (!g || !h || rank < 0 || rank >= g->W) return CA_ERR_INVALID;
  *h = g->h[(size_t)rank];
  return CA_OK;
}

#define GROUP_ALIVE(g) do { if (!(g)) return CA_ERR_INVALID; if ((g)->dead) { (g)->err = "the group is dead (a rank failed or fell out of step earlier): destroy it"; return CA_ERR_STATE; } } while (0)

int ca_group_init_psi_pca(ca_group_handle g, const double* noise, int32_t n_iter, uint64_t seed, double* pcs_out) {
  GROUP_ALIVE(g);
  if (g->K == 0) return CA_OK;
  std::vector<std::vector<double>> nz((size_t)g->W), pcs((size_t)g->W);
  for (int r = 0; r < g->W; ++r) {
    const Shard& s = g->shard[(size_t)r];
    if (noise) slice_rows(noise, g->layout, g->N, g->K, s.lo, s.hi, nz[(size_t)r]);
    if (pcs_out) pcs[(size_t)r].resize((size_t)((s.hi - s.lo) * g->K));
  }
  std::vector<int> rc = dispatch(g, [&](int r) {
    return ca_init_psi_pca(g->h[(size_t)r], noise ? nz[(size_t)r].data() : nullptr, n_iter, seed, pcs_out ? pcs[(size_t)r].data() : nullptr);
  });
  const int s = settle(g, rc, "ca_init_psi_pca");
  if (s == CA_OK && pcs_out)
    for (int r = 0; r < g->W; ++r) scatter_rows(pcs[(size_t)r].data(), g->layout, g->N, g->K, g->shard[(size_t)r].lo, g->shard[(size_t)r].hi, pcs_out);
  return s;
}

int ca_group_gamma_init(ca_group_handle g, const float* eps) {
  GROUP_ALIVE(g);
  return settle(g, dispatch(g, [&](int r) { return ca_gamma_init(g->h[(size_t)r], eps); }), "ca_gamma_init");
}

int ca_group_elbo(ca_group_handle g, const float* eps, double* elbo) {
  GROUP_ALIVE(g);
  if (!elbo) return CA_ERR_INVALID;
  std::vector<double> v((size_t)g->W, 0.0);
  const int s = settle(g, dispatch(g, [&](int r) { return ca_elbo(g->h[(size_t)r], eps, &v[(size_t)r]); }), "ca_elbo");
  *elbo = v[0];
  return s;
}

int ca_group_step(ca_group_handle g, const float* eps) {
  GROUP_ALIVE(g);
  return settle(g, dispatch(g, [&](int r) { return ca_step(g->h[(size_t)r], eps); }), "ca_step");
}

// rank 0 asks the caller's hook (on the calling thread) and publishes the answer; the other ranks wait for it at the same iteration
static int poll_rank0(void* user, int32_t iter, double elbo) {
  ca_group* g = static_cast<ca_group*>(user);
  const int d = g->user_poll ? (g->user_poll(g->user_ptr, iter, elbo) != 0) : 0;
  g->decision[iter & 63].store(d, std::memory_order_relaxed);
  g->decided.store(iter, std::memory_order_release);
  return d;
}
static int poll_follow(void* user, int32_t iter, double elbo) {
  (void)elbo;
  ca_group* g = static_cast<ca_group*>(user);
  for (int spin = 0; g->decided.load(std::memory_order_acquire) < iter; ++spin) {
    if (g->abort.load(std::memory_order_acquire)) return 1;
    if (spin > 200) std::this_thread::yield();
  }
  return g->decision[iter & 63].load(std::memory_order_relaxed);
}

int ca_group_run_ex(ca_group_handle g, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws, double* trace, int32_t* n_elbo,
                    ca_poll_fn poll, void* user) {
  GROUP_ALIVE(g);
  if (!trace || !n_elbo || max_iter < 0) return CA_ERR_INVALID;
  g->user_poll = poll; g->user_ptr = user;
  g->decided.store(-1, std::memory_order_release);
  std::vector<std::vector<double>> tr((size_t)g->W);
  std::vector<int32_t> cnt((size_t)g->W, 0);
  for (int r = 1; r < g->W; ++r) tr[(size_t)r].assign((size_t)max_iter + 1, 0.0);
  std::vector<int> rc = dispatch(g, [&](int r) {
    // (a follower can never be more than one decision behind rank 0: its next ELBO needs rank 0's next all-reduce)
    return ca_run_ex(g->h[(size_t)r], max_iter, rel_tol, eps_stream, n_draws, r == 0 ? trace : tr[(size_t)r].data(), &cnt[(size_t)r],
                     poll ? (r == 0 ? poll_rank0 : poll_follow) : nullptr, g);
  });
  g->user_poll = nullptr; g->user_ptr = nullptr;
  *n_elbo = cnt[0];
  const int s = settle(g, rc, "ca_run");
  if (s == CA_OK || s == CA_INTERRUPTED)
    for (int r = 1; r < g->W; ++r)
      if (cnt[(size_t)r] != cnt[0] || memcmp(tr[(size_t)r].data(), trace, sizeof(double) * (size_t)cnt[0]) != 0) {
        g->err = "ca_run: rank " + std::to_string(r) + " saw another ELBO trace than rank 0 (replicas out of step)";
        g->dead = true;
        return CA_ERR_STATE;
      }
  return s;
}

int ca_group_iterate(ca_group_handle g, int32_t n_iter, const float* eps_stream, int64_t n_draws, double* last_elbo) {
  GROUP_ALIVE(g);
  std::vector<double> v((size_t)g->W, 0.0);
  const int s = settle(g, dispatch(g, [&](int r) { return ca_iterate(g->h[(size_t)r], n_iter, eps_stream, n_draws, last_elbo ? &v[(size_t)r] : nullptr); }), "ca_iterate");
  if (last_elbo) *last_elbo = v[0];
  return s;
}

int ca_group_final_elbo(ca_group_handle g, int32_t n_rep, const float* eps_stream, int64_t n_draws, double* values, double* mean, double* sd) {
  GROUP_ALIVE(g);
  if (n_rep < 1) return CA_ERR_INVALID;
  std::vector<std::vector<double>> v((size_t)g->W, std::vector<double>((size_t)n_rep, 0.0));
  std::vector<double> m((size_t)g->W, 0.0), d((size_t)g->W, 0.0);
  const int s = settle(g, dispatch(g, [&](int r) { return ca_final_elbo(g->h[(size_t)r], n_rep, eps_stream, n_draws, v[(size_t)r].data(), &m[(size_t)r], &d[(size_t)r]); }),
                       "ca_final_elbo");
  if (values) memcpy(values, v[0].data(), sizeof(double) * (size_t)n_rep);
  if (mean) *mean = m[0];
  if (sd) *sd = d[0];
  return s;
}

int ca_group_get_param(ca_group_handle g, const char* name, double* out) {
  GROUP_ALIVE(g);
  if (!name || !out) return CA_ERR_INVALID;
  const std::string n(name);
  int64_t cols = -1;   // >= 0: cell-indexed, that many columns
  if (n == "clone_probs" || n == "gamma_logits") cols = g->C;
  else if (n == "s") cols = 1;
  else if (n == "psi") cols = g->K;
  if (cols < 0) {   // replicated: rank 0's copy (bit-identical on every rank)
    const int c = ca_get_param(g->h[0], name, out);
    if (c != CA_OK) g->err = std::string("ca_get_param: ") + ca_last_error(g->h[0]);
    return c;
  }
  if (cols == 0) return CA_OK;
  std::vector<std::vector<double>> part((size_t)g->W);
  for (int r = 0; r < g->W; ++r) part[(size_t)r].resize((size_t)((g->shard[(size_t)r].hi - g->shard[(size_t)r].lo) * cols));
  const int s = settle(g, dispatch(g, [&](int r) { return ca_get_param(g->h[(size_t)r], name, part[(size_t)r].data()); }), "ca_get_param");
  if (s != CA_OK) return s;
  for (int r = 0; r < g->W; ++r) scatter_rows(part[(size_t)r].data(), g->layout, g->N, cols, g->shard[(size_t)r].lo, g->shard[(size_t)r].hi, out);
  return CA_OK;
}

int ca_group_reinit(ca_group_handle g, const double* psi0, const double* loc0) {
  GROUP_ALIVE(g);
  if (g->K > 0 && !psi0) return CA_ERR_INVALID;
  std::vector<std::vector<double>> ps((size_t)g->W);
  if (g->K > 0) for (int r = 0; r < g->W; ++r) slice_rows(psi0, g->layout, g->N, g->K, g->shard[(size_t)r].lo, g->shard[(size_t)r].hi, ps[(size_t)r]);
  return settle(g, dispatch(g, [&](int r) { return ca_reinit(g->h[(size_t)r], g->K > 0 ? ps[(size_t)r].data() : nullptr, loc0); }), "ca_reinit");
}

int ca_group_clone_gene_sums(ca_group_handle g, const int32_t* clone_of_cell, double* T, double* Syy) {
  GROUP_ALIVE(g);
  if (!clone_of_cell || !T || !Syy) return CA_ERR_INVALID;
  // every rank returns the totals over ALL cells (ca_clone_gene_sums all-reduces them); rank 0's copy is the caller's
  std::vector<std::vector<double>> t((size_t)g->W), y((size_t)g->W);
  for (int r = 1; r < g->W; ++r) { t[(size_t)r].resize((size_t)g->G * g->C); y[(size_t)r].resize((size_t)g->G); }
  return settle(g, dispatch(g, [&](int r) {
    return ca_clone_gene_sums(g->h[(size_t)r], clone_of_cell + g->shard[(size_t)r].lo, r == 0 ? T : t[(size_t)r].data(), r == 0 ? Syy : y[(size_t)r].data());
  }), "ca_clone_gene_sums");
}

}  // extern "C"
