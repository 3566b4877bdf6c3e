// Balanced forward sweep for SMALL problems (round 5): one sweep block per CU, eight waves each, the left-over tiles spread over
// the blocks gene-wise with their partial Z exchanged through L2-bypassing tagged words.  Included at the end of ca_kernels.hip.h.
//
// Why (profiles/r05_small_shapes.txt).  The fused forward sweep (k_fwd_cell_mix_ys, ca_fwd_cell_body) gives a block 16 TL cells for
// ALL genes, four waves taking every fourth 32-gene k-step.  Its instruction stream is vector-issue bound, and
// on this part a SIMD issues vector instructions at full rate only with TWO OR MORE waves to pick from (tools/inst_lab.hip: v_fma 2.1
// cycles with two waves, 4.1 alone).  A shard of 12 500 cells is 782 tiles of 16 cells = 391 blocks of 32 cells over 256 CUs: 135 CUs
// hold two blocks and run at the issue limit (0.45 us per k-step for four tiles), 121 hold one block and idle every other issue slot
// (0.35 us per k-step for two tiles), and the launch is as long as the two-block CUs: 17.7-18.9 us of k-loop for work that is 13.0 us
// of issue time spread evenly over 1024 SIMDs.  Every shape below ~28k cells sits somewhere on this staircase
// (gpurun_out/r5/stair_5000.txt: 26.8 us at 2.00 tiles per CU, 32.2 at 3.00, 35.6 at 4.00).
//
// What.  tiles = q n_cu + r.  Sweep block b (b < n_cu, dispatched first: one lands on every CU) owns q whole tiles -- cells
// [16 q b, 16 q (b + 1)) -- for all genes, EIGHT waves taking every eighth k-step: two waves on every SIMD, equal work.  The r
// left-over tiles are cut gene-wise into `nchunk` chunks each (r nchunk <= n_cu); block g = j nchunk + c sweeps chunk c of left-over
// tile j FIRST (sixteen cells, a few k-steps, all eight waves), leaves the chunk's partial Z -- 256 doubles -- in a workspace as
// 512 words of (32 bits of the double | this launch's 32-bit tag), written and read with device-scope atomics (they bypass the
// XCDs' non-coherent L2s; a word whose tag is the launch's is complete by itself: no fence, no flag, no counter -- the transport of
// k_p2p_allreduce, between blocks instead of between GPUs), then its own tiles; the block holding a tile's LAST chunk (the highest
// block index among the tile's contributors, so everything it waits for was dispatched before it: no residency assumption) adds
// the tile's chunks in chunk order (fp64, fixed) and runs the cell epilogue for those sixteen cells together with its own.  The
// count-matrix stream's blocks follow in the same grid as before (their four waves; waves 4-7 of a stream block leave at once).
// Z is the same sum of products grouped differently (eight wave partials and up to sixteen chunks instead of four wave partials),
// so results differ from the four-wave kernel's in the last bits of an fp32 accumulation -- this path is held to the float64 oracle
// like the others (tests/test_gpu_parity.py), not bitwise to them.
#pragma once

#define CA_BAL_TB 512
#define CA_BAL_NW 8
#define CA_BAL_MAXCHUNK 16

struct ca_bal_args {
  int nb;                  // sweep blocks (= CUs)
  int r;                   // left-over tiles (tile index q nb + j)
  int nchunk;              // gene chunks per left-over tile (0 when r == 0)
  unsigned tag;            // this launch's tag, never 0 (the workspace starts zeroed)
  unsigned long long* xw;  // workspace [r][nchunk][512] tagged words
  unsigned long long timeout_ticks;   // bound of a consumer's wait for a chunk (s_memrealtime, 100 MHz)
  unsigned int* err;       // pinned host word: set when a wait ran out (the host reports CA_ERR_STATE at its next synchronisation)
  int stream_units;        // count-matrix stream units per 512-thread stream block: 1 (waves 4-7 leave at once) or 2 (waves 0-3 and 4-7 one unit each)
  int extra;               // > 0 (opt-in CA_VARX_BAL_TILES): the r left-over tiles are single-tile blocks of their own behind the sweep blocks (then nchunk == 0: no exchange)
};

// `span` k-steps of TL tiles, this wave taking virtual steps wv, wv + NW, ...; virtual step v is k-step k_lo + v below `gap_at` and
// k_lo + v + gap_len from there on (a block that has swept a chunk [k0, k1) of a left-over tile together with its own tiles goes on with
// the rest of its own: one range with a hole).  The fused two-draw contraction of ca_fwd_cell_body (same exp2 / bf16 hi-lo split / three
// MFMAs per tile and k-step); NS operand register sets in rotation, NS - 1 k-steps in flight.
template <int D, int TL, int NS>
__device__ __forceinline__ void ca_bal_sweep(const float (*f)[D], const float* em, const float* __restrict__ Vs, const uint4* __restrict__ Bq,
                                             int k_lo, int span, int gap_at, int gap_len, int wv, int lane, ca_f32x4* acc) {
  static_assert(NS == 3 || NS == 4, "three or four operand sets");
  const int q = lane >> 4;
  unsigned m0, m1;   // (-1, 0) and (0, -1) as bf16 pairs, see k_fwd_mfma
  asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  const ca_bf16x2 neg_lo = __builtin_bit_cast(ca_bf16x2, m0), neg_hi = __builtin_bit_cast(ca_bf16x2, m1);
  constexpr int NV4 = 2 * D;
  uint4 b1r[NS], b2r[NS];
  float4 vr[NS][NV4];
  auto fetch = [&](int set, int ks) {
    const uint4* bp = Bq + (int64_t)ks * 128;
    b1r[set] = bp[lane];
    b2r[set] = bp[64 + lane];
    const float4* vp = reinterpret_cast<const float4*>(Vs + ((int64_t)ks * 32 + 8 * q) * D);
#pragma unroll
    for (int i = 0; i < NV4; ++i) vr[set][i] = vp[i];
  };
  auto step = [&](int set) {
    const ca_bf16x8 B1 = __builtin_bit_cast(ca_bf16x8, b1r[set]), B2 = __builtin_bit_cast(ca_bf16x8, b2r[set]);
    auto vf = [&](int i) -> float { const float4& w = vr[set][i >> 2]; return (i & 3) == 0 ? w.x : (i & 3) == 1 ? w.y : (i & 3) == 2 ? w.z : w.w; };
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        ca_f32x2 eta = (ca_f32x2){vf((2 * pp) * D), vf((2 * pp + 1) * D)} * f[t][0] - em[t];
#pragma unroll
        for (int d = 1; d < D; ++d) eta = (ca_f32x2){vf((2 * pp) * D + d), vf((2 * pp + 1) * D + d)} * f[t][d] + eta;
        const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
        hi[pp] = ca_pk_bf16(e0, e1);
        const ca_bf16x2 hb = __builtin_bit_cast(ca_bf16x2, hi[pp]);
        const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_lo, e0, false);
        const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_hi, e1, false);
        lo[pp] = ca_pk_bf16(r0, r1);
      }
      const ca_bf16x8 A1 = __builtin_bit_cast(ca_bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
      const ca_bf16x8 A2 = __builtin_bit_cast(ca_bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
      ca_f32x4 a = acc[t];
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
      acc[t] = a;
    }
  };
  const int nkw = span > wv ? (span - wv + CA_BAL_NW - 1) / CA_BAL_NW : 0;   // this wave's k-steps
  if (nkw <= 0) return;   // (wave-uniform)
  auto kc = [&](int i) { const int v = wv + CA_BAL_NW * (i < nkw ? i : nkw - 1); return k_lo + (v < gap_at ? v : v + gap_len); };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) fetch(i, kc(i));
  const int ntrip = nkw / NS;
  for (int ti = 0; ti < ntrip; ++ti) {   // ONE basic block, no branch inside (see ca_fwd_cell_body on why)
    const int i0 = NS * ti;
#pragma unroll
    for (int u = 0; u < NS; ++u) {       // step u reads set u while the refill of set (u + NS - 1) % NS lands
      fetch((u + NS - 1) % NS, kc(i0 + u + NS - 1)); __builtin_amdgcn_sched_barrier(0); step(u); __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int rem = nkw - NS * ntrip;   // 0 .. NS - 1 k-steps left, their operands in sets 0, 1, ...
#pragma unroll
  for (int u = 0; u < NS - 1; ++u)
    if (rem > u) step(u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (refills past the end re-read the last k-step; they have landed before their registers mean anything else)
}

// the head of a sweep over `TL` tiles starting at cell `cell0`: latent positions and the exponent bound of lane j's cell of every tile
// (ca_fwd_cell_body's head; store = this block writes the bound where the epilogue and the backward sweep read it)
template <int D, int TL>
__device__ __forceinline__ void ca_bal_head(const float* __restrict__ F, const float* __restrict__ etamax2, const ca_cell_ptrs& p, int64_t N,
                                            int64_t cell0, int lane, int wv, bool store, float (*f)[D], float* em) {
  const int j = lane & 15, q = lane >> 4;
  float vmn[D], vmx[D];
  if (p.vmm_at) {
#pragma unroll
    for (int d = 0; d < D; ++d) { vmn[d] = ca_ord2f(p.vmm_at[d]); vmx[d] = ca_ord2f(p.vmm_at[8 + d]); }
  }
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const int64_t n = cell0 + 16 * t + j;
    const int64_t nn = n < N ? n : N - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) f[t][d] = F[nn * D + d];
    em[t] = p.vmm_at ? 0.f : etamax2[nn];
  }
  if (p.vmm_at) {
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int64_t n = cell0 + 16 * t + j;
      float e = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) e += fmaxf(f[t][d] * vmn[d], f[t][d] * vmx[d]);   // (k_etamax's arithmetic)
      em[t] = e;
      if (store && wv == 0 && q == 0 && n < N) p.etamax_w[n] = e;
    }
  }
}

__device__ __forceinline__ double ca_bal_block_sum(double v, double* sm /* >= CA_BAL_NW */) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sm[0];
#pragma unroll
  for (int w = 1; w < CA_BAL_NW; ++w) r += sm[w];
  return r;
}

template <int D, int TL, int DEPTH>
__global__ void __launch_bounds__(CA_BAL_TB, 2) k_fwd_bal_ys(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                             const float* __restrict__ Vs, const unsigned short* __restrict__ Mq, ca_cell_ptrs p,
                                                             const float* __restrict__ alpha_u, double* __restrict__ cell_part, int64_t N, int C,
                                                             int K, int nk, ca_bal_args ba, ca_ysride_args y) {
  constexpr size_t COMB = sizeof(ca_f32x4) * CA_BAL_NW * TL * 64;            // the eight waves' accumulators of the block's own tiles
  constexpr size_t COMBX = sizeof(ca_f32x4) * CA_BAL_NW * 64;                // ... and of the left-over chunk it sweeps
  constexpr size_t FW = COMB + COMBX + sizeof(double) * (256 + 64 + 64);     // + a left-over tile's summed Z, block-sum scratch, log alpha
  constexpr size_t SM = FW > 2 * (size_t)CA_YS_LDS_BYTES ? FW : 2 * (size_t)CA_YS_LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  if (p.gate) {   // queued ahead of the host's decision (ca_cell_ptrs::gate): anything but "go" and the launch does nothing
    if (*p.gate != p.gate_go) return;
  }
  const int b = (int)blockIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (b >= ba.nb + ba.extra) {   // ---- the count-matrix stream's blocks (and the overflow list's)
    // TWO units per block, waves 0-3 and waves 4-7 each on its own LDS region (the body's two barriers are the block's: both halves pass the
    // same two, or one half has left): two waves per SIMD beside the sweep block's two, as two 256-thread stream blocks were
    const int half = wv >> 2;
    const int sb = b - ba.nb - ba.extra, nsb = ba.stream_units == 2 ? (y.nb_main + 1) >> 1 : y.nb_main;
    if (sb >= nsb) {   // the overflow list's blocks, counted in 256 threads: its cell side, then its gene side
      if (half) return;
      const int ob = sb - nsb;
      if (ob < y.ovf.nb_rows) ca_ovf_rows_body(ob, y.ovf.rowptr, y.ovf.col, y.ovf.val, y.V, y.Df, y.ovf.YWextra, N, 1, 0, CA_TB);
      else ca_ovf_chunks_body(ob - y.ovf.nb_rows, y.ovf.chunk_start, y.ovf.row2, y.ovf.val2, y.F, y.Df, y.ovf.csum, y.ovf.nchunk, 1, 0);
      return;
    }
    if (ba.stream_units != 2 && half) return;
    const int idx = ba.stream_units == 2 ? 2 * sb + half : sb;
    if (idx >= y.nb_main) return;
    CA_PRIO_STREAM();
    ca_ys_mfma_body<DEPTH>(idx, y.Ys, y.io, N, y.Gp, y.RS, smem + (size_t)half * CA_YS_LDS_BYTES);
    return;
  }
  // ---- a sweep block
  ca_f32x4* comb = reinterpret_cast<ca_f32x4*>(smem);
  ca_f32x4* combx = reinterpret_cast<ca_f32x4*>(smem + COMB);
  double* zx = reinterpret_cast<double*>(smem + COMB + COMBX);   // [256]: the consumed tile's Z, entry = accumulator lane * 4 + r
  double* sm = zx + 256;                                        // [64]
  double* la = sm + 64;                                         // [64]
  const int lane = threadIdx.x & 63;
  const uint4* Bq = reinterpret_cast<const uint4*>(Mq);
  // log softmax(alpha): the last wave, as ca_log_softmax_alpha does it (C <= 8 here)
  if ((int)threadIdx.x >= CA_BAL_TB - 64) {
    const int c = lane;
    const double au = c < C ? (double)alpha_u[c] : -INFINITY;
    double mx = au;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    double se = c < C ? exp(au - mx) : 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) se += __shfl_xor(se, o, 64);
    if (c < C) la[c] = au - (mx + log(se));
  }
  const bool xtile = b >= ba.nb;                      // (block-uniform) a left-over tile as a single-tile block of its own
  const int64_t cellR = ((int64_t)TL * ba.nb) * 16;   // first left-over cell
  const int64_t cell0 = xtile ? cellR + 16 * (int64_t)(b - ba.nb) : (int64_t)b * (TL * 16);
  const int tls = xtile ? 1 : TL;                     // tiles per wave in the combine buffer
  int my_tile = -1, my_chunk = 0;
  if (xtile) {
    // 1x. ONE tile, all genes, eight waves: no exchange.  These blocks sit behind the sweep blocks in the grid and in front of the stream's: a
    //     CU that holds one beside its sweep block has no slot for a stream block meanwhile, so the dispatcher hands that CU fewer stream
    //     units -- the stream (a third of a sweep block's issue time at these sizes) is what evens the CUs out, not an exchange of partials.
    float f1[1][D], em1[1];
    ca_f32x4 a1[1] = {(ca_f32x4){0.f, 0.f, 0.f, 0.f}};
    ca_bal_head<D, 1>(F, etamax2, p, N, cell0, lane, wv, true, f1, em1);
    ca_bal_sweep<D, 1, 4>(f1, em1, Vs, Bq, 0, nk, nk, 0, wv, lane, a1);
    comb[wv * 64 + lane] = a1[0];
  } else {
  // 1. heads: the block's own TL tiles and -- tile index TL -- the left-over tile whose chunk it sweeps (one batch of loads)
  constexpr int NS = TL >= 5 ? 3 : 4;                 // (128 registers: two sweep waves and two stream waves per SIMD)
  if (ba.nchunk > 0 && b < ba.r * ba.nchunk) { my_tile = b / ba.nchunk; my_chunk = b - my_tile * ba.nchunk; }
  float f[TL + 1][D], em[TL + 1];
  ca_f32x4 acc[TL + 1];
#pragma unroll
  for (int t = 0; t <= TL; ++t) acc[t] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
  ca_bal_head<D, TL>(F, etamax2, p, N, cell0, lane, wv, true, f, em);
  if (my_tile >= 0) {   // (block-uniform)
    // 2a. the chunk's k-steps with TL + 1 tiles -- FIRST, so that the tile's consumer never waits -- and the chunk's partial Z on its way
    ca_bal_head<D, 1>(F, etamax2, p, N, cellR + 16 * (int64_t)my_tile, lane, wv, my_chunk == ba.nchunk - 1, f + TL, em + TL);
    const int k0 = (int)(((int64_t)nk * my_chunk) / ba.nchunk), k1 = (int)(((int64_t)nk * (my_chunk + 1)) / ba.nchunk);
    ca_bal_sweep<D, TL + 1, NS>(f, em, Vs, Bq, k0, k1 - k0, k1 - k0, 0, wv, lane, acc);
    combx[wv * 64 + lane] = acc[TL];
    __syncthreads();
    if (threadIdx.x < 256) {   // entry e = accumulator lane * 4 + r; eight wave partials in wave order, fp64
      const int e = (int)threadIdx.x, ln = e >> 2, r = e & 3;
      double z = 0.0;
#pragma unroll
      for (int w = 0; w < CA_BAL_NW; ++w) z += (double)combx[w * 64 + ln][r];
      const unsigned long long bits = (unsigned long long)__double_as_longlong(z), tg = (unsigned long long)ba.tag << 32;
      unsigned long long* dst = ba.xw + ((int64_t)my_tile * ba.nchunk + my_chunk) * 512 + 2 * e;
      __hip_atomic_store(dst, (bits & 0xFFFFFFFFull) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, (bits >> 32) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // 2b. the rest of the block's own tiles: every k-step but the chunk's
    ca_bal_sweep<D, TL, NS>(f, em, Vs, Bq, 0, nk - (k1 - k0), k0, k1 - k0, wv, lane, acc);
  } else {
    ca_bal_sweep<D, TL, NS>(f, em, Vs, Bq, 0, nk, nk, 0, wv, lane, acc);
  }
#pragma unroll
  for (int t = 0; t < TL; ++t) comb[(wv * TL + t) * 64 + lane] = acc[t];
  }
  // 3. the left-over tile this block finishes: its chunks, each word taken as soon as it can be read complete.  Thread e holds word e of
  //    every chunk (even e: the low half of double e / 2, odd e: the high half); the halves meet by a lane shuffle, the even lanes add the
  //    chunks in chunk order -- the same additions whoever arrived when
  const bool consume = !xtile && my_tile >= 0 && my_chunk == ba.nchunk - 1;
  if (consume) {
    const int e = (int)threadIdx.x;
    const unsigned long long* src = ba.xw + (int64_t)my_tile * ba.nchunk * 512 + e;
    unsigned long long w[CA_BAL_MAXCHUNK];
#pragma unroll
    for (int c = 0; c < CA_BAL_MAXCHUNK; ++c) w[c] = c < ba.nchunk ? __hip_atomic_load(src + (int64_t)c * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double z = 0.0;
#pragma unroll
    for (int c = 0; c < CA_BAL_MAXCHUNK; ++c) {
      if (c < ba.nchunk) {   // (uniform)
        while ((unsigned)(w[c] >> 32) != ba.tag) {
          if (__builtin_amdgcn_s_memrealtime() - t0 > ba.timeout_ticks) { __hip_atomic_store(ba.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
          __builtin_amdgcn_s_sleep(2);
          w[c] = __hip_atomic_load(src + (int64_t)c * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned mine = (unsigned)(w[c] & 0xFFFFFFFFull), other = (unsigned)__shfl_xor((int)mine, 1, 64);
        if ((e & 1) == 0) z += __longlong_as_double((long long)((unsigned long long)mine | ((unsigned long long)other << 32)));
      }
    }
    if ((e & 1) == 0) zx[e >> 1] = z;
  }
  __syncthreads();   // comb (all eight waves' accumulators) and zx are complete
  // 4. cell epilogue: the block's 16 TL cells and, for a consumer, the sixteen of its left-over tile -- 64 cells per pass of the 512 threads
  constexpr int CP = 8, CPB = CA_BAL_TB / CP;
  const int c = threadIdx.x % CP, cc = c < C ? c : C - 1;
  ca_cell_acc cacc = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int nown = tls * 16, ncell = nown + (consume ? 16 : 0);
  for (int g0 = 0; g0 < ncell; g0 += CPB) {
    const int lc = g0 + (int)threadIdx.x / CP;
    const bool inb = lc < ncell;
    const int lcc = inb ? lc : 0;
    double ZA, ZB;
    int64_t n;
    if (lcc < nown) {
      const int t = lcc >> 4, row = lcc & 15, qq = row >> 2, r = row & 3;
      const int la_ = 16 * qq + cc, lb_ = 16 * qq + C + cc;
      auto cz = [&](int w, int col) { return (double)comb[(w * tls + t) * 64 + col][r]; };
      ZA = ((cz(0, la_) + cz(1, la_)) + (cz(2, la_) + cz(3, la_))) + ((cz(4, la_) + cz(5, la_)) + (cz(6, la_) + cz(7, la_)));
      ZB = ((cz(0, lb_) + cz(1, lb_)) + (cz(2, lb_) + cz(3, lb_))) + ((cz(4, lb_) + cz(5, lb_)) + (cz(6, lb_) + cz(7, lb_)));
      n = cell0 + lcc;
    } else {
      const int row = lcc - nown, qq = row >> 2, r = row & 3;
      ZA = zx[(16 * qq + cc) * 4 + r];
      ZB = zx[(16 * qq + C + cc) * 4 + r];
      n = cellR + 16 * (int64_t)my_tile + row;
    }
    ca_cell_fused_group<CP>(p, la, inb ? n : N, N, C, D, K, ZA, ZB, cacc);
  }
  // 5. the block's partial sums of the ELBO (ca_cell_fused_finish's, over eight waves)
  const int W_ = 3 + C;
  {
    double v4[4] = {cacc.ee, cacc.pr, cacc.q, cacc.eeB};   // four block sums on one pair of barriers (wave butterflies, then the eight waves in order)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v4[i] += __shfl_xor(v4[i], o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sm[(threadIdx.x >> 6) * 4 + i] = v4[i];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double r = sm[threadIdx.x];
#pragma unroll
      for (int w = 1; w < CA_BAL_NW; ++w) r += sm[w * 4 + threadIdx.x];
      if (threadIdx.x < 3) cell_part[(int64_t)b * W_ + threadIdx.x] = r;
      else if (p.ee_partB) p.ee_partB[b] = r;
    }
  }
  double* smg = reinterpret_cast<double*>(smem);   // (comb is done with: the per-clone gamma sums take its place)
  __syncthreads();
  smg[threadIdx.x] = cacc.gsumc;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double a = 0.0;
    for (int i = 0; i < CPB; ++i) a += smg[i * CP + threadIdx.x];
    cell_part[(int64_t)b * W_ + 3 + threadIdx.x] = a;
  }
}
