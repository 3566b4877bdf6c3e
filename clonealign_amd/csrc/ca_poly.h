// ca_poly.h -- internal interface between the engine (clonealign_hip.hip) and the series form of the cells x genes x clones reduction (ca_poly.hip).
// Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define CA_PL_R 20      // degree of a bin's Taylor piece
#define CA_PL_NB 32     // most gene bins: covers |x|max (vmax - vmin) <= 4 CA_PL_NB
#define CA_PL_A 2.0     // bound of |x| times half a bin's width
#ifndef CA_PL_NBL
#define CA_PL_NBL 4     // bins whose coefficient tables the cell kernel keeps in LDS
#endif

struct ca_poly_hdr { double vlo, delta, xmax; int nb, bad; };
struct ca_poly_ws {
  ca_poly_hdr* hdr; unsigned int* xbits; double *tabB, *partB, *tabQ, *Qpart;
  int n_cell_blocks, n_gene_blocks;
};

// does this shape take the series form?  (one exponent dimension, one MC sample, 3 .. 8 clones)
inline bool ca_poly_ok(int D, int S, int C) { return D == 1 && S == 1 && C >= 3 && C <= 8; }
size_t ca_poly_workspace_bytes(int G, int n_cell_blocks);
void ca_poly_bind(ca_poly_ws* w, void* base, int G, int n_cell_blocks);
// forward, two steps: (1) the moments of both draws' M over the gene bins (three small launches); (2) per cell Z (both draws), dZ/dx (train draw), the cell
// epilogue (cell_ptrs: a ca_cell_ptrs whose etamax2 is a zero vector), d/dF into dF[N] and the backward moments.  backward: red_g[g][0] = d/dmu,
// red_g[g][1] = d/dV (the sweep's share, as k_bwd_mfma + k_colsum leave it).
// (xpart / nx in either: max |x| of this state per piece of cells as the merged update that made the state left it, ca_merge_args::xpart -- no k_poly_xmax launch;
//  NULL / 0: the launch is made)
hipError_t ca_poly_moments(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, const float* muA, const float* muB, const float* Lb, int G,
                           int64_t N, int C, unsigned int* bad_word /* mapped host word set to 1 when the exponent range needs more than CA_PL_NB bins, or NULL */,
                           double* mirror /* mapped host slot {seq, max|x|, min v, max v} of this state's ranges, or NULL */, double seq, const float* xpart, int nx,
                           const double* xglob /* sharded: every rank's max |x| one Adam step back (ca_poly_xslot + the collective), or NULL */, int nglob, double xadd);
// the ranges alone (two tiny launches): a pass that takes the sweeps keeps the host's picture current with it
hipError_t ca_poly_ranges(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, int G, int64_t N, double* mirror, double seq, const float* xpart, int nx,
                          const double* xglob, int nglob, double xadd);
// sharded: this rank's max |x| of the current state into slots[rank], zeros into the other world - 1 slots (they are summed by the fit's collective)
hipError_t ca_poly_xslot(hipStream_t st, const float* xpart, int nx, const float* F, int64_t N, double* slots, int rank, int world);
// can the series form cover a state whose ranges were (xmax, vlo, vhi) `steps` Adam steps ago, none of which moved a variable by more than `step_bound`?
inline bool ca_poly_covers(double xmax, double vlo, double vhi, int steps, double step_bound) {
  const double x = xmax + steps * step_bound, wdt = (vhi - vlo) + 2.0 * steps * step_bound;
  return x == x && wdt == wdt && x * wdt <= 2.0 * CA_PL_A * CA_PL_NB;   // nb = ceil(x w / (2 a)) <= NB
}
hipError_t ca_poly_cells(hipStream_t st, const ca_poly_ws* w, int64_t N, int C, int K, const void* cell_ptrs, const float* alpha_u, double* cell_part, float* dF,
                         const void* yfin_args /* a ca_yfin_args: the count-matrix stream's finishing sums as extra blocks of the cell launch, or NULL */,
                         const void* local_tail /* cell-sharded: the pending monitor pass's ca_small_args with reduce_only (its local block sums ride on the moments'
                                                   reduction launch), or NULL */,
                         const float* xs_part, int xs_n, const float* xs_F, double* xs_slots /* cell-sharded: this rank's max |x| slot rides there too, or NULL */,
                         int rank, int world);
hipError_t ca_poly_backward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* mu, const float* Lb, int G, int C, double* red_g,
                            const void* small_tail /* a ca_small_args (pending monitor tail, run by an extra block) or NULL */);
