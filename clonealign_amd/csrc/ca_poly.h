// ca_poly.h -- internal interface between the engine (clonealign_hip.hip) and the series form of the cells x genes x clones reduction (ca_poly.hip).
// Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define CA_PL_R 20      // degree of a bin's Taylor piece
#define CA_PL_NB 32     // most gene bins: covers |x|max (vmax - vmin) <= 4 CA_PL_NB
#define CA_PL_A 2.0     // bound of |x| times half a bin's width

struct ca_poly_hdr { double vlo, delta, xmax; int nb, bad; };
struct ca_poly_ws {
  ca_poly_hdr* hdr; double *tabB, *partB, *tabQ, *Qpart;
  int n_cell_blocks, n_gene_blocks;
};

// does this shape take the series form?  (one exponent dimension, one MC sample, 3 .. 8 clones)
inline bool ca_poly_ok(int D, int S, int C) { return D == 1 && S == 1 && C >= 3 && C <= 8; }
size_t ca_poly_workspace_bytes(int G, int n_cell_blocks);
void ca_poly_bind(ca_poly_ws* w, void* base, int G, int n_cell_blocks);
// forward: the moments of both draws' M, then per cell Z (both draws), dZ/dx (train draw), the cell epilogue (cell_ptrs: a ca_cell_ptrs whose etamax2 is a
// zero vector), d/dF into dF[N] and the backward moments.  backward: red_g[g][0] = d/dmu, red_g[g][1] = d/dV (the sweep's share, as k_bwd_mfma + k_colsum leave it).
hipError_t ca_poly_forward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, const float* muA, const float* muB, const float* Lb, int G,
                           int64_t N, int C, int K, const void* cell_ptrs, const float* alpha_u, double* cell_part, float* dF,
                           unsigned int* bad_word /* mapped host word set to 1 when the exponent range needs more than CA_PL_NB bins, or NULL */);
hipError_t ca_poly_backward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* mu, const float* Lb, int G, int C, double* red_g);
