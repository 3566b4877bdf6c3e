"""Cell-sharded data parallelism: host-side plan shared by bench.py, the engine and the tests.

Cells are conditionally independent given the global parameters, so rank r of W holds the contiguous
block ``cell_range(N, r, W)`` of cells together with everything indexed by cell (Y shard, A, c_n, s_n,
q(z) logits, psi and their Adam slots).  The gene-indexed parameters (loc, ls, W, beta), chi and alpha are
replicated; every rank applies the identical Adam update after ONE all-reduce per train pass of
``reduce_plan(...)`` doubles, plus a 3+C all-reduce per monitor pass and a one-off reduce of the per-gene
count totals at setup.  Restarts (run_clonealign) are independent replicas and use no collective.
"""


def cell_range(N, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; sizes differ by at most one cell."""
    if not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (N * rank) // world, (N * (rank + 1)) // world


def reduce_plan(G, C, K, P, S):
    """Layout of the summand buffer all-reduced per TRAIN pass (doubles), matching the engine's `red`:
    [EE_cell, Ep_cell, Eq_cell | sum_n gamma_nc (C) | per gene: d/dmu (S) then d/dV (D) | Y^T psi (G*K)]."""
    D = K + P if K > 0 else 0
    off_g = 3 + C
    off_y = off_g + G * (S + D)
    return {"cell_terms": (0, 3), "sum_gamma": (3, C), "gene": (off_g, G * (S + D)), "ytpsi": (off_y, G * K),
            "total": off_y + G * K, "monitor_total": 3 + C}
