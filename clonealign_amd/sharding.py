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


def reduce_plan(G, C, K, P, S, series=False, world=1):
    """Layout of the summand buffer all-reduced per TRAIN pass (doubles), matching the engine's `red`:
    [EE_cell, Ep_cell, Eq_cell | sum_n gamma_nc (C) | per gene: d/dmu (S) then d/dV (D) | Y^T psi (G*K)].

    ``series=True`` (round 6: an engine whose shape can take the series form of the contraction, ``ca_info.fwd_series``; K = 1, P = 0, S = 1): the gene sums are
    the LAST part and the backward moments and every rank's max |psi| sit in front of them,
    [cell terms | sum gamma | Y^T psi (G rounded up to 1024) | Q (32 bins x 22 moments x 8 clones) | max |psi| per rank (world) | gene sums]:
    a pass on the series form reduces the prefix up to the gene sums (``series_total`` doubles), a pass that falls back to the sweeps all of it."""
    D = K + P if K > 0 else 0
    if series:
        Gp = (G + 1023) // 1024 * 1024
        off_y = 3 + C
        off_q = off_y + Gp * K
        off_x = off_q + 32 * 22 * 8
        off_g = off_x + max(world, 1)
        return {"cell_terms": (0, 3), "sum_gamma": (3, C), "ytpsi": (off_y, G * K), "moments": (off_q, 32 * 22 * 8), "max_psi": (off_x, max(world, 1)),
                "gene": (off_g, G * (S + D)), "total": off_g + G * (S + D), "series_total": off_g, "monitor_total": 3 + C}
    off_g = 3 + C
    off_y = off_g + G * (S + D)
    return {"cell_terms": (0, 3), "sum_gamma": (3, C), "gene": (off_g, G * (S + D)), "ytpsi": (off_y, G * K),
            "total": off_y + G * K, "monitor_total": 3 + C}
