"""``preprocess_for_clonealign`` -- host mirror of R/preprocess.R:93-147.

One-shot O(N*G) filtering ahead of the fit.  Kept on the host for
now; it is needed to reproduce the reference's only recorded numeric run (the vignette,
docs/introduction_to_clonealign.html:746-819).
"""
import numpy as np

from .api import _parse_cnv, _parse_expression


def _mad(x, constant=1.4826):
    """stats::mad: constant * median(|x - median(x)|)."""
    x = np.asarray(x, dtype=np.float64)
    return constant * np.median(np.abs(x - np.median(x)))


def get_outlying_genes(Y, nmads):
    """R/preprocess.R:59-63."""
    gene_means = np.asarray(Y, dtype=np.float64).mean(0)
    md = _mad(gene_means)
    return gene_means > gene_means.mean() + nmads * md


def preprocess_for_clonealign(gene_expression_data, copy_number_data, min_counts_per_gene=20,
                              min_counts_per_cell=100, remove_outlying_genes=True, nmads=10,
                              max_copy_number=6, remove_genes_same_copy_number=True,
                              gene_names=None, cell_names=None):
    """Filter genes/cells exactly in the order of R/preprocess.R:114-139."""
    Y, gn = _parse_expression(gene_expression_data)
    L, _ = _parse_cnv(copy_number_data)
    G = Y.shape[1]
    if L.shape[0] != G:
        raise ValueError("copy_number_data must have same number of genes (rows) as gene_expression_data")
    genes = np.array(gene_names if gene_names is not None else (gn if gn is not None else np.arange(G)))
    cells = np.array(cell_names if cell_names is not None else np.arange(Y.shape[0]))

    def keep_genes(mask):
        nonlocal Y, L, genes
        Y, L, genes = Y[:, mask], L[mask, :], genes[mask]

    keep_genes(~(L.max(1) > max_copy_number))                       # :114-116
    keep_genes(Y.sum(0) > min_counts_per_gene)                      # :118-120
    if remove_outlying_genes:                                       # :123-128
        keep_genes(~get_outlying_genes(Y, nmads))
    if remove_genes_same_copy_number:                               # :131-135
        keep_genes(~(L.var(1, ddof=1) == 0))
    cells_with_coverage = Y.sum(1) > min_counts_per_cell            # :138-139
    Y = Y[cells_with_coverage]
    cells = cells[cells_with_coverage]
    return {
        "gene_expression_data": Y,
        "copy_number_data": L,
        "retained_cells": cells,
        "retained_genes": genes,
    }
