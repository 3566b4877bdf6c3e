"""``preprocess_for_clonealign`` -- host mirror of R/preprocess.R:93-147.

One-shot O(N*G) filtering ahead of the fit.  The two O(N*G) statistics (colSums, then rowSums
over the retained genes) can be taken on the device from the raw matrix (``on="device"``: ca_preprocess); the filtered
matrices the reference returns are then cut on the host with the masks.  The host form is what reproduces the
reference's only recorded numeric run (the vignette, docs/introduction_to_clonealign.html:746-819) and is the check
of the device form.
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
                              gene_names=None, cell_names=None, on="auto", device=0, return_masks=False):
    """Filter genes/cells exactly in the order of R/preprocess.R:114-139.

    ``on`` in {"auto", "host", "device"}: where colSums / rowSums are taken ("auto": on the device above 2e7 elements).
    ``return_masks=True`` (device statistics): no filtered copy of the count matrix is made -- the result carries
    ``keep_cells`` / ``keep_genes`` (boolean masks over the input) next to the filtered copy-number matrix and names, for
    ``clonealign(raw, result["copy_number_data"], cell_index=result["keep_cells"], gene_index=result["keep_genes"])``: the
    engine cuts the raw matrix at upload (ca_problem.cell_index / gene_index; the reference returns copies, :141-147)."""
    Y, gn = _parse_expression(gene_expression_data)
    L, _ = _parse_cnv(copy_number_data)
    G = Y.shape[1]
    if L.shape[0] != G:
        raise ValueError("copy_number_data must have same number of genes (rows) as gene_expression_data")
    genes = np.array(gene_names if gene_names is not None else (gn if gn is not None else np.arange(G)))
    cells = np.array(cell_names if cell_names is not None else np.arange(Y.shape[0]))
    if on not in ("auto", "host", "device"):
        raise ValueError("on must be 'auto', 'host' or 'device'")
    if on == "device" or (on == "auto" and Y.size > 20_000_000):
        from .engine import preprocess_masks
        kg, kc, _gs, _cs = preprocess_masks(Y, L, min_counts_per_gene, min_counts_per_cell, remove_outlying_genes, nmads,
                                            max_copy_number, remove_genes_same_copy_number, device=device)
        if return_masks:
            return {"keep_cells": kc, "keep_genes": kg, "copy_number_data": L[kg, :], "retained_cells": cells[kc],
                    "retained_genes": genes[kg]}
        return {
            "gene_expression_data": Y[np.ix_(kc, kg)],
            "copy_number_data": L[kg, :],
            "retained_cells": cells[kc],
            "retained_genes": genes[kg],
        }

    if return_masks:
        raise ValueError("return_masks=True needs the device statistics (on='device')")

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
