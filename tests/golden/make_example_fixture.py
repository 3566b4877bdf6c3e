#!/usr/bin/env python3
"""Dump the reference's bundled example *data* into a small committed fixture.

Reads ``/root/reference/data/example_sce.rda`` (a data file the reference's own
tests load: tests/testthat/test_clonealign.R:6,11) and writes
``tests/golden/example_sce.npz`` holding

  Y      int32 [200 cells, 100 genes]   = t(assay(example_sce, "counts"))   (R/clonealign.R:213)
  L      int32 [100 genes, 3 clones]    = rowData(example_sce)[, c("A","B","C")]
  genes  str   [100], cells str [200], clones str [3]

Run in the build container only (``/root/reference`` does not travel).  The data file itself is kept beside the fixture
(``tests/golden/example_sce.rda``: data the reference's own tests hold, not source) so that the reader can be tested anywhere:
tests/test_host_api.py::test_rdx2_reader_rederives_the_fixture_and_the_vignette_preprocessing.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import rdx2  # noqa: E402


def extract(ref):
    """(Y int32 [200, 100], L int32 [100, 3], genes, cells, clones) out of an example_sce.rda, every checksum of SURVEY.md section 7.2 asserted."""
    sce = rdx2.read_rda(ref)["example_sce"]
    data = sce.attr["assays"].attr[".xData"].value[".->data"]
    counts = data.attr["listData"].get("counts")
    M = rdx2.matrix(counts)  # genes x cells
    dn = counts.attr["dimnames"].value
    genes = list(dn[0].value)
    cells = list(dn[1].value)
    rd = sce.attr["rowRanges"].attr["elementMetadata"].attr["listData"]
    clones = rd.names()
    L = np.stack([rd.get(c).value for c in clones], axis=1).astype(np.int32)
    Y = M.T
    assert np.all(Y == np.round(Y)) and Y.min() >= 0
    Y = Y.astype(np.int32)
    # checksums recorded in SURVEY.md §7.2
    assert Y.shape == (200, 100) and L.shape == (100, 3)
    assert Y.sum() == 16090 and (Y > 0).sum() == 5845 and Y.max() == 163
    assert Y[10, 83] == 163
    assert list(Y.sum(1)[:4]) == [104, 67, 146, 73]
    assert L[:3].tolist() == [[1, 2, 2], [2, 1, 1], [3, 2, 2]]
    assert sce.attr["rowRanges"].attr["partitioning"].attr["NAMES"].value == genes
    assert sce.attr["colData"].attr["rownames"].value == cells
    return Y, L, genes, cells, clones


def main(ref="/root/reference/data/example_sce.rda"):
    Y, L, genes, cells, clones = extract(ref)
    out = os.path.join(HERE, "example_sce.npz")
    np.savez_compressed(out, Y=Y, L=L, genes=np.array(genes), cells=np.array(cells),
                        clones=np.array(clones))
    print("wrote", out, Y.shape, L.shape, genes[:3], cells[:3], clones)


if __name__ == "__main__":
    main(*sys.argv[1:])
