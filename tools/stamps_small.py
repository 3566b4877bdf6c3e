"""Block timelines of the two small kernels of an iteration (k_final_gene, k_adam_cell) from per-block stamps: which kind of block
starts when and ends when.  Needs a lab build (-DCA_LAB):  tools/lab_stamps_small.sh [bench-like args]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = {0: "gene (+ prologue + W image when merged)", 1: "monitor tail", 2: "psi (+ psi image when merged)", 3: "adam_cell: gene prologue", 4: "small (chi/alpha)",
         5: "cell (q(z) logits)", 6: "adam_cell: quantiser"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=100_000)
    ap.add_argument("--genes", type=int, default=5_000)
    ap.add_argument("--clones", type=int, default=8)
    args = ap.parse_args()
    from clonealign_amd import engine as E
    import synth_data as synth
    from tests._cases import eps_for
    N, G, Cn = args.cells, args.genes, args.clones
    Yd, aux = synth.make_problem_torch(N, G, Cn, seed=20243, device="cuda:0")
    psi0 = np.random.default_rng(1).normal(size=(N, 1))
    loc0 = np.zeros(G) + 0.5
    eng = E.HipEngine(None, aux["L"], psi0, loc0, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G))
    eps = np.stack([eps_for(1, G, 10 + i) for i in range(80)])
    eng.iterate(40, eps)
    eng.synchronize()
    lib = E.load_library()
    nb = 4096
    buf = np.zeros((nb, 4), dtype=np.uint64)
    assert lib.ca_lab_read_stamps2(buf.ctypes.data_as(C.c_void_p), nb) == 0
    eng.close()
    checkpoints(buf)
    for lo, hi, name in ((0, 1024, "k_final_gene / k_update_merged"), (1024, 3072, "k_adam_cell")):
        b = buf[lo:hi]
        b = b[b[:, 3] == 1]
        if len(b) == 0:
            continue
        t0 = b[:, 0].min()
        start = (b[:, 0] - t0).astype(np.float64) / 100.0
        end = (b[:, 1] - t0).astype(np.float64) / 100.0
        kind = b[:, 2].astype(int)
        print(f"{name}: {len(b)} blocks, first start -> last end {end.max():.2f} us")
        for k in sorted(set(kind)):
            m = kind == k
            print(f"  {KINDS[k]:32s} n {m.sum():4d}  start min {start[m].min():5.2f} med {np.median(start[m]):5.2f} max {start[m].max():5.2f} | "
                  f"end med {np.median(end[m]):5.2f} max {end[m].max():5.2f} | duration med {np.median((end - start)[m]):5.2f} max {(end - start)[m].max():5.2f}")


def checkpoints(buf):
    cp = buf[3072:3072 + 8 * 64].reshape(64, 8, 4)
    for name, blocks in (("merged gene block (entry, after the Adam step + V' range, after the prologue, after the W image)", range(0, 20)),
                         ("monitor tail block (entry, done)", [40]), ("chi / alpha block (entry, done)", [41]),
                         ("first psi block (entry, after the step, after the psi image)", [42]), ("first logits block (entry, done)", [43])):
        rows = []
        for b in blocks:
            if cp[b, 0, 3] == 2:
                n = int(np.sum(cp[b, :, 3] == 2))
                t = cp[b, :n, 0].astype(np.float64)
                rows.append((t - t[0]) / 100.0)
        if rows:
            n = min(len(r) for r in rows)
            med = np.median(np.array([r[:n] for r in rows]), axis=0)
            print(f"  checkpoints, {name}:", " ".join("%.2f" % v for v in med))


if __name__ == "__main__":
    main()
