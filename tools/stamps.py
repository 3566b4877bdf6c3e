"""Timeline of ONE merged forward launch (k_fwd_cell_mix_y) from per-block stamps: when the sweep's and the stream's blocks
start and end, how many of each kind are resident over time, and how the stream's rate develops.  Needs a lab build:
    hipcc ... -DCA_LAB -o /tmp/lab_stamps.so ...;  CLONEALIGN_HIP_LIB=/tmp/lab_stamps.so python tools/stamps.py [bench-like args]
(tools/lab_stamps.sh builds and runs it on the GPU box)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=100_000)
    ap.add_argument("--genes", type=int, default=5_000)
    ap.add_argument("--clones", type=int, default=8)
    ap.add_argument("--tune", default="")
    ap.add_argument("--dump", default="")
    args = ap.parse_args()
    import torch
    from clonealign_amd import engine as E
    import synth_data as synth
    from tests._cases import eps_for
    N, G, Cn = args.cells, args.genes, args.clones
    Yd, aux = synth.make_problem_torch(N, G, Cn, seed=20243, device="cuda:0")
    psi0 = np.random.default_rng(1).normal(size=(N, 1))
    loc0 = np.zeros(G) + 0.5
    tune = {k: (v if ":" in v else int(v)) for k, v in (kv.split("=") for kv in args.tune.split(",") if kv)}
    eng = E.HipEngine(None, aux["L"], psi0, loc0, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G), tune=tune)
    eps = np.stack([eps_for(1, G, 10 + i) for i in range(12)])
    eng.iterate(6, eps)
    eng.synchronize()
    lib = E.load_library()
    nb = 8192
    buf = np.zeros((nb, 4), dtype=np.uint64)
    assert lib.ca_lab_read_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
    ph = np.zeros((2048, 4), dtype=np.uint64)
    if hasattr(lib, "ca_lab_read_stamps3") and lib.ca_lab_read_stamps3(ph.ctypes.data_as(C.c_void_p), 2048) == 0:
        # phases of the sweep blocks (wave 0): entry -> head done -> k-loop done -> past the combine barrier (-> block end from the block records)
        idx_of = {int(r[2] & np.uint64(0xFFFFFFFF)): r for r in buf[buf[:, 1] > 0] if int(r[2] >> np.uint64(32)) != 0}
        rows = []
        for i in range(2048):
            if ph[i, 0] > 0 and i in idx_of:
                r = idx_of[i]
                rows.append([(float(ph[i, 1]) - float(ph[i, 0])) / 100, (float(ph[i, 2]) - float(ph[i, 1])) / 100, (float(ph[i, 3]) - float(ph[i, 2])) / 100,
                             (float(r[1]) - float(ph[i, 3])) / 100, (float(r[1]) - float(r[0])) / 100])
        if rows:
            a = np.array(rows)
            for j, nm in enumerate(("head", "k-loop (wave 0)", "wait at the combine barrier", "cell epilogue", "whole block")):
                print(f"  sweep block phase {nm:28s} us: min {a[:, j].min():6.2f} med {np.median(a[:, j]):6.2f} p90 {np.percentile(a[:, j], 90):6.2f} max {a[:, j].max():6.2f}")
    eng.close()
    live = buf[:, 1] > 0
    b = buf[live]
    t0 = b[:, 0].min()
    start = (b[:, 0] - t0).astype(np.float64) / 100.0      # us (100 MHz)
    end = (b[:, 1] - t0).astype(np.float64) / 100.0
    kind = (b[:, 2] >> np.uint64(32)).astype(int)
    hw = (b[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (b[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"blocks {len(b)}: sweep big {np.sum(kind == 1)}, sweep small {np.sum(kind == 2)}, stream {np.sum(kind == 0)}; launch {end.max():.1f} us; "
          f"distinct CU ids {len(np.unique(cuid))}")
    for k, name in ((1, "sweep big"), (2, "sweep small"), (0, "stream")):
        m = kind == k
        if not m.any():
            continue
        d = end[m] - start[m]
        q = lambda a: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())  # noqa: E731
        print(f"  {name:11s} start [{q(start[m])}]  end [{q(end[m])}]  duration [{q(d)}]")
    edges = np.arange(0, end.max() + 10, 10.0)
    print("  t(us)   resident sweep / stream blocks at t   stream blocks finished in [t, t+10)   sweep finished")
    for t in edges:
        rs = int(np.sum((kind != 0) & (start <= t) & (end > t)))
        ry = int(np.sum((kind == 0) & (start <= t) & (end > t)))
        fy = int(np.sum((kind == 0) & (end >= t) & (end < t + 10)))
        fs = int(np.sum((kind != 0) & (end >= t) & (end < t + 10)))
        print(f"  {t:6.0f}   {rs:5d} / {ry:5d}      {fy:5d}      {fs:5d}")
    # stream block duration against its start time
    m = kind == 0
    for lo in range(0, int(end.max()), 20):
        mm = m & (start >= lo) & (start < lo + 20)
        if mm.any():
            print(f"  stream blocks started in [{lo},{lo + 20}) us: n {mm.sum():4d}  median duration {np.median((end - start)[mm]):6.1f} us")
    print("  per XCD (big sweep blocks): n, median start, median end, max end | blocks that stream first: median end | sweep first: median end")
    first = (((b[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 3) ^ ((b[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 8)) & 1
    for x in range(8):
        m = (kind == 1) & (xcc == x)
        if not m.any():
            continue
        f, s_ = m & (first == 1), m & (first == 0)
        print(f"    XCD {x}: n {m.sum():4d}  start {np.median(start[m]):6.1f}  end med {np.median(end[m]):6.1f} max {end[m].max():6.1f} | "
              f"stream-first {np.median(end[f]) if f.any() else float('nan'):6.1f} | sweep-first {np.median(end[s_]) if s_.any() else float('nan'):6.1f}"
              f" | stream blocks on it: {int(np.sum((kind == 0) & (xcc == x)))}")
    per_cu = np.array([np.sum((kind == 1) & (cuid == c)) for c in np.unique(cuid)])
    print(f"  big sweep blocks per CU over the launch: min {per_cu.min()} median {np.median(per_cu):.0f} max {per_cu.max()}")
    if args.dump:
        np.save(args.dump, buf[live])


if __name__ == "__main__":
    main()
