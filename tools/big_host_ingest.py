"""Host-memory ingestion of matrices beyond 2^31 and 2^32 elements (R long vectors), both layouts, int32 and float64: exact library sizes after ca_create.  python tools/big_host_ingest.py (needs ~40 GB of host memory)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from clonealign_amd.engine import HipEngine
N, G = 900_000, 5000
rng = np.random.default_rng(3)
L = rng.integers(1, 5, size=(G, 4)).astype(float)
for dtype in (np.int32, np.float64):
    for layout in ("row", "col"):
        t0 = time.time()
        Y = rng.integers(0, 6, size=(N, G), dtype=np.int32) if layout == "row" else np.asfortranarray(rng.integers(0, 6, size=(G, N), dtype=np.int32).T)
        Y[-1, -1] = 300; Y[N // 2 + 7, 11] = 70000 if dtype != np.int32 or True else 1000
        Y = Y.astype(dtype, order="K")
        want = Y.sum(1, dtype=np.float64)
        t1 = time.time()
        eng = HipEngine(Y, L, rng.normal(size=(N, 1)), np.zeros(G) + 0.5, 1, layout=layout)
        t2 = time.time()
        s = eng.get("s")
        ok = np.array_equal(s, want)
        print(dtype.__name__, layout, "elements %.2e" % Y.size, "gen %.0f s, create %.1f s" % (t1 - t0, t2 - t1), "storage", eng.info()["y_storage_name"], "OK" if ok else "MISMATCH %d first %d" % ((s != want).sum(), np.flatnonzero(s != want)[0]), flush=True)
        eng.close()
        del Y
