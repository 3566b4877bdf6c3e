"""The engine sequence of tests/test_gpu_parity.py::test_riding_dispatch_order_does_not_change_a_single_bit[shape1], repeated: which engine gives which result.
   python tools/repro_flake2.py [loops] [extra variant_off,...]"""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clonealign_amd.engine import HipEngine  # noqa: E402
from tests._cases import eps_for, make_case  # noqa: E402

loops = int(sys.argv[1]) if len(sys.argv) > 1 else 6
extra = tuple(v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if v)
case = make_case(seed=77, N=40_100, G=1100, C=8, K=1)
rng = np.random.default_rng(3)
idx = rng.integers(0, case["Y"].size, size=max(3, case["Y"].size // 5000))
case["Y"].reshape(-1)[idx] += rng.integers(200, 900, size=idx.size)
G = case["Y"].shape[1]
epss = np.stack([eps_for(1, G, 300 + i) for i in range(10)])
res = Counter()
state_hashes = {}
for loop in range(loops):
    for stream in ("int8", "vector"):
        for pat in (None, "1:1", "3:2", "16:8", "1:200", "255:1", -3, -64, "seq", "mixed", "fin"):
            if (stream == "int8" and pat in ("seq", "mixed")) or (stream == "vector" and pat == "fin"):
                continue
            voff = (() if stream == "int8" else ("y_mfma1",)) + extra
            kw = (dict(variant_on=("ride_seq",), variant_off=voff) if pat == "seq" else dict(variant_off=voff + ("ride_seq",)) if pat == "mixed" else
                  dict(variant_off=voff + ("yfin_ride",)) if pat == "fin" else dict(variant_off=voff, tune=({} if pat is None else {"ride_pattern": pat})))
            eng = HipEngine(**case, **kw)
            try:
                eng.gamma_init(eps_for(1, G, 0))
                last = eng.iterate(5, epss)
                st = eng.get_state()
            finally:
                eng.close()
            key = (stream, str(pat), round(last, 3))
            res[key] += 1
            import hashlib
            hs = hashlib.sha1(b"".join(np.ascontiguousarray(st[n]).tobytes() for n in sorted(st))).hexdigest()[:12]
            state_hashes.setdefault(stream, Counter())[hs] += 1
            if abs(last + 132399942.666) > 1.0:
                print("DEVIATION loop", loop, stream, pat, last, "state hash", hs, "(the stream's hashes so far:", dict(state_hashes[stream]), ")")
vals = Counter()
for (s, p, v), n in res.items():
    vals[v] += n
print("results:", dict(vals), "state hashes:", {k: dict(v) for k, v in state_hashes.items()}, "extra variants off:", extra)
