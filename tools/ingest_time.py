"""Ingestion at the C ABI, from HOST memory: what clonealign() hands over is an N x G column-major double matrix
(R/clonealign.R:212-222; R/inference-tflow.R:355,401,403 re-feed it to every sess$run), here once per fit.

  python tools/ingest_time.py [cells genes clones] [--tag before|after]

For float64 column-major (R), int32 column-major and int32 row-major host matrices of the same counts:
  ca_create (HipEngine(...)) wall time, against the same engine built from a DEVICE pointer (no host bytes to move) -- the
  difference is the ingestion; then ca_run(200) + 20 final ELBOs.  Beside it the box's own copy rates for the same bytes:
  pinned hipMemcpy (the roof of any upload) and pageable hipMemcpy (what a single full-size copy gets)."""
import argparse
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import synth_data as synth  # noqa: E402
from clonealign_amd.engine import HipEngine, build_id  # noqa: E402
from clonealign_amd.hostprep import safe_inverse_softplus  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int, default=[100_000, 5_000, 8])
ap.add_argument("--tag", default="")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
N, G, C = args.shape
print(f"== ingest_time {args.tag}: {N} x {G} x {C}, build {build_id()}")
Yd, aux = synth.make_problem_torch(N, G, C, seed=20243, device="cuda:0")
rm = Yd.sum(1, keepdim=True).to(torch.float64) / G
col = torch.zeros(G, dtype=torch.float64, device="cuda:0")
for b0 in range(0, N, 8192):
    col += (Yd[b0:b0 + 8192].to(torch.float64) / rm[b0:b0 + 8192]).sum(0)
loc0 = safe_inverse_softplus(np.maximum(col.cpu().numpy() / N, 1e-6))
psi0 = np.random.default_rng(1).normal(size=(N, 1))
Yh = Yd.cpu().numpy()                                   # int32 row-major
forms = {
    "float64 col-major (R)": (np.asfortranarray(Yh.astype(np.float64)), "col"),
    "int32 col-major": (np.asfortranarray(Yh), "col"),
    "int32 row-major": (Yh, "row"),
}

# --- the box's copy rates for these byte counts
def rate(host_t, reps=3):
    dev = torch.empty_like(host_t, device="cuda:0")
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(host_t, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return host_t.numel() * host_t.element_size() / min(ts) / 1e9, min(ts) * 1e3


for nbytes in (N * G * 4, N * G * 8):
    pg = torch.empty(nbytes // 4, dtype=torch.int32)
    pg.fill_(1)
    t0 = time.perf_counter()
    pn = torch.empty(nbytes // 4, dtype=torch.int32).pin_memory()
    t_pin = time.perf_counter() - t0
    pn.fill_(1)
    r_pin, ms_pin = rate(pn)
    r_pg, ms_pg = rate(pg)
    print(f"{nbytes / 1e9:.1f} GB: pinned copy {r_pin:6.1f} GB/s ({ms_pin:6.1f} ms; pinning the buffer itself took {t_pin * 1e3:.0f} ms)   "
          f"pageable copy {r_pg:6.1f} GB/s ({ms_pg:6.1f} ms)")
    del pg, pn

# --- the engine from a device pointer: everything but the host bytes
def build(**kw):
    ts, eng = [], None
    for _ in range(args.reps):
        if eng is not None:
            eng.close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng = HipEngine(**kw)
        eng.synchronize()
        ts.append(time.perf_counter() - t0)
    return eng, min(ts) * 1e3, float(np.median(ts)) * 1e3


eng, t_dev, t_dev_med = build(Y=None, L=aux["L"], psi0=psi0, loc0=loc0, K=1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(N, G))
ref = None
print(f"ca_create from a DEVICE int32 pointer: {t_dev:7.1f} ms (median {t_dev_med:.1f}) -- scan + conversion + fit constants, no host bytes")
eng.close()
for name, (Y, lay) in forms.items():
    L = np.asfortranarray(aux["L"]) if lay == "col" else aux["L"]
    eng, t_min, t_med = build(Y=Y, L=L, psi0=psi0, loc0=loc0, K=1, layout=lay)
    gb = Y.nbytes / 1e9
    t0 = time.perf_counter()
    tr = eng.run(None, 200, 1e-6)
    fin = eng.final_elbo(None, 20)
    t_fit = (time.perf_counter() - t0) * 1e3
    st = (tr[-1], eng.get("mu")[:4].tolist())
    if ref is None:
        ref = st
    same = st[0] == ref[0] and st[1] == ref[1]
    print(f"{name:24s} {gb:4.1f} GB host: ca_create {t_min:7.1f} ms (median {t_med:.1f}) = {gb / (t_min * 1e-3):5.1f} GB/s of host bytes end to end; "
          f"ingestion over the device-pointer build {t_min - t_dev:7.1f} ms = {gb / max((t_min - t_dev) * 1e-3, 1e-9):5.1f} GB/s;  "
          f"ca_run(200) + 20 final ELBOs {t_fit:6.1f} ms; storage {eng.info()['y_storage_name']}; same fit as the first form: {same}")
    eng.close()
