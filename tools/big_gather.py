"""Selection lists (ca_problem.cell_index / gene_index) on a raw matrix of 4.5e9 elements, device and host source: exact library sizes of the 4.3e9 selected elements."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import synth_data as synth
from clonealign_amd.engine import HipEngine
Ns, Gs = 900_000, 5000
Yd, aux = synth.make_problem_torch(Ns, Gs, 4, seed=7, device="cuda:0")
rng = np.random.default_rng(1)
ci = np.sort(rng.choice(Ns, size=870_000, replace=False)).astype(np.int64)
gi = np.sort(rng.choice(Gs, size=4950, replace=False)).astype(np.int32)
print("selected elements %.3e" % (ci.size * gi.size))
want = Yd[:, torch.as_tensor(gi.astype(np.int64), device="cuda:0")].sum(1)[torch.as_tensor(ci, device="cuda:0")].cpu().numpy().astype(np.float64)
for src in ("device", "host"):
    if src == "device":
        eng = HipEngine(None, aux["L"][gi], rng.normal(size=(ci.size, 1)), np.zeros(gi.size) + 0.5, 1, y_device_ptr=Yd.data_ptr(), y_device_dtype=np.int32, shape=(Ns, Gs), cell_index=ci, gene_index=gi)
    else:
        Yh = Yd.cpu().numpy()
        eng = HipEngine(Yh, aux["L"][gi], rng.normal(size=(ci.size, 1)), np.zeros(gi.size) + 0.5, 1, cell_index=ci, gene_index=gi)
    s = eng.get("s")
    print(src, "source:", "OK" if np.array_equal(s, want) else "MISMATCH %d first %d" % ((s != want).sum(), np.flatnonzero(s != want)[0]), "storage", eng.info()["y_storage_name"], flush=True)
    eng.close()
