"""Summarise one tools/prof_round.sh output directory into the files kept under profiles/.

  python tools/summarise_prof.py gpurun_out/prof_v10 r01_v11

writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats), <tag>_bench.json (the default bench line),
<tag>_sq_counters.json (per-launch means of the SQ counters) and <tag>_pmc_hbm.json (HBM bytes per launch:
FETCH_SIZE / WRITE_SIZE are in 64-byte units... see MI355X_MICROARCH.md: KiB units, FETCH_SIZE doubled on gfx950).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
shutil.copy(glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_default.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
build_id = bench.get("config", {}).get("build_id")


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


sq = per_kernel(glob.glob(os.path.join(src, "pmc_sq", "*counter_collection.csv"))[0])
keep = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": len(next(iter(cs.values())))}
        for k, cs in sq.items() if k.startswith("k_") or k.startswith("ca_series::")}
# average launch duration per kernel from the --stats pass of the same command (ns), for the busy fractions below
dur_ns = {}
for r in csv.DictReader(open(os.path.join(out, f"{tag}_kernel_stats.csv"))):
    dur_ns[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
PREFIX = {"fwd": ("k_fwd_cell", "k_fwd_mfma", "k_fwd_lds", "ca_series::k_poly_cell"), "bwd": ("k_bwd_mfma", "k_bwd", "ca_series::k_poly_gene"),
          "ypass": ("k_ypass", "k_yw_mfma", "k_yt_mfma", "k_ys_mfma")}
CLOCK_GHZ, N_SIMD = 2.4, 1024   # nominal shader clock (MI355X_MICROARCH.md); the chip runs nearer 2.0-2.1 GHz under these kernels, so the fractions are lower bounds
sq_classes = {}
for cls, pre in PREFIX.items():
    cand = [(v.get("launches", 0), k) for k, v in keep.items() if k.startswith(pre) and k in dur_ns]
    if not cand:
        continue
    k = max(cand)[1]
    cyc = dur_ns[k] * CLOCK_GHZ * N_SIMD
    sq_classes[cls] = {"kernel": k, "avg_us": dur_ns[k] / 1e3,
                       # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the SIMDs
                       "valu_active_frac": 4.0 * keep[k].get("SQ_ACTIVE_INST_VALU", 0.0) / cyc,
                       "mfma_busy_frac": keep[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cyc}
json.dump({"_doc": f"SQ counters per launch (mean over the sampled launches), one rocprofv3 --pmc pass of `bench.py --steps 6 --warmup 2 "
                   f"--no-cpu-baseline`, MI355X, 100k x 5k x 8, build {tag}. Quad-cycle units summed over waves (MI355X_MICROARCH.md).  classes: "
                   f"fractions of {N_SIMD} SIMDs x launch duration x {CLOCK_GHZ} GHz.",
           "build_id": build_id, "classes": sq_classes, "kernels": keep}, open(os.path.join(out, f"{tag}_sq_counters.json"), "w"), indent=1)

hbm = {}
for which, sub, scale in (("fetch", "pmc_fetch", 2.0), ("write", "pmc_write", 1.0)):
    for k, cs in per_kernel(glob.glob(os.path.join(src, sub, "*counter_collection.csv"))[0]).items():
        if not (k.startswith("k_") or k.startswith("ca_series::")):
            continue
        for c, v in cs.items():
            hbm.setdefault(k, {})[which + "_bytes"] = scale * 1024.0 * sum(v) / len(v)   # KiB units; FETCH_SIZE doubled on gfx950
for k, d in hbm.items():
    d["hbm_bytes"] = d.get("fetch_bytes", 0.0) + d.get("write_bytes", 0.0)
# per kernel class of bench.py's roofline objects: the kernel of that class with the most launches in the profiled run
classes = {}
for cls, pre in PREFIX.items():
    cand = [(keep.get(k, {}).get("launches", 0), k) for k in hbm if k.startswith(pre)]
    if cand:
        k = max(cand)[1]
        classes[cls] = {"kernel": k, "hbm_bytes": hbm[k]["hbm_bytes"]}
json.dump({"build_id": build_id, "classes": classes, "_doc": f"HBM bytes per launch from separate FETCH_SIZE / WRITE_SIZE passes (KiB counters; FETCH_SIZE x2 on gfx950 as "
                   f"MI355X_MICROARCH.md prescribes), build {tag}, same command as the SQ pass.", "kernels": hbm},
          open(os.path.join(out, f"{tag}_pmc_hbm.json"), "w"), indent=1)
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag)))
