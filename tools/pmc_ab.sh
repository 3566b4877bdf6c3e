#!/bin/bash
# SQ counters of the sweeps with and without the side stream (one --pmc pass each; no other trace domain).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
for tag in async serial; do
  extra=""; [ $tag = serial ] && extra="--variant-off async_y"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/r2/pmc_$tag -o ps --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline $extra > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob
for tag in ("async", "serial"):
    f = glob.glob(f"gpurun_out/r2/pmc_{tag}/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag)
    for k, cs in acc.items():
        if not any(k.startswith(p) for p in ("k_fwd_cell", "k_bwd_mfma", "k_yw_mfma", "k_yt_mfma", "k_ypass")): continue
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        wc = m["SQ_WAVE_CYCLES"]
        print("  %-28s waves %6.0f wave_cyc %8.1fM busy %6.2fM | wait_any %4.1f%% wait_inst %4.1f%% active_valu %4.1f%% | insts_valu %7.2fM mfma_busy %6.1fM" % (
            k[:28], m["SQ_WAVES"], wc / 1e6, m["SQ_BUSY_CYCLES"] / 1e6, 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc,
            100 * m["SQ_ACTIVE_INST_VALU"] / wc, m["SQ_INSTS_VALU"] / 1e6, m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1e6))
PY
