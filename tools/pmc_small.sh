#!/bin/bash
# SQ counters of the loop's kernels at a small shard (one --pmc pass; no other trace domain):  tools/pmc_small.sh "<bench args>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3/pmc_small
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $OUT -o ps --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --busy-seconds 0 $1 > $OUT/log.txt 2>&1; tail -3 $OUT/log.txt; ls $OUT
cd $R
python3 - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/r3/pmc_small/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(k.startswith(p) for p in ("k_fwd_cell", "k_bwd_mfma")): continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k[:40], {c: round(v) for c, v in m.items()})
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print("   per wave-cycle: VALU active %.2f, waiting on waitcnt %.2f, waiting to issue %.2f; MFMA busy / busy cycles %.2f; waves %d" % (
        m["SQ_ACTIVE_INST_VALU"] * 4 / wc if wc else 0, m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(m["SQ_BUSY_CYCLES"], 1), m["SQ_WAVES"]))
PY
