#!/bin/bash
# LDS / wait counters of the series path's kernels (one --pmc pass, kernel trace only):  gpurun -- 'bash tools/lab/pmc_stream.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_stream
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -d $OUT -o pm --output-format csv -- python3 $R/tools/series_prof.py 100000 5000 8 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("k_ys_mfma_ovf", "k_poly_cell<8>", "k_update_merged"):
    for name, d in acc.items():
        if k in name:
            m = {c: sum(v) / len(v) for c, v in d.items()}
            print(name, {c: f"{v:.3g}" for c, v in m.items()})
            wc = m.get("SQ_WAVE_CYCLES", 0)
            if wc: print("   of wave cycles: LDS instr active %.3f, waiting on LDS %.3f, waiting on anything %.3f; LDS bank-conflict cycles / LDS active cycles %.3f" % (m.get("SQ_ACTIVE_INST_LDS", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
