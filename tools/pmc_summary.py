"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv) for the engine's kernels."""
import csv, sys, collections
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "at::" in k or "rocclr" in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
if __name__ == "__main__":
    for p in sys.argv[1:]:
        for k, cs in load(p).items():
            print(k, {c: (sum(v) / len(v), len(v)) for c, v in cs.items()})
