"""Per-kernel mean of each PMC counter from rocprofv3 counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/pmc_xxx > profiles/rNN_pmc.md"""
import csv
import glob
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for fn in sorted(glob.glob(d + "/*/*_counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    names = sorted({c for k in acc for c in acc[k]})
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
        print("### `%s`  (dispatches: %d)" % (k[:100], max(len(v) for v in acc[k].values())))
        for c in names:
            if c in acc[k]:
                v = acc[k][c]
                print("- %s: mean %.4g" % (c, sum(v) / len(v)))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
