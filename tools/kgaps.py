"""Kernel sequence of one training step with the idle gaps between kernels (rocprofv3 kernel_trace.csv).
    python tools/kgaps.py gpurun_out/ktrace/kt_kernel_trace.csv [step index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if any(s in r["Kernel_Name"] for s in ("k_snet4", "k_snet3", "k_snet6"))]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
prev = int(rows[i0 - 1]["End_Timestamp"])
tot = 0.0
for j in range(i0, i1):
    r = rows[j]
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += max(gap, 0.0)
    print("%-60s %8.1f us   gap %6.1f" % (r["Kernel_Name"][:60], dur, gap))
    prev = int(r["End_Timestamp"])
print("step %.1f us, idle %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3, tot))
