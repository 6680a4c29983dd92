#!/usr/bin/env python3
"""profiles/<tag>_{kernel_stats,pmc_default}.md, <tag>_configs.json and traffic.json from what tools/gpu_round.sh <tag> left under
gpurun_out/ (headers of the existing summaries are kept, their source hash updated).  usage: tools/refresh_profiles.py r04"""
import csv, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
d = os.path.join(ROOT, "gpurun_out", tag + "_default")
sha = open(os.path.join(d, "csrc_sha.txt")).read().strip()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), d, "profiles/%s_pmc_default.md" % tag], stdout=subprocess.DEVNULL)
rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
ks = os.path.join(ROOT, "profiles", tag + "_kernel_stats.md")
hdr = re.sub(r"csrc [0-9a-f]{16}", "csrc " + sha, open(ks).read().split("| kernel |")[0])
out = [hdr.rstrip("\n"), "", "| kernel | calls | total ns | avg ns | % | min ns | max ns |", "|---|---|---|---|---|---|---|"]
for r in rows:
    out.append("| `%s` | %s | %s | %d | %s | %s | %s |" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
open(ks, "w").write("\n".join(out) + "\n")
pm = os.path.join(ROOT, "profiles", tag + "_pmc_default.md")
hdrp = re.sub(r"csrc [0-9a-f]{16}", "csrc " + sha, open(pm).read().split("### `")[0])
open(pm, "w").write(hdrp + open(os.path.join(d, "pmc.md")).read())
# the per-config table: tools/bench_configs.py's JSON DOCUMENT (gpu_round.sh copies it next to the per-line log)
import json
cfg = os.path.join(ROOT, "gpurun_out", tag + "_configs_doc.json")
if os.path.exists(cfg):
    json.dump(json.load(open(cfg)), open(os.path.join(ROOT, "profiles", tag + "_configs.json"), "w"), indent=1)
print(sha, rows[0]["Name"][:30], rows[0]["AverageNs"])
