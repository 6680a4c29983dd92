#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh: mean FETCH_SIZE / WRITE_SIZE (KB) per launch of the fused
training kernel and of k_given_w, stamped with the git head and the content hash of nif_amd/csrc/ the passes ran on
(<dir>/csrc_sha.txt, written on the GPU box).  bench.py drops `roofline.traffic` when that hash is not the running build's.
usage: tools/make_traffic.py gpurun_out/r04_default profiles/r04_pmc_default.md"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, src = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for fn in sorted(glob.glob(d + "/*/*_counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))

    def pick(prefix):
        ks = [k for k in acc if k.replace("void ", "").startswith(prefix)]
        if not ks:
            return None
        k = max(ks, key=lambda k: sum(acc[k]["WRITE_SIZE"]) + sum(acc[k]["FETCH_SIZE"]))
        mean = lambda v: sum(v) / len(v)
        return {"kernel": k, "FETCH_SIZE_KB": mean(acc[k]["FETCH_SIZE"]), "WRITE_SIZE_KB": mean(acc[k]["WRITE_SIZE"])}

    out = {"source": "%s (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes, bench.py --steps 6 --warmup 3 "
                     "--no-cpu-baseline, 2^20 points; tools/profile_round.sh + tools/make_traffic.py)" % src,
           "points": 1 << 20,
           "head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, universal_newlines=True).stdout.strip(),
           "csrc_sha": open(os.path.join(d, "csrc_sha.txt")).read().strip(),
           "calibration": "gfx950 FETCH_SIZE counts exactly 1/2 of a 16-B/lane coalesced read stream (MI355X_MICROARCH.md #HBM; k_given_w "
                          "reads 8.826e6 KB algorithmically and FETCH_SIZE reports 4.37e6 KB) -> fetch bytes = 2 x FETCH_SIZE x 1024; "
                          "WRITE_SIZE is 1:1"}
    for key, prefix in (("snet6", "k_snet6<"), ("snet", "k_snet4<4, true"), ("k_given_w", "k_given_w<")):
        v = pick(prefix)
        if v:
            out[key] = v
    if "k_given_w" in out:
        out["k_given_w"]["points"] = 131072
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
