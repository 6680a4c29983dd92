#!/usr/bin/env python3
"""Per-loop instruction totals of one kernel (hipcc -S output): groups the basic blocks by the `in Loop: Header=` comment
LLVM prints, so that the steady-state loop bodies can be compared between builds.  usage: isa_loops.py file.s mangled_prefix"""
import re, sys
from collections import Counter, defaultdict
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_hist import classify

def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    cur = "top"
    tot = defaultdict(Counter)
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):\s*;\s*(.*)$", l)
        if m:
            h = re.search(r"Header=(BB[0-9_]+) Depth=(\d+)", m.group(2))
            cur = ("%s d%s" % (h.group(1), h.group(2))) if h else "top"
            continue
        if re.match(r"^(\.LBB[0-9_]+):", l):
            cur = "top"; continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if not m: continue
        op = m.group(1)
        tot[cur][classify(op)] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"): tot[cur]["lane"] += 1
        if "f64" in op: tot[cur]["f64"] += 1
        if op.startswith("v_cndmask"): tot[cur]["cnd"] += 1
    for k, c in tot.items():
        print("%-14s valu %4d mfma %3d lds %3d vmem %3d salu %4d lane %3d f64 %3d cnd %3d" % (k, c["valu"], c["mfma"], c["lds"], c["vmem"], c["salu"], c["lane"], c["f64"], c["cnd"]))

if __name__ == "__main__":
    main()
