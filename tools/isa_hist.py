#!/usr/bin/env python3
"""Instruction histogram of one kernel's ISA, per basic block (hipcc -S output).

usage: isa_hist.py file.s mangled_kernel_prefix [--blocks]
Prints the totals per instruction class and, with --blocks, every basic block with its VALU / MFMA / LDS / VMEM /
SALU counts so that the hot loop bodies can be told from the cold paths (tools for DESIGN 5.3 / profiles/r03_regs.md)."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    blocks, cur, name = [], Counter(), "entry"
    ops = Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append((name, cur))
            cur, name = Counter(), m.group(1)
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if not m:
            continue
        op = m.group(1)
        c = classify(op)
        cur[c] += 1
        ops[op] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"):
            cur["lane"] += 1
        if "f64" in op:
            cur["f64"] += 1
        if op.startswith("scratch_"):
            cur["scratch"] += 1
    blocks.append((name, cur))
    tot = Counter()
    for _, c in blocks:
        tot.update(c)
    print("total:", dict(tot))
    print("top ops:", ops.most_common(40))
    if show_blocks:
        for nm, c in blocks:
            if sum(c.values()) >= 8:
                print("%-12s valu %4d mfma %3d lds %3d vmem %3d salu %3d wait %3d lane %3d f64 %3d scratch %2d" % (
                    nm, c["valu"], c["mfma"], c["lds"], c["vmem"], c["salu"], c["wait"], c["lane"], c["f64"], c["scratch"]))


if __name__ == "__main__":
    main()
