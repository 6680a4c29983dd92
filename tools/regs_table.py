#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS table of the built library, from the code-object metadata of every
translation unit (build/obj/*.o -> .hip_fatbin -> gfx950 code object -> llvm-readelf --notes).
usage: tools/regs_table.py [--all] > profiles/r03_regs.md      (default: kernels that spill or use scratch, plus totals)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, universal_newlines=True)
    return p.stdout.split("\n")


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "co.elf")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "x.o")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        return []          # a translation unit without device code
    lst = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--input=" + fat, "--list"],
                         stdout=subprocess.PIPE, universal_newlines=True).stdout.split()
    tgt = [t for t in lst if "gfx950" in t]
    if not tgt:
        return []
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--input=" + fat, "--targets=" + tgt[0],
                           "--output=" + co, "--unbundle"])
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, universal_newlines=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        get = lambda k, d=0: int(re.search(r"\." + k + r":\s*(\d+)", blk).group(1)) if re.search(r"\." + k + r":\s*(\d+)", blk) else d
        nm = re.search(r"\.name:\s*(\S+)", blk)
        if not nm:
            continue
        out.append(dict(name=nm.group(1), vgpr=get("vgpr_count"), vspill=get("vgpr_spill_count"), sspill=get("sgpr_spill_count"),
                        scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"), sgpr=get("sgpr_count")))
    return out


def main():
    show_all = "--all" in sys.argv
    objdir = os.path.join(ROOT, "build", "obj")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(objdir)):
            if f.endswith(".o"):
                for k in kernels_of(os.path.join(objdir, f), tmp):
                    k["tu"] = f[:-2]
                    rows.append(k)
    dem = demangle([r["name"] for r in rows])
    for r, d in zip(rows, dem):
        r["dem"] = re.sub(r"\(.*$", "", d)[:110]
    nsp = sum(1 for r in rows if r["vspill"] > 0)
    print("# registers / spills / scratch per kernel (tools/regs_table.py, from the code objects of build/obj/*.o)\n")
    print("%d kernels in %d translation units; **%d spill VGPRs**, %d use scratch, %d spill SGPRs (to VGPR lanes).\n"
          % (len(rows), len(set(r["tu"] for r in rows)), nsp, sum(1 for r in rows if r["scratch"] > 0), sum(1 for r in rows if r["sspill"] > 0)))
    print("| tu | kernel | vgpr | spilled vgpr | scratch B | spilled sgpr |\n|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: (-r["vspill"], -r["scratch"], r["tu"], r["dem"])):
        if show_all or r["vspill"] > 0 or r["scratch"] > 0:
            print("| %s | `%s` | %d | %d | %d | %d |" % (r["tu"], r["dem"], r["vgpr"], r["vspill"], r["scratch"], r["sspill"]))
    print("\n## per translation unit\n\n| tu | kernels | spilling | max spilled vgpr |\n|---|---|---|---|")
    for tu in sorted(set(r["tu"] for r in rows)):
        rr = [r for r in rows if r["tu"] == tu]
        print("| %s | %d | %d | %d |" % (tu, len(rr), sum(1 for r in rr if r["vspill"] > 0), max(r["vspill"] for r in rr)))


if __name__ == "__main__":
    main()
