#!/usr/bin/env python3
"""Measurement builds next to the product library: recompile the named translation units with extra flags into
build/obj_<tag>/, reuse every other object of the regular build, link nif_amd/libnif_hip_<tag>.so (load it with
NIF_LIB=...).  usage: build_variant.py <tag> "<extra flags>" k_snet4.hip [nif_api.hip ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    tag, extra, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    G.build()
    od = os.path.join(ROOT, "build", "obj_" + tag)
    os.makedirs(od, exist_ok=True)
    objs = []
    procs = []
    for s in G.SOURCES:
        if s in srcs:
            o = os.path.join(od, s.replace(".hip", ".o"))
            procs.append(subprocess.Popen([G.HIPCC] + G.FLAGS + extra + ["-c", os.path.join(G.CSRC, s), "-o", o]))
        else:
            o = os.path.join(G.OBJ, s.replace(".hip", ".o"))
        objs.append(o)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("compile failed")
    lib = os.path.join(ROOT, "nif_amd", "libnif_hip_%s.so" % tag)
    subprocess.check_call([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    print(lib)


if __name__ == "__main__":
    main()
