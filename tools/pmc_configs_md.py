#!/usr/bin/env python3
"""gpurun_out/pmcc/<cfg>.md (tools/pmc_configs.sh) -> profiles/<tag>_configs_traffic.md: HBM bytes per launch of the kernels that
carry each config's step (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of profiles/traffic.json), next to the stash bytes the
design moves for them (tools/bench_configs.py roofline(): rows written + re-read) and the kernel's time in the same step.
    python tools/pmc_configs_md.py r04 > profiles/r04_configs_traffic.md"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(fn):
    out, cur = {}, None
    for line in open(fn):
        m = re.match(r"### `(.*)`  \(dispatches: (\d+)\)", line)
        if m:
            cur = m.group(1); out[cur] = {"n": int(m.group(2))}
        m = re.match(r"- (\w+): mean ([0-9.e+-]+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def write_json(tag):
    """profiles/configs_traffic.json: {csrc_sha, head, source, configs: {cfg: {kernel: {FETCH_SIZE_KB, WRITE_SIZE_KB, launches}}},
    configs_points: {cfg: points}} -- what tools/bench_configs.py's roofline() reads (traffic_stale when the hash is not the build's)"""
    import subprocess
    doc = {"source": "profiles/%s_configs_traffic.md (tools/pmc_configs.sh: rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes around "
                     "tools/bench_configs.py --only <cfg> --exact --steps 6 --warmup 3)" % tag,
           "head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, universal_newlines=True).stdout.strip(),
           "csrc_sha": None, "configs": {}, "configs_points": {}}
    try:
        doc["csrc_sha"] = open(os.path.join(ROOT, "gpurun_out", "pmcc", "csrc_sha.txt")).read().strip()
    except OSError:
        pass
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmcc", "*.md"))):
        cfg = os.path.basename(f)[:-3]
        doc["configs"][cfg] = {name: {"FETCH_SIZE_KB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0.0), "launches": v["n"]}
                               for name, v in parse(f).items()}
        try:
            rec = json.loads(open(f[:-3] + ".out").read().strip().partition(" ")[2])
            doc["configs_points"][cfg] = rec["points"]
        except (OSError, ValueError, KeyError):
            pass
    json.dump(doc, open(os.path.join(ROOT, "profiles", "configs_traffic.json"), "w"), indent=1)


def main(tag):
    print("# HBM traffic per launch of the other BASELINE configs' kernels, round %s (tools/pmc_configs.sh: rocprofv3 --pmc, FETCH_SIZE and "
          "WRITE_SIZE in separate passes)\n" % tag)
    print("bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950's FETCH_SIZE counts half of a 16-B/lane read stream: profiles/traffic.json). "
          "The ms / step in a heading is the step UNDER the counter pass (dispatches serialised: 1.5-2x the untraced step of profiles/%s_configs.json).\n" % tag)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmcc", "*.md"))):
        cfg = os.path.basename(f)[:-3]
        k = parse(f)
        rec = None
        try:
            line = open(f[:-3] + ".out").read().strip()
            rec = json.loads(line.partition(" ")[2])
        except (OSError, ValueError):
            pass
        print("## %s%s\n" % (cfg, " — %d points, %.3f ms / step under the counter pass" % (rec["points"], rec["ms_per_step"]) if rec else ""))
        print("| kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | HBM GB per launch |")
        print("|---|---|---|---|---|")
        rows = []
        for name, v in k.items():
            gb = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 / 1e9
            rows.append((gb * v["n"], name, v, gb))
        tot = 0.0
        for _, name, v, gb in sorted(rows, reverse=True):
            if gb < 0.005:
                continue
            print("| `%s` | %d | %.4g | %.4g | %.3f |" % (name[:90], v["n"], v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0), gb))
        if rec and rec.get("roofline"):
            r = rec["roofline"]
            print("\nfused ShapeNet kernel (%s): %.3f ms per launch under the counter pass, executed 16-bit matrix work %.0f TFLOP/s."
                  % (r.get("kernel", "?"), r["avg_ms"], r["executed_bf16_TFLOPs"]))
        print()


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "json":
    write_json(sys.argv[1])
elif __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
