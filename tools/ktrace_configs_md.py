#!/usr/bin/env python3
"""gpurun_out/ktc/<cfg>_kernel_stats.csv (tools/ktrace_configs.sh) -> one markdown file: per config the kernels of a training
step with calls per step, average duration and share, next to the untraced ms/step of the same process.
    python tools/ktrace_configs_md.py rNN > profiles/rNN_configs_kernel_stats.md"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    print("# rocprofv3 --kernel-trace --stats per config, round %s (tools/ktrace_configs.sh: `tools/bench_configs.py --only <cfg> --exact "
          "--steps 10 --warmup 3` under the tracer)\n" % tag)
    print("Every BASELINE-named config beyond the headline (whose trace is %s_kernel_stats.md).  `per step` = calls / the steps the process "
          "ran (ramp + warm-up + timed + the instrumented leg); set-up kernels run once.\n" % tag)
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "ktc", "*_kernel_stats.csv"))):
        cfg = os.path.basename(f)[:-len("_kernel_stats.csv")]
        rows = list(csv.DictReader(open(f)))
        rec = None
        try:
            for line in open(os.path.join(ROOT, "gpurun_out", "ktc", cfg + ".out")):
                if line.startswith(cfg + " "):
                    rec = json.loads(line.partition(" ")[2])
        except OSError:
            pass
        steps = max(int(r["Calls"]) for r in rows if r["Name"].startswith("k_adam"))
        head = "## %s" % cfg
        if rec:
            head += " — %d points, %.3f ms / step under the tracer (%.1f M points/s)" % (rec["points"], rec["ms_per_step"], rec["Mpts_per_s"])
        print(head + "\n")
        print("| kernel | calls | per step | avg us | us per step | % of GPU time |")
        print("|---|---|---|---|---|---|")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
            print("| `%s` | %d | %.2f | %.1f | %.1f | %s |" % (r["Name"][:96], calls, calls / steps, avg, calls * avg / steps, r["Percentage"]))
        print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
