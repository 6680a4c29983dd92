"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the same table `--stats` prints:
per-kernel calls, total / average / min / max duration (ns) and share of GPU time.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total_ns | avg_ns | min_ns | max_ns | % | vgpr | agpr | sgpr | lds | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %d | %d | %.0f | %d | %d | %.2f | %s | %s | %s | %s | %s |" % (
            r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main(sys.argv[1])
