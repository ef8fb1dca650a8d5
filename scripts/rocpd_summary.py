#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (-d DIR -o NAME => NAME_results.db) into the per-kernel stats
table that `rocprofv3 --kernel-trace --stats` reports: calls, total / average / min / max duration, share.
Usage: scripts/rocpd_summary.py gpurun_out/prof/NAME_results.db > profiles/NAME_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)           # drop the argument list
    return name.replace("void ", "").replace("eesen::", "")


def main(path: str):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "min(grid_x)||'x'||min(grid_y)||'x'||min(grid_z), min(workgroup_x), max(vgpr_count), max(lds_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid (threads) | wg | vgpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---|---:|---:|---:|")
    for n, calls, tot, avg, mn, mx, grid, wg, vg, lds in rows:
        print(f"| `{short(n)}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} | {grid} | {wg} | {vg} | {lds} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
