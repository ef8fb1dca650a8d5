#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc runs (rocpd SQLite).  Usage:
   scripts/rocpd_pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db ... > profiles/X.md
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE tallies a 128-byte request
of a wide coalesced streaming read as 64 bytes (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the `x2` column
applies that correction for kernels whose reads are 16 B/lane streams."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("eesen::", "")


def main(paths):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        c = sqlite3.connect(p)
        for kn, cn, v, d in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            a = agg[short(kn)][cn]
            a[0] += v; a[1] += 1
            dd = dur[short(kn)]; dd[0] += d; dd[1] += 1
    counters = sorted({cn for k in agg.values() for cn in k})
    print("# rocprofv3 --pmc per-kernel averages (per dispatch)\n")
    print("| kernel | dispatches | avg us (profiled) | " + " | ".join(counters) + " |")
    print("|---|---:|---:|" + "---:|" * len(counters))
    order = sorted(agg, key=lambda k: -dur[k][0])
    for k in order:
        n = max(v[1] for v in agg[k].values())
        row = [f"{agg[k][cn][0] / agg[k][cn][1]:.4g}" if cn in agg[k] else "" for cn in counters]
        print(f"| `{k}` | {n} | {dur[k][0] / dur[k][1] / 1e3:.2f} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
