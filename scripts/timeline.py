#!/usr/bin/env python3
"""Kernel timeline of the LAST training step in a rocprofv3 rocpd database (kernels >= 50 us, plus every persistent launch).
usage: python scripts/timeline.py results.db"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,stream_id,queue_id from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*$", "", n)
    return n.replace("void ", "").replace("eesen::", "")
upd = [i for i, r in enumerate(rows) if "sgd_update" in r[0] or "adaptive_update" in r[0]]
last_sm = max(i for i, r in enumerate(rows) if "softmax_rows" in r[0])
beg = max(i for i in upd if i < last_sm) + 1
t0 = rows[beg][1]
for r in rows[beg:]:
    n = short(r[0]); st = (r[1] - t0) / 1e6; du = (r[2] - r[1]) / 1e6
    if du > 0.05 or "persistent" in n:
        print(f"{n[:46]:46s} start {st:8.3f} dur {du:7.3f} end {st+du:8.3f} q{r[4]}")
print("step total", (rows[-1][2] - t0) / 1e6)
