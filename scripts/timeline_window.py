#!/usr/bin/env python3
"""Every kernel (no duration filter) of the LAST training step in a rocprofv3 rocpd database between two substrings of kernel names,
with the idle gap in front of each.  usage: python scripts/timeline_window.py results.db [from-substring [to-substring]]"""
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
prev_end = {}
for r in rows[beg:]:
    n = short(r[0]); st = (r[1] - t0) / 1e3; du = (r[2] - r[1]) / 1e3
    gap = st - prev_end.get(r[4], st)
    print(f"{n[:60]:60s} start {st:9.1f} us  dur {du:8.1f}  gap {gap:7.1f}  q{r[4]}")
    prev_end[r[4]] = st + du
