#!/usr/bin/env python3
"""Runs of consecutive launches of one kernel on a stream: length, what came before the run and after it.  rocpd_runs.py db pattern"""
import re, sqlite3, sys, collections
db, pat = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select stream_id, start, name from kernels order by stream_id, start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n)[:80]
runs = collections.Counter()
i = 0
while i < len(rows):
    if re.search(pat, rows[i][2]):
        j = i
        while j + 1 < len(rows) and rows[j + 1][0] == rows[i][0] and re.search(pat, rows[j + 1][2]):
            j += 1
        prev = short(rows[i - 1][2]) if i > 0 and rows[i - 1][0] == rows[i][0] else "-"
        nxt = short(rows[j + 1][2]) if j + 1 < len(rows) and rows[j + 1][0] == rows[i][0] else "-"
        runs[(rows[i][0], j - i + 1, prev, nxt)] += 1
        i = j + 1
    else:
        i += 1
for (sid, ln, prev, nxt), n in sorted(runs.items(), key=lambda kv: -kv[0][1] * kv[1])[:20]:
    print(f"{n:5d} runs of {ln:4d} on stream {sid}: after [{prev}] before [{nxt}]")
