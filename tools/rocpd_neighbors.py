#!/usr/bin/env python3
"""What runs right before / after a given kernel on its stream (rocprofv3 rocpd sqlite).  Usage: rocpd_neighbors.py db pattern [n]"""
import re, sqlite3, sys, collections
db, pat, top = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 15
c = sqlite3.connect(db)
rows = c.execute("select stream_id, start, name, grid_x, workgroup_x from kernels order by stream_id, start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n)[:70]
ctx = collections.Counter(); sizes = collections.Counter()
for i, (sid, st, name, gx, wx) in enumerate(rows):
    if re.search(pat, name):
        prev = short(rows[i - 1][2]) if i > 0 and rows[i - 1][0] == sid else "-"
        nxt = short(rows[i + 1][2]) if i + 1 < len(rows) and rows[i + 1][0] == sid else "-"
        ctx[(sid, prev, nxt)] += 1
        sizes[(gx, wx)] += 1
for (sid, prev, nxt), n in ctx.most_common(top):
    print(f"{n:6d}  stream {sid}   after [{prev}]   before [{nxt}]")
print("grid sizes:", sizes.most_common(8))
