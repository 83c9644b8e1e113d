#!/usr/bin/env python3
"""Kernel time per HIP stream / queue of a rocprofv3 kernel trace (which work runs beside the main stream).
Usage: python tools/rocpd_queues.py results.db [steps]"""
import sqlite3, sys, re
db = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
rows = c.execute(f"select {qcol}, name, (end - start) from kernels").fetchall()
agg = {}
for q, name, dur in rows:
    a = agg.setdefault(q, {"n": 0, "t": 0, "fps": 0, "top": {}, "cnt": {}})
    a["n"] += 1; a["t"] += dur
    if "k_fps" in name: a["fps"] += dur
    k = re.sub(r"\(.*", "", name)[:60]
    a["top"][k] = a["top"].get(k, 0) + dur
    a["cnt"][k] = a["cnt"].get(k, 0) + 1
for q, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    print(f"{qcol}={q}: {a['n'] / steps:8.1f} dispatches/step  {a['t'] / 1e6 / steps:8.3f} ms/step  (k_fps {a['fps'] / 1e6 / steps:.3f})")
    for k, t in sorted(a["top"].items(), key=lambda kv: -kv[1])[:8]:
        print(f"       {t / 1e6 / steps:8.3f}  {k}")
    print("    by launch count (launches/step, ms/step):")
    for k, n in sorted(a["cnt"].items(), key=lambda kv: -kv[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
        print(f"       {n / steps:8.1f} {a['top'][k] / 1e6 / steps:8.3f}  {k}")
