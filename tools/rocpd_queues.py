#!/usr/bin/env python3
"""Kernel time and dispatch counts per HIP stream of a rocprofv3 kernel trace, per STEADY-STATE training step.

Steps are delimited by the optimizer kernel (`k_sgd`: exactly one launch per step, the last kernel of a step): the window between the
end of the (last - w)-th and the end of the last `k_sgd` launch holds exactly w steps -- parameter initialisation (thousands of one-off
H2D copies), graph capture warm-ups and the first group's pre-pass are outside it.  (Round 1 / 2 divided the WHOLE trace by the step
count, which booked ~150 one-off `copyBuffer` launches per step.)
Usage: python tools/rocpd_queues.py results.db [steps_in_window=12] [rows=12]"""
import re
import sqlite3
import sys

db = sys.argv[1]
w = int(float(sys.argv[2])) if len(sys.argv) > 2 else 12
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
rows = c.execute(f"select {qcol}, name, start, end from kernels order by end").fetchall()
sgd = [r[3] for r in rows if "k_sgd" in r[1]]
if len(sgd) > w:
    t0, t1, steps = sgd[-w - 1], sgd[-1], float(w)
    print(f"steady-state window: the last {w} of {len(sgd)} optimizer steps ({(t1 - t0) / 1e6 / w:.3f} ms per step wall on the device timeline)")
else:
    t0, t1, steps = rows[0][2], rows[-1][3], float(max(len(sgd), 1))
    print(f"no k_sgd delimiters ({len(sgd)}): whole trace divided by {steps}")
agg = {}
for q, name, start, end in rows:
    if end <= t0 or end > t1:
        continue
    a = agg.setdefault(q, {"n": 0, "t": 0, "fps": 0, "top": {}, "cnt": {}})
    dur = end - start
    a["n"] += 1; a["t"] += dur
    if "k_fps" in name:
        a["fps"] += dur
    k = re.sub(r"\(.*", "", name)[:60]
    a["top"][k] = a["top"].get(k, 0) + dur
    a["cnt"][k] = a["cnt"].get(k, 0) + 1
tot_n = sum(a["n"] for a in agg.values())
print(f"all streams: {tot_n / steps:.1f} dispatches/step")
for q, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    print(f"{qcol}={q}: {a['n'] / steps:8.1f} dispatches/step  {a['t'] / 1e6 / steps:8.3f} ms/step  (k_fps {a['fps'] / 1e6 / steps:.3f})")
    for k, t in sorted(a["top"].items(), key=lambda kv: -kv[1])[:8]:
        print(f"       {t / 1e6 / steps:8.3f}  {k}")
    print("    by launch count (launches/step, ms/step):")
    for k, n in sorted(a["cnt"].items(), key=lambda kv: -kv[1])[:top]:
        print(f"       {n / steps:8.1f} {a['top'][k] / 1e6 / steps:8.3f}  {k}")
