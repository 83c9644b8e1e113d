#!/usr/bin/env python3
"""Idle gaps of the busiest HIP stream of a rocprofv3 kernel trace, per STEADY-STATE training step (steps delimited by `k_sgd`, as in
rocpd_queues.py): busy time, idle time, and the largest gaps with the kernels on either side -- where a host-driven stage (the PDF
pseudo-label pass) leaves the device waiting.  Usage: python tools/rocpd_gaps.py results.db [steps_in_window=8] [gaps=12]"""
import re
import sqlite3
import sys

db = sys.argv[1]
w = int(sys.argv[2]) if len(sys.argv) > 2 else 8
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
rows = c.execute(f"select {qcol}, name, start, end from kernels order by start").fetchall()
sgd = sorted(r[3] for r in rows if "k_sgd" in r[1])
t0, t1 = sgd[-w - 1], sgd[-1]
win = [r for r in rows if r[3] > t0 and r[3] <= t1]
per_q = {}
for q, name, s, e in win:
    per_q.setdefault(q, []).append((s, e, name))
main = max(per_q, key=lambda q: sum(e - s for s, e, _ in per_q[q]))
ks = sorted(per_q[main])
busy = sum(e - s for s, e, _ in ks)
print(f"window: {w} steps, {(t1 - t0) / 1e6 / w:.2f} ms per step; busiest stream {main}: {len(ks) / w:.0f} dispatches/step, "
      f"busy {busy / 1e6 / w:.2f} ms/step, idle {(t1 - t0 - busy) / 1e6 / w:.2f} ms/step")
others = sum(e - s for q in per_q if q != main for s, e, _ in per_q[q])
print(f"other streams: kernel time {others / 1e6 / w:.2f} ms/step")
short = lambda n: re.sub(r"\(.*", "", n)[:70]
gaps = []
end = ks[0][1]
prev = ks[0][2]
for s, e, name in ks[1:]:
    if s > end:
        gaps.append((s - end, end, prev, name))
    if e > end:
        end, prev = e, name
for lo, hi in ((0, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f"gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(g) / w:.1f} per step, {sum(x[0] for x in g) / 1e6 / w:.2f} ms/step")
print("largest gaps of the window (offset from the end of the preceding optimizer launch):")
import bisect
for d, at, a, b in sorted(gaps, reverse=True)[:top]:
    k = bisect.bisect_right(sgd, at) - 1
    print(f"  {d / 1e3:9.1f} us at +{(at - sgd[k]) / 1e6:6.2f} ms of step {k}   after {short(a)}   before {short(b)}")
