#!/usr/bin/env python3
"""Sequential timeline of ONE steady-state training step on the busiest HIP stream of a rocprofv3 kernel trace: for every dispatch its
offset from the step's start, duration, the gap to the predecessor's end, grid and a short kernel name -- the view that shows what a
Bottleneck on a small level is made of (which launches, how long each, how much of the block is boundaries).
Steps are delimited by the optimizer kernel (`k_sgd`), as in rocpd_queues.py.
Usage: python tools/rocpd_timeline.py results.db [step_from_end=2] [filter_regex]
With a filter only the matching dispatches are listed, the summary lines cover all of them."""
import re
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
rows = c.execute(f"select {qcol}, name, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
sgd = sorted(r[3] for r in rows if "k_sgd" in r[1])
if len(sgd) <= back + 1:
    sys.exit(f"need more than {back + 1} optimizer launches in the trace, found {len(sgd)}")
t0, t1 = sgd[-back - 1], sgd[-back]
win = [r for r in rows if r[3] > t0 and r[3] <= t1]
per_q = {}
for r in win:
    per_q.setdefault(r[0], []).append(r)
main = max(per_q, key=lambda q: sum(r[3] - r[2] for r in per_q[q]))
ks = per_q[main]
short = lambda n: re.sub(r"\s+", " ", re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))).replace("void ", "")[:64]
busy = sum(r[3] - r[2] for r in ks)
print(f"step window {(t1 - t0) / 1e6:.3f} ms; stream {main}: {len(ks)} dispatches, busy {busy / 1e6:.3f} ms, "
      f"gaps {((t1 - t0) - busy) / 1e6:.3f} ms")
print(f"{'#':>4} {'at_us':>9} {'dur_us':>8} {'gap_us':>7} {'grid':>9} {'wg':>4} {'lds':>6} {'vgpr':>4}  kernel")
end = t0
for i, (q, name, s, e, gx, gy, wx, lds, vg, ag) in enumerate(ks):
    gap = s - end
    if flt is None or flt.search(name):
        wgs = (gx // max(wx, 1)) * max(gy, 1)
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap / 1e3:7.2f} {wgs:9d} {wx:4d} {lds:6d} {vg + ag:4d}  {short(name)}")
    end = max(end, e)
