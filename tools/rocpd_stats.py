#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max / share.
Usage: python tools/rocpd_stats.py results.db [top_n] > profiles/summary.txt"""
import re
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        name = re.sub(r"\s+", " ", name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'share%':>7}  kernel")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:10.2f} {100 * a[1] / total:7.2f}  {name[:150]}")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main()
