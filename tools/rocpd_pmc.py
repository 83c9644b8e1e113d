#!/usr/bin/env python3
"""Per-kernel PMC summary from a rocprofv3 (rocpd sqlite) --pmc run.
Usage: python tools/rocpd_pmc.py results.db [top_n]   -> calls, avg counter value per launch, per kernel and counter."""
import re
import sqlite3
import sys


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
    cname = "counter_name" if "counter_name" in cols else "name"
    val = "value" if "value" in cols else "counter_value"
    rows = c.execute(f"select {name_col}, {cname}, {val} from counters_collection").fetchall()
    agg = {}
    for k, cn, v in rows:
        k = re.sub(r"\s+", " ", str(k))
        a = agg.setdefault((k, cn), [0, 0.0])
        a[0] += 1; a[1] += float(v)
    print("columns:", cols)
    print(f"{'calls':>7} {'avg_per_launch':>16} {'total':>16}  counter  kernel")
    for (k, cn), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{a[0]:7d} {a[1] / a[0]:16.1f} {a[1]:16.1f}  {cn}  {k[:130]}")


if __name__ == "__main__":
    main()
