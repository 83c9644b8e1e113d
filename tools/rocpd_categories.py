#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace (rocpd sqlite) into coarse categories: ms, dispatch count, share -- per STEADY-STATE training step: the
window between the end of the (last - w)-th and the end of the last optimizer launch (`k_sgd`: one per step) holds exactly w steps; parameter
initialisation, capture warm-ups and the first group's pre-pass lie outside it (rounds 1-3 divided the whole trace by a step count).
Usage: python tools/rocpd_categories.py results.db [steps in the window = 12]"""
import re
import sqlite3
import sys

CATS = [
    ("fps", r"k_fps|k_bbox|k_hist|k_scan\b|k_scatter|k_pad|k_meta|fps_"),
    ("knn grid/scan", r"kg::|knn_"),
    ("pt-layer fwd (k_p*)", r"fl[ms]?::k_p\d"),
    ("pt-layer bwd (k_b*)", r"fl[ms]?::k_b\d"),
    ("bn finalize/colsum/eval", r"fl::k_bn_finalize|fl::k_colsum|fl::k_bn_eval"),
    ("pointwise bn (pw::)", r"pw::"),
    ("rowlin (rl:: / rl2::)", r"rl2?::"),
    ("cross-entropy", r"k_ce_"),
    ("rocBLAS GEMM", r"Cijk_"),
    ("group/interp/other pdfops", r"anonymous namespace"),
    ("torch reduce", r"reduce_kernel"),
    ("torch elementwise", r"elementwise_kernel|at::native"),
    ("copy/fill (runtime)", r"__amd_rocclr"),
]


def main():
    db = sys.argv[1]
    w = int(float(sys.argv[2])) if len(sys.argv) > 2 else 12
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by end").fetchall()
    sgd = [r[2] for r in rows if "k_sgd" in r[0]]
    if len(sgd) > w:
        t0, t1, steps = sgd[-w - 1], sgd[-1], float(w)
        print(f"steady-state window: the last {w} of {len(sgd)} optimizer steps, all streams ({(t1 - t0) / 1e6 / w:.3f} ms per step wall on the device timeline)")
    else:
        t0, t1, steps = rows[0][1] - 1, rows[-1][2], float(max(len(sgd), 1))
        print(f"no window of {w} optimizer steps ({len(sgd)} k_sgd launches): whole trace divided by {steps}")
    agg = {}
    for name, start, end in rows:
        if end <= t0 or end > t1:
            continue
        dur = end - start
        for cat, pat in CATS:
            if re.search(pat, name):
                break
        else:
            cat = "other"
        a = agg.setdefault(cat, [0, 0])
        a[0] += 1; a[1] += dur
    tot = sum(a[1] for a in agg.values())
    print(f"{'category':34s} {'dispatches/step':>16s} {'ms/step':>10s} {'share%':>8s}")
    for cat, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{cat:34s} {a[0] / steps:16.1f} {a[1] / 1e6 / steps:10.3f} {100 * a[1] / tot:8.2f}")
    print(f"{'total':34s} {sum(a[0] for a in agg.values()) / steps:16.1f} {tot / 1e6 / steps:10.3f}")


if __name__ == "__main__":
    main()
