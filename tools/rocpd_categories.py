#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace (rocpd sqlite) into coarse categories: total ms, dispatch count, share.
Usage: python tools/rocpd_categories.py results.db [steps]"""
import re
import sqlite3
import sys

CATS = [
    ("fps", r"k_fps|k_bbox|k_hist|k_scan\b|k_scatter|k_pad|k_meta|fps_"),
    ("knn grid/scan", r"kg::|knn_"),
    ("pt-layer fwd (k_p*)", r"flm?::k_p\d"),
    ("pt-layer bwd (k_b*)", r"flm?::k_b\d"),
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
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    agg = {}
    for name, dur in c.execute("select name, (end - start) from kernels"):
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
