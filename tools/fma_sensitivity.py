"""How many kNN rows / FPS picks change when the squared distance is evaluated with FMA contraction (what an `nvcc -O2` build of
libs/pointops most likely does) instead of as written?  CPU oracle only (oracle/pdfops_oracle.c: oracle_sqdist3, modes 0 / 1 / 2).

    python tools/fma_sensitivity.py [--json profiles/r03_fma_sensitivity.json]

`measure()` is also what tests/test_oracle_fma.py runs (smaller query sample)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODES = {"as_written": 0, "fma1": 1, "fma2": 2}


def _one(be, xyz, n_queries, fps_levels, seed):
    n = xyz.shape[0]
    off = torch.tensor([n], dtype=torch.int32)
    q = torch.from_numpy(np.random.default_rng(seed).choice(n, size=min(n_queries, n), replace=False).astype(np.int64)).sort()[0]
    qxyz, qoff = xyz[q].contiguous(), torch.tensor([q.numel()], dtype=torch.int32)
    out, base = {"points": n, "queries": int(q.numel())}, None
    for name, mode in MODES.items():
        be.set_dist_mode(mode)
        cur = {}
        for k in (8, 16):
            cur[f"knn{k}"] = be.knn_query(k, xyz, qxyz, off, qoff)
        pts, picks, m = xyz, [], n
        for lvl in range(fps_levels):
            m_next = m // 4
            f = be.farthest_point_sampling(pts, torch.tensor([m], dtype=torch.int32), torch.tensor([m_next], dtype=torch.int32), m, m_next)
            picks.append(f)
            pts, m = pts[f.long()].contiguous(), m_next
        cur["fps"] = picks
        if base is None:
            base = cur
            continue
        r = {}
        for k in (8, 16):
            (i0, d0), (i1, d1) = base[f"knn{k}"], cur[f"knn{k}"]
            r[f"knn_k{k}_idx_rows_differ"] = int((i0 != i1).any(1).sum())
            r[f"knn_k{k}_idx_sets_differ"] = int((i0.sort(1)[0] != i1.sort(1)[0]).any(1).sum())
            r[f"knn_k{k}_dist_rows_differ"] = int((d0 != d1).any(1).sum())
        r["fps_picks_differ"] = int(sum((a != b).sum() for a, b in zip(base["fps"], cur["fps"]) if a.shape == b.shape))
        r["fps_pick_sets_differ"] = int(sum(np.setdiff1d(a.numpy(), b.numpy()).size for a, b in zip(base["fps"], cur["fps"])))
        r["fps_first_difference_at"] = [int((a != b).float().argmax()) if bool((a != b).any()) else -1 for a, b in zip(base["fps"], cur["fps"])]
        r["fps_picks"] = int(sum(a.numel() for a in base["fps"]))
        out[name] = r
    return out


def measure(be, n_points=100000, n_queries=20000, fps_levels=2, scene_id=0):
    from pointcloudpdf_amd import synthetic

    xyz = torch.from_numpy(synthetic.make_scene(n_points, scene_id)["coord"]).contiguous()
    snapped = (torch.floor(xyz / 0.02) * 0.02).contiguous()   # every coordinate on a 2 cm lattice: ties everywhere
    prev = be.set_dist_mode(0)
    try:
        res = {"scene": _one(be, xyz, n_queries, fps_levels, 1), "snapped": _one(be, snapped, n_queries, fps_levels, 2)}
    finally:
        be.set_dist_mode(prev)
    res["oracle_mode_after"] = int(be.lib.oracle_get_dist_mode())
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--queries", type=int, default=100000)
    a = ap.parse_args()
    import oracle

    res = measure(oracle.backend(), a.points, a.queries, fps_levels=2)
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)
