#!/usr/bin/env python3
"""GPU probe: time the FPS kernels per level of the headline config and print the bucketed kernel's work counters."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native, synthetic

be = _native.hip_backend()
batch = synthetic.make_batch([100000, 100000], device="cuda")
xyz, off = batch["coord"], batch["offset"]
sizes = [100000, 100000]
for lvl in range(4):
    msizes = [s // 4 for s in sizes]
    noff = torch.tensor([msizes[0], msizes[0] + msizes[1]], dtype=torch.int32, device="cuda")
    res = {}
    for mode in ("bucketed", "plain") if lvl > 0 or "--plain" in sys.argv else ("bucketed",):
        be.fps_mode = mode
        be.collect_fps_stats = mode == "bucketed"
        idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / 3 * 1e3
        if mode == "bucketed":
            st = be.last_fps_stats.tolist()
    print(f"level {lvl + 1}: n={sizes} -> m={msizes}  " + "  ".join(f"{k} {v:.2f} ms" for k, v in res.items()),
          " stats [updates, supers, samples, buckets] per scene:", st,
          f" updates/sample {st[0][0] / max(st[0][2], 1):.2f} supers/sample {st[0][1] / max(st[0][2], 1):.2f}", flush=True)
    xyz = xyz.index_select(0, idx.long()).contiguous()
    off = noff
    sizes = msizes
