#!/usr/bin/env python3
"""GPU probe: samples-per-round FPS (PDFOPS_FPS_K = 1 / 4 / 8) -- identical indices, time per level of the headline config,
rounds per sample.  The env knob is read at every launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native, synthetic

be = _native.hip_backend()
be.collect_fps_stats = True
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = synthetic.make_batch([100000] * scenes, device="cuda")
xyz, off = batch["coord"], batch["offset"]
sizes = [100000] * scenes
for lvl in range(4):
    msizes = [s // 4 for s in sizes]
    noff = torch.tensor(msizes, device="cuda").cumsum(0).int()
    ref = None
    line = []
    for k, nw, mw in (("1", "4", "0"), ("8", "8", "8"), ("8", "8", "16")):
        os.environ["PDFOPS_FPS_K"] = k; os.environ["PDFOPS_FPS_NW"] = nw; os.environ["PDFOPS_FPS_MW"] = mw
        idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        st = be.last_fps_stats.tolist()[0]
        if ref is None:
            ref = idx.clone()
        same = bool(torch.equal(ref, idx))
        rounds = st[3] if k != "1" else st[2]
        line.append(f"K={k},NW={nw},MW={mw}: {ms:7.2f} ms same={same} samples/round {st[2] / max(rounds, 1):.2f} bucket-updates/sample {st[0] / max(st[2], 1):.2f}")
    print(f"level {lvl + 1}: n={sizes[0]} -> m={msizes[0]} x{scenes} | " + " | ".join(line), flush=True)
    xyz = xyz.index_select(0, ref.long()).contiguous()
    off = noff
    sizes = msizes
