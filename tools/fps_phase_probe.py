"""Cycle split of a k_fps_mw round (build with PDFOPS_FPS_PROFILE=1: the work counters then hold kilo-cycles per phase of workgroup 0's
wave 0: candidates | lists | updates | refresh).   PDFOPS_FPS_PROFILE=1 python tools/fps_phase_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native, synthetic
be = _native.hip_backend()
be.collect_fps_stats = True
batch = synthetic.make_batch([100000, 100000], device="cuda")
xyz, off, sizes = batch["coord"], batch["offset"], [100000, 100000]
for lvl in range(3):
    msizes = [s // 4 for s in sizes]
    noff = torch.tensor(msizes, device="cuda").cumsum(0).int()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    print(f"level {lvl + 1}: {ms:.2f} ms  kilo-cycles candidates|lists|updates|refresh (or work counters) = {be.last_fps_stats.tolist()}")
    xyz = xyz[idx.long()].contiguous(); off = noff; sizes = msizes
