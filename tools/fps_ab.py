import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pointcloudpdf_amd import _native, synthetic
be = _native.hip_backend()
scenes = 2
batch = synthetic.make_batch([100000] * scenes, device="cuda")
xyz, off = batch["coord"], batch["offset"]
sizes = [100000] * scenes
for lvl in range(4):
    msizes = [s // 4 for s in sizes]
    noff = torch.tensor(msizes, device="cuda").cumsum(0).int()
    idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        idx = be.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
    torch.cuda.synchronize()
    print(f"level {lvl+1}: n={sizes[0]} -> {msizes[0]}: {(time.perf_counter()-t0)/3*1e3:.2f} ms  checksum {int(idx.long().sum())}")
    xyz = xyz[idx.long()].contiguous(); off = noff; sizes = msizes
