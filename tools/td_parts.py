import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pointcloudpdf_amd import synthetic, _native
from pointcloudpdf_amd.geometry import Geometry
b = synthetic.make_batch([100000] * 12, first_scene_id=3, device="cuda")
g = Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
be = _native.hip_backend()
S, Q = g.levels[0], g.levels[1]
idx, _ = g.knn(16, 0, 1)
def T(f, n=3):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
t, (rel4, Z, _) = T(lambda: be.td_tables(S.p, Q.p, idx, Q.o)); print("td_tables kernel + allocs", round(t, 2), "ms")
r0, r1, r2 = rel4[..., 0], rel4[..., 1], rel4[..., 2]
t, mom = T(lambda: torch.stack([(r0 * r0).sum(1), (r0 * r1).sum(1), (r0 * r2).sum(1), (r1 * r1).sum(1), (r1 * r2).sum(1), (r2 * r2).sum(1), r0.sum(1), r1.sum(1), r2.sum(1)], -1)); print("moments", round(t, 2))
t, cs = T(lambda: mom.double().cumsum(0)); print("cumsum", round(t, 2))
