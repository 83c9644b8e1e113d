"""Grid kNN timing on the level shapes of a 12-scene group (the grouped pre-pass) -- self queries of levels 1-3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic, _native
from pointcloudpdf_amd.geometry import Geometry
be = _native.hip_backend()
b = synthetic.make_batch([100000] * 12, first_scene_id=3, device="cuda")
geom = Geometry(b["coord"], b["offset"], b["offset_host"])
lvl = 0
for k, stride in [(8, 4), (16, 4), (16, 4)]:
    L = geom.levels[lvl]
    for _ in range(2): be.knn_query(k, L.p, L.p, L.o, L.o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): idx, d = be.knn_query(k, L.p, L.p, L.o, L.o)
    e1.record(); torch.cuda.synchronize()
    ties = (d[:, 1:] == d[:, :-1]).any(1).float().mean().item()
    print(f"level {lvl}: n={L.p.shape[0]} k={k} self kNN {e0.elapsed_time(e1) / 5 * 1e3:9.1f} us   checksum {int(idx.long().sum())}  rows with equal distances among the k: {ties:.4f}")
    lvl, _ = geom.down(lvl, stride)
