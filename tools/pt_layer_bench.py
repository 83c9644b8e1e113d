"""Fused PointTransformerLayer (csrc/fused_layer.hip) forward+backward alone, on the real level shapes of a 2 x 100k batch.
Run on the GPU box; with rocprofv3 around it for per-kernel times / PMC counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
from pointcloudpdf_amd.point_transformer import PointTransformerLayer

reps = int(os.environ.get("REPS", "5"))
levels = [int(x) for x in os.environ.get("LEVELS", "0,1,2,3").split(",")]
batch = synthetic.make_batch([100000, 100000], device="cuda")
geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
planes, nsample = [32, 64, 128, 256, 512], [8, 16, 16, 16, 16]
for lv in levels:
    c, k = planes[lv], nsample[lv]
    layer = PointTransformerLayer(c, c, 8, k).cuda().train()
    p, o = geom.coord(lv), geom.offset(lv)
    n = p.shape[0]
    if os.environ.get("MORTON"):   # experiment: spatially coherent point order inside every scene
        q = ((p - p.min(0).values) / 0.05).long().clamp_(0, 1023)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
        key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        scene = torch.bucketize(torch.arange(n, device="cuda"), o.long(), right=True)
        p = p[torch.argsort(key + (scene << 32))].contiguous()
        g2 = Geometry(p, o, geom.offset_host(lv))   # tagged coordinates again: the kNN table comes from the memo, as in the natural order
        p, o = g2.coord(0), g2.offset(0)
        g2.knn(k, 0, 0)
    x = torch.randn(n, c, device="cuda")
    xq, xk, xv = [torch.randn(n, c, device="cuda", requires_grad=True) for _ in range(3)]
    go = torch.randn(n, c, device="cuda")
    for it in range(reps + 2):
        if it == 2:
            torch.cuda.synchronize()
            e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = layer.attend(p, x, o, xq, xk, xv)
        e[1].record()
        y.backward(go)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"level {lv}: n={n} c={c} k={k}  fwd {tf / reps * 1e3:8.1f} us  bwd {tb / reps * 1e3:8.1f} us", flush=True)
