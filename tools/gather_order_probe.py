"""Probe: do the forward gathers speed up when the points (hence queries AND table rows) are in Morton order?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic, _native
from tools.ops_roofline import timeit
be = _native.hip_backend()
b = synthetic.make_batch([100000, 100000], first_scene_id=3, device="cuda")
def morton_perm(p):
    q = ((p - p.min(0)[0]) / 0.16).long().clamp(0, 1023)   # 16 cm cells
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return torch.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), stable=True)
def morton_order_of(p, off_host):
    parts, s = [], 0
    for e in off_host:
        parts.append(morton_perm(p[s:e]) + s); s = e
    return torch.cat(parts).int().contiguous()
# natural storage order, queries VISITED in Morton order + XCD-contiguous chunks (pdf_grouping_forward_ordered)
p0 = b["coord"]; off0 = b["offset"].int(); n0 = p0.shape[0]
idx0, _ = be.knn_query(8, p0, p0, off0, off0)
order0 = morton_order_of(p0, b["offset_host"])
ident = torch.arange(n0, dtype=torch.int32, device="cuda")
feat0 = torch.randn(n0, 32, device="cuda")
out_ref = be.grouping_forward(feat0, idx0)
for name, o in (("identity order", ident), ("morton visiting order", order0)):
    out = torch.empty(n0, 8, 32, device="cuda")
    fn = lambda: be._call("grouping_forward_ordered", n0, 8, 32, feat0, idx0, o, out)
    fn(); assert torch.equal(out, out_ref), name
    s = timeit(fn, 20); gb = 4 * n0 * 32 + 4 * n0 * 8 + 4 * n0 * 8 * 32 + 4 * n0
    print(f"ordered kernel, {name:24s} {s * 1e6:8.1f} us  {gb / s / 1e9:8.0f} GB/s  {gb / s / 8e12 * 100:5.1f} %")
for mode in ("natural", "morton"):
    p = b["coord"].clone()
    off = b["offset"].int()
    if mode == "morton":
        parts, s = [], 0
        for e in b["offset_host"]:
            parts.append(morton_perm(p[s:e]) + s); s = e
        p = p[torch.cat(parts)].contiguous()
    n = p.shape[0]
    idx, _ = be.knn_query(8, p, p, off, off)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    feat = torch.randn(n, 32, device="cuda", generator=g)
    f2 = torch.randn(n, 32, device="cuda", generator=g)
    pos = torch.randn(n, 8, 32, device="cuda", generator=g); wgt = torch.randn(n, 8, 4, device="cuda", generator=g)
    go = torch.randn(n, 8, 32, device="cuda", generator=g)
    gout = torch.randn(n, 32, device="cuda", generator=g)
    gb = 4 * n * 32 + 4 * n * 8 + 4 * n * 8 * 32
    for name, fn, nb in (("grouping2 fwd", lambda: be.grouping_forward(feat, idx), gb),
                         ("grouping xyz fwd", lambda: be.group_forward(feat, p, p, idx, True), gb + 24 * n + 12 * n * 8),
                         ("subtraction fwd", lambda: be.subtraction_forward(feat, f2, idx), gb + 4 * n * 32),
                         ("aggregation fwd", lambda: be.aggregation_forward(feat, pos, wgt, idx), 8 * n * 32 + 4 * n * 8 * 36 + 4 * n * 8),
                         ("grouping2 bwd (inverse)", lambda: be.grouping_backward(go, idx, n), gb),
                         ("aggregation bwd", lambda: be.aggregation_backward(feat, pos, wgt, idx, gout), 0)):
        s = timeit(fn, 20)
        print(f"{mode:8s} {name:26s} {s * 1e6:8.1f} us  {nb / s / 1e9:8.0f} GB/s  {nb / s / 8e12 * 100:5.1f} %")
