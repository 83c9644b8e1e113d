import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import max_rel
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
from pointcloudpdf_amd.point_transformer import PointTransformerLayer

def run(fused, C, K, sizes, mode, dtype=torch.float32):
    torch.manual_seed(0)
    batch = synthetic.make_batch(sizes, first_scene_id=50, grid_size=0.25, device="cuda")
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
    layer = PointTransformerLayer(C, C, 8, K).cuda()
    synthetic.fill_parameters_deterministic(layer, seed=3)
    layer = layer.to(dtype); layer.train()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(sum(sizes), C, device="cuda", generator=g)
    if mode == "shift": x = x + 5
    if mode == "relu": x = torch.relu(x * 3 + 1)
    x = x.to(dtype).requires_grad_(True)
    PointTransformerLayer.fused = fused
    y = layer([geom.coord(0), x, geom.offset(0)])
    go = torch.randn(y.shape, device="cuda", generator=g).to(dtype)
    if mode == "smooth": go = go * 0 + torch.linspace(-1, 1, C, device="cuda").to(dtype)[None]
    y.backward(go)
    PointTransformerLayer.fused = True
    return y.detach().double().cpu().numpy(), x.grad.double().cpu().numpy(), {n: p.grad.double().cpu().numpy() for n, p in layer.named_parameters()}

for C, K, sizes in [(32, 8, [2048, 1600]), (64, 16, [512, 400]), (128, 16, [128, 100])]:
    for mode in ["plain", "shift", "relu", "smooth"]:
        yt, gt, pt = run(False, C, K, sizes, mode, torch.float64)
        res = []
        for fused in (True, False):
            y, gx, p = run(fused, C, K, sizes, mode)
            worst = max((max_rel(p[n], pt[n]), n) for n in p if np.abs(pt[n]).max() > 1e-6)
            res.append(f"{'fused' if fused else 'comp '}: y {max_rel(y, yt):.1e} gx {max_rel(gx, gt):.1e} worst-param {worst[0]:.1e} ({worst[1]})")
        print(C, K, sizes, mode, " | ".join(res))
