import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch, numpy as np
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
from pointcloudpdf_amd.point_transformer import Bottleneck
def run(mc, C=256, K=16, sizes=(300, 260), dt=torch.float32):
    batch = synthetic.make_batch(list(sizes), first_scene_id=60, grid_size=0.25, device="cuda")
    torch.manual_seed(0)
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
    blk = Bottleneck(C, C, 8, K).cuda()
    synthetic.fill_parameters_deterministic(blk, seed=7)
    blk.train(True)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(sum(sizes), C, device="cuda", generator=g).requires_grad_(True)
    Bottleneck.matrix_core = mc
    y = blk([geom.coord(0), x, geom.offset(0)])[1]
    y.backward(torch.randn(y.shape, device="cuda", generator=g))
    out = {"gx": x.grad.cpu().numpy()}
    out.update({"g_" + n: p.grad.cpu().numpy() for n, p in blk.named_parameters() if p.grad is not None})
    return out
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
a1, a2, b1, b2 = run(True), run(True), run(False), run(False)
for k in ["gx", "g_transformer.linear_v.weight", "g_transformer.linear_q.weight", "g_linear1.weight", "g_linear3.weight"]:
    print(f"{k:34s} mc-vs-mc {rel(a1[k], a2[k]):.2e}  ref-vs-ref {rel(b1[k], b2[k]):.2e}  mc-vs-ref {rel(a1[k], b1[k]):.2e}")
if os.environ.get("DUMP"):
    np.savez(os.environ["DUMP"], **b1)
if os.environ.get("CMP"):
    ref = np.load(os.environ["CMP"])
    for k in b1:
        print(f"   vs dump {k:36s} {rel(b1[k], ref[k]):.2e}   scale {np.abs(ref[k]).max():.2e}")
