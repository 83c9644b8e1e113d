import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
from pointcloudpdf_amd.point_transformer import Bottleneck, PointTransformerLayer
def run(fused, C, K, sizes):
    batch = synthetic.make_batch(list(sizes), first_scene_id=60, grid_size=0.25, device="cuda")
    torch.manual_seed(0)
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
    blk = Bottleneck(C, C, 8, K).cuda()
    synthetic.fill_parameters_deterministic(blk, seed=7)
    blk.train(True)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(sum(sizes), C, device="cuda", generator=g).requires_grad_(True)
    Bottleneck.matrix_core = False
    PointTransformerLayer.fused = fused
    y = blk([geom.coord(0), x, geom.offset(0)])[1]
    y.backward(torch.randn(y.shape, device="cuda", generator=g))
    out = {"y": y.detach().cpu().numpy(), "gx": x.grad.cpu().numpy()}
    out.update({"g_" + n: p.grad.cpu().numpy() for n, p in blk.named_parameters() if p.grad is not None})
    return out
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
C = int(os.environ.get("C", "256"))
import ast
for sizes in ast.literal_eval(os.environ.get("SIZES", "[(300, 260), (900,), (560,), (3000,)]")):
    a, b = run(True, C, 16, sizes), run(False, C, 16, sizes)
    worst = sorted(((rel(a[k], b[k]), k) for k in a if np.abs(b[k]).max() > 1e-2), reverse=True)[:3]
    print(sizes, f"y:{rel(a['y'], b['y']):.1e}", " ".join(f"{k}:{v:.1e}" for v, k in worst))
if os.environ.get("DETAIL"):
    sizes = ast.literal_eval(os.environ["DETAIL"])
    a, b = run(True, C, 16, sizes), run(False, C, 16, sizes)
    for k in ["g_bn2.bias", "g_transformer.linear_v.weight", "gx", "y"]:
        d = np.abs(a[k] - b[k]); flat = d.reshape(-1)
        top = np.argsort(flat)[::-1][:6]
        print(k, a[k].shape, "n>1e-4*max:", int((flat > 1e-4 * np.abs(b[k]).max()).sum()), "of", flat.size, " top idx", [np.unravel_index(t, d.shape) for t in top], " top diffs", flat[top])
if os.environ.get("F64"):
    sizes = ast.literal_eval(os.environ["F64"])
    def run64(sizes):
        batch = synthetic.make_batch(list(sizes), first_scene_id=60, grid_size=0.25, device="cuda")
        torch.manual_seed(0)
        geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
        blk = Bottleneck(C, C, 8, 16).cuda()
        synthetic.fill_parameters_deterministic(blk, seed=7)
        blk = blk.double(); blk.train(True)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(sum(sizes), C, device="cuda", generator=g).double().requires_grad_(True)
        Bottleneck.matrix_core = False; PointTransformerLayer.fused = False
        y = blk([geom.coord(0), x, geom.offset(0)])[1]
        t_std = None
        y.backward(torch.randn(y.shape, device="cuda", generator=g).double())
        out = {"y": y.detach().cpu().numpy(), "gx": x.grad.cpu().numpy()}
        out.update({"g_" + n: p.grad.cpu().numpy() for n, p in blk.named_parameters() if p.grad is not None})
        return out
    r = run64(sizes); a = run(True, C, 16, sizes); b = run(False, C, 16, sizes)
    for k in ["y", "gx", "g_bn2.bias", "g_transformer.linear_v.weight", "g_linear1.weight"]:
        print(f"{k:34s} fused-vs-f64 {rel(a[k], r[k]):.2e}   unfused32-vs-f64 {rel(b[k], r[k]):.2e}")
if os.environ.get("TFWD"):
    sizes = ast.literal_eval(os.environ["TFWD"])
    outs = {}
    for fused in (True, False):
        batch = synthetic.make_batch(list(sizes), first_scene_id=60, grid_size=0.25, device="cuda")
        geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
        blk = Bottleneck(C, C, 8, 16).cuda()
        synthetic.fill_parameters_deterministic(blk, seed=7); blk.train(True)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(sum(sizes), C, device="cuda", generator=g).requires_grad_(True)
        Bottleneck.matrix_core = False; PointTransformerLayer.fused = fused
        cap = {}
        hk = blk.transformer.register_forward_hook(lambda m, i, o: cap.__setitem__("t", o.detach().cpu().numpy()))
        y = blk([geom.coord(0), x, geom.offset(0)])[1]
        hk.remove()
        outs[fused] = cap["t"]
    d = np.abs(outs[True] - outs[False])
    print("t shape", d.shape, "max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "scale", np.abs(outs[False]).max())
    ch = d.max(0); print("worst channels", np.argsort(ch)[::-1][:5], np.sort(ch)[::-1][:5])
    print("std of t per worst channel", outs[False][:, np.argsort(ch)[::-1][:3]].std(0), " median std", np.median(outs[False].std(0)))
if os.environ.get("BN2"):
    sizes = ast.literal_eval(os.environ["BN2"])
    a, b = run(True, C, 16, sizes), run(False, C, 16, sizes)
    for k in ["g_bn2.bias", "g_bn2.weight", "g_bn3.bias", "g_transformer.linear_w.3.bias", "g_transformer.linear_w.3.weight", "g_transformer.linear_w.0.bias"]:
        d = np.abs(a[k] - b[k]); i = int(d.argmax())
        print(k, "argmax", i, "fused", a[k].reshape(-1)[i], "unfused", b[k].reshape(-1)[i], " second worst", np.sort(d.reshape(-1))[-2])
