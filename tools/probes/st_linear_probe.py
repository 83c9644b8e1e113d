"""The Linear layers of config 5's StratifiedTransformer (qkv / proj / fc1 / fc2 at the four levels, 2 x 80k points) piece by piece:
library GEMM compositions (what stratified._Linear / dense._LinearSplitK issue) against the package's own matrix-core kernels
(csrc/rowlin*.hip through the backend), HIP-event time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from pointcloudpdf_amd import _native, dense

be = _native.hip_backend()
dev = "cuda"
g_ = torch.Generator(device=dev); g_.manual_seed(1)

def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def splitk_wgrad(g, x):
    n, ci = x.shape; co = g.shape[1]; C = dense._CHUNK
    s = n // C; body = s * C
    gw = torch.bmm(g[:body].view(s, C, co).transpose(1, 2), x[:body].view(s, C, ci)).sum(0)
    if body < n: gw = gw + g[body:].t() @ x[body:]
    gb = torch.bmm(dense._ones_row(g.device, g.dtype).expand(s, 1, C), g[:body].view(s, C, co)).sum(0).view(co)
    if body < n: gb = gb + g[body:].sum(0)
    return gw, gb

tot = {"lib": 0.0, "own": 0.0}
for n, c, blocks in ((160000, 48, 2), (40002, 96, 2), (10002, 192, 6), (2502, 384, 2)):
    for name, ci, co in (("qkv", c, 3 * c), ("proj", c, c), ("fc1", c, 4 * c), ("fc2", 4 * c, c)):
        x = torch.randn(n, ci, device=dev, generator=g_); w = torch.randn(co, ci, device=dev, generator=g_) * 0.05
        b = torch.randn(co, device=dev, generator=g_); g = torch.randn(n, co, device=dev, generator=g_)
        r = dict(
            fwd_lib=t(lambda: F.linear(x, w, b)), fwd_own=t(lambda: be.rowlin(x, w, b)),
            dgrad_lib=t(lambda: g @ w), dgrad_own=t(lambda: be.rowlin(g, w, None, transpose_w=True)),
            wgrad_lib=t(lambda: splitk_wgrad(g, x)), wgrad_own=t(lambda: be.rowlin_wgrad(g, x, None, False, True)))
        gw, gb = splitk_wgrad(g, x); dw, db = be.rowlin_wgrad(g, x, None, False, True)
        err = float((gw - dw).abs().max() / gw.abs().max()), float((gb - db).abs().max() / gb.abs().max())
        tot["lib"] += blocks * (r["fwd_lib"] + r["dgrad_lib"] + r["wgrad_lib"]); tot["own"] += blocks * (r["fwd_own"] + r["dgrad_own"] + r["wgrad_own"])
        print(f"n={n:6d} {name:4s} {ci:4d}->{co:4d}  fwd {r['fwd_lib']:7.1f} | {r['fwd_own']:7.1f}   dgrad {r['dgrad_lib']:7.1f} | {r['dgrad_own']:7.1f}   "
              f"wgrad+bias {r['wgrad_lib']:7.1f} | {r['wgrad_own']:7.1f} us   (library | own)  wgrad rel err {err[0]:.1e} {err[1]:.1e}", flush=True)
print(f"per step (x blocks): library {tot['lib'] / 1e3:.2f} ms, own {tot['own'] / 1e3:.2f} ms")
