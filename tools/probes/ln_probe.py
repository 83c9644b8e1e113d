"""dense.LayerNorm (csrc/layernorm.hip) forward and backward at config 5's shapes, HIP-event time per call and bytes / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudpdf_amd import dense

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for n, c in ((640032, 48), (160000, 48), (160032, 96), (40002, 96), (40016, 192), (10002, 192), (2502, 384)):
    ln = dense.LayerNorm(c).cuda()
    x = torch.randn(n, c, device="cuda", requires_grad=True)
    g = torch.randn(n, c, device="cuda")
    y = ln(x)
    f = t(lambda: ln(x))
    b = t(lambda: torch.autograd.grad(y, (x, ln.weight, ln.bias), g, retain_graph=True))
    mb = n * c * 4 / 1e6
    print(f"n={n:7d} c={c:4d}  fwd {f:7.1f} us ({2 * mb / f * 1e-3:5.2f} TB/s)   bwd {b:7.1f} us ({3 * mb / b * 1e-3:5.2f} TB/s)", flush=True)
