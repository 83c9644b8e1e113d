import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pointcloudpdf_amd import _native
be = _native.hip_backend()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
n = 200000
for k, o, bias in [(6, 32, False), (32, 13, True), (32, 1, True), (1024, 512, True)]:
    nn = n if k != 1024 else 780
    x = torch.randn(nn, k, device="cuda"); w = torch.randn(o, k, device="cuda"); b = torch.randn(o, device="cuda") if bias else None
    g = torch.randn(nn, o, device="cuda")
    print(f"n={nn} k={k} o={o}: fwd {t(lambda: be.rowlin(x, w, b)):7.1f} us  dgrad {t(lambda: be.rowlin(g, w, transpose_w=True)):7.1f} us  wgrad {t(lambda: be.rowlin_wgrad(g, x, None, False, bias)):7.1f} us | torch fwd {t(lambda: torch.nn.functional.linear(x, w, b)):7.1f} dgrad {t(lambda: g @ w):7.1f} wgrad {t(lambda: g.t() @ x):7.1f}")
