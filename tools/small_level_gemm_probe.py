"""The Linear products of levels 4-5 (3,124 x 256, 780 x 512 rows) through csrc/rowlin.hip against the library GEMM of the same shape:
how far the streaming skinny-GEMM kernels (built for 10^5 rows x 32-64 channels) are from a tiled GEMM where the rows are few."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native
be = _native.hip_backend()
torch.backends.cuda.matmul.allow_tf32 = False


def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for n, k, o in [(780, 512, 512), (3124, 256, 256), (12496, 128, 128), (49984, 64, 64), (200000, 32, 32), (780, 512, 1536), (3124, 256, 768)]:
    x = torch.randn(n, k, device="cuda"); w = torch.randn(o, k, device="cuda") / k ** 0.5; g = torch.randn(n, o, device="cuda")
    gf = 2.0 * n * k * o / 1e9
    a = t(lambda: be.rowlin(x, w, None)); b = t(lambda: torch.nn.functional.linear(x, w))
    c = t(lambda: be.rowlin(g, w, transpose_w=True)); d = t(lambda: g @ w)
    e = t(lambda: be.rowlin_wgrad(g, x, None, False, False)); f = t(lambda: g.t() @ x)
    print(f"n={n:6d} k={k:4d} o={o:4d} ({gf:5.2f} GFLOP): fwd {a:6.1f} us (lib {b:6.1f})  dgrad {c:6.1f} (lib {d:6.1f})  wgrad {e:6.1f} (lib {f:6.1f})   fwd {gf / a * 1e3:5.1f} TFLOP/s", flush=True)
