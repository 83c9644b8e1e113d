"""Per-level timing of the per-point Linear kernels: torch (rocBLAS) vs csrc/rowlin2.hip.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native

be = _native.hip_backend()
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


print(f"{'shape':>16s} {'op':>10s} {'torch_us':>9s} {'hip_us':>9s} {'GB/s(hip)':>10s}")
for n, c in [(200000, 32), (50000, 64), (12500, 128), (3125, 256), (782, 512)]:
    x = torch.randn(n, c, device="cuda")
    ws = [torch.randn(c, c, device="cuda") for _ in range(3)]
    bs = [torch.randn(c, device="cuda") for _ in range(3)]
    coef = torch.cat([torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda"), torch.zeros(2 * c, device="cuda")])
    gs = [torch.randn(n, c, device="cuda") for _ in range(3)]
    y = torch.empty(n, c, device="cuda")
    mb = n * c * 4 / 1e6
    rows = [
        ("linear", lambda: torch.nn.functional.linear(x, ws[0]), lambda: be.rowlin(x, ws[0], out=y), 2 * mb),
        ("lin+pre+st", lambda: torch.nn.functional.linear(torch.relu(x * coef[:c] + coef[c:2 * c]), ws[0]),
         lambda: be.rowlin(x, ws[0], coef=coef, relu=True, out=y, stats=True), 2 * mb),
        ("qkv", lambda: [torch.nn.functional.linear(x, w, b) for w, b in zip(ws, bs)],
         lambda: be.rowlin_multi([x], ws, bs, coef=coef, relu=True, nout=3), 4 * mb),
        ("dgrad3", lambda: gs[0] @ ws[0] + gs[1] @ ws[1] + gs[2] @ ws[2], lambda: be.rowlin_multi(gs, ws, None, transpose_w=True, nout=1), 4 * mb),
        ("dgrad", lambda: gs[0] @ ws[0], lambda: be.rowlin(gs[0], ws[0], transpose_w=True, out=y), 2 * mb),
        ("wgrad", lambda: gs[0].t() @ x, lambda: be.rowlin_wgrad(gs[0], x, None, False, True), 2 * mb),
        ("wgrad3", lambda: [g.t() @ x for g in gs], lambda: be.rowlin_wgrad_multi(gs, x, coef, True), 4 * mb),
    ]
    for name, ft, fh, traffic in rows:
        tt, th = timeit(ft), timeit(fh)
        print(f"{n:>9d}x{c:<6d} {name:>10s} {tt:9.1f} {th:9.1f} {traffic / th * 1e3:10.0f}")
