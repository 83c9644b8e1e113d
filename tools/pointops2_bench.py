#!/usr/bin/env python3
"""GPU probe: the pointops2 window-attention ops at the reference's own test-script shape (N = 35000, M = 800000, C = 96, h = 6,
libs/pointops2/functions/test_attention_op_step1_v2.py:13-18), forward and backward, HIP events; algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import _native

be = _native.hip_backend()
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)
n, h, d, L, m = 35000, 6, 16, 48, 800000
C = h * d
index0, _ = torch.sort(torch.randint(0, n, (m,), device=dev, generator=g))
off = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), index0.bincount(minlength=n).cumsum(0)]).int()
i1 = torch.randint(0, n, (m,), device=dev, generator=g).int()
rel = torch.randint(0, L, (m, 3), device=dev, generator=g).int()
q, k, v = (torch.rand(n, h, d, device=dev, generator=g) for _ in range(3))
tq, tk, tv = (torch.rand(L, h, d, 3, device=dev, generator=g) for _ in range(3))
n_max = int((off[1:] - off[:-1]).max())
attn = be.attention_step1_v2(q, k, i1, off, n_max)
go_e = torch.rand(m, h, device=dev, generator=g)
go_n = torch.rand(n, h, d, device=dev, generator=g)


def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


row = 4 * m * C   # gathered rows
cases = [
    ("attention_step1_v2 fwd", 4 * n * C + row + 8 * m + 4 * m * h, lambda: be.attention_step1_v2(q, k, i1, off, n_max)),
    ("attention_step1_v2 bwd", 4 * n * C * 2 + 2 * row + 8 * m + 4 * m * h, lambda: be.attention_step1_v2_backward(go_e, q, k, i1, off, n_max)),
    ("dot_prod_with_idx_v3 fwd", 4 * n * C + row + 20 * m + 4 * m * h, lambda: be.dot_prod_with_idx_v3(q, off, n_max, k, i1, tq, tk, rel)),
    ("dot_prod_with_idx_v3 bwd", 4 * n * C * 2 + 2 * row + 20 * m + 4 * m * h, lambda: be.dot_prod_with_idx_v3_backward(go_e, q, off, n_max, k, i1, tq, tk, rel)),
    ("attention_step2_with_rel_pos_value_v2 fwd", 4 * n * C + row + 20 * m + 4 * m * h, lambda: be.attention_step2_with_rel_pos_value_v2(attn, v, off, n_max, i1, tv, rel)),
    ("attention_step2_with_rel_pos_value_v2 bwd", 4 * n * C * 2 + 2 * row + 20 * m + 8 * m * h, lambda: be.attention_step2_with_rel_pos_value_v2_backward(go_n, attn, v, off, n_max, i1, tv, rel)),
]
print(f"N={n} M={m} C={C} h={h} L={L} n_max={n_max}")
for name, nbytes, fn in cases:
    us = t(fn)
    print(f"{name:46s} {us:9.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
