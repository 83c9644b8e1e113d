"""The window-attention backward kernels (csrc/window_attention_bwd.hip) one by one on the edge tables of config 5's own batch
(2 x 80k S3DIS-shaped points, the model's four levels): HIP-event time per call of every piece of `window_attention_core_backward`.
    python tools/probes/wa_bwd_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloudpdf_amd import _native, engine, synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
step = engine.OpenSegStep(backbone="ST-v1m1", loss_weight=0.008).to(dev)
bb = step.model.backbone
b = synthetic.make_batch([80000, 80000], first_scene_id=0, device=dev)
geom = bb.make_geometry(b["coord"], b["offset"], b["offset_host"]).precompute(bb.layers_by_level())
be = _native.hip_backend()
g = torch.Generator(device=dev); g.manual_seed(3)

def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

total = 0.0
xyzs, offs = {0: b["coord"]}, {0: b["offset"]}
for l in range(3):
    xyzs[l + 1], offs[l + 1] = geom.neighbors[("td", l)][0], geom.samples[("down", l)][1]
for level, layer in bb.layers_by_level().items():
    index0, index1, offsets, n_max, rel = geom.windows[level][0]
    lo, hi = xyzs[level].min(0).values, xyzs[level].max(0).values
    kf = be.window_keys(xyzs[level], offs[level], lo, hi, layer.window_size, 0)[0]
    worder = torch.sort(kf, stable=True)[1].int()      # the owners window by window
    attn_mod = layer.blocks[0].attn
    tq, tk, tv = attn_mod.relative_pos_query_table.detach(), attn_mod.relative_pos_key_table.detach(), attn_mod.relative_pos_value_table.detach()
    L, h, d, _ = tq.shape
    c, n, m = h * d, offsets.shape[0] - 1, index1.shape[0]
    qkv = torch.randn(n, 3 * c, device=dev, generator=g)
    go = torch.randn(n, c, device=dev, generator=g)
    out, attn = be.window_attention_core(qkv, index1, offsets, tq, tk, tv, rel, attn_mod.scale)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    key_off, key_edge, key_q, key_rel = _native.window_csc(index1, offsets, rel, n_keys=n)
    ga = torch.empty((m, h), device=dev)
    gqkv = torch.empty_like(qkv)
    attn_key = attn.index_select(0, key_edge)
    gsm = be.segment_softmax_backward(attn, torch.randn(m, h, device=dev, generator=g), offsets)
    g_key = gsm.index_select(0, key_edge)
    klen = (key_off[1:] - key_off[:-1])[worder.long()]                      # rows of the key side: long (FPS keys: a whole coarse window refers to them) and short
    lorder = worder[torch.sort(-klen, stable=True)[1]].contiguous()         # longest first, window order inside equal lengths
    qlen = (offsets[1:] - offsets[:-1])[worder.long()]
    qorder = worder[torch.sort(-qlen, stable=True)[1]].contiguous()
    rows = [
        ("rows fwd out, length then window order", lambda: be._wa_rows(n, h, d, L, offsets, None, index1, rel, attn, v, tv, out, ldx=3 * c, order=qorder)),
        ("logits_fwd", lambda: be._call("wa_logits_forward", n, m, h, d, L, q, k, 3 * c, 1.0, offsets, index1, tq, tk, rel, ga)),
        ("rows fwd out (CSR, rows+table)", lambda: be._wa_rows(n, h, d, L, offsets, None, index1, rel, attn, v, tv, out, ldx=3 * c)),
        ("grad_attn", lambda: be._call("wa_grad_attn", n, m, h, d, L, go, c, offsets, index1, v, 3 * c, tv, rel, ga)),
        ("edge scalars in key order (M, h)", lambda: be._wa_permute(attn, key_edge)),
        ("rows grad_v (CSC, rows)", lambda: be._wa_rows(n, h, d, 0, key_off, None, key_q, None, attn_key, go, None, gqkv[:, 2 * c:], ldo=3 * c)),
        ("rows grad_v (CSC, edge ids)", lambda: be._wa_rows(n, h, d, 0, key_off, key_edge, key_q, None, attn, go, None, gqkv[:, 2 * c:], ldo=3 * c)),
        ("rows grad_k (CSC, edge ids)", lambda: be._wa_rows(n, h, d, L, key_off, key_edge, key_q, key_rel, gsm, q, tk, gqkv[:, c:2 * c], ldx=3 * c, xscale=0.25, ldo=3 * c)),
        ("table gtk (key order, edge ids)", lambda: be._wa_table_grad(n, h, d, L, key_off, key_edge, key_rel, gsm, k, qkv, ldx=3 * c)),
        ("logits_fwd, window order", lambda: be._call("wa_logits_forward_ordered", n, m, h, d, L, q, k, 3 * c, 1.0, offsets, index1, tq, tk, rel, ga, worder)),
        ("grad_attn, window order", lambda: be._call("wa_grad_attn_ordered", n, m, h, d, L, go, c, offsets, index1, v, 3 * c, tv, rel, ga, worder)),
        ("rows fwd out, window order", lambda: be._wa_rows(n, h, d, L, offsets, None, index1, rel, attn, v, tv, out, ldx=3 * c, order=worder)),
        ("rows grad_v (CSC), window order", lambda: be._wa_rows(n, h, d, 0, key_off, None, key_q, None, attn_key, go, None, gqkv[:, 2 * c:], ldo=3 * c, order=worder)),
        ("rows grad_k (CSC), window order", lambda: be._wa_rows(n, h, d, L, key_off, None, key_q, key_rel, g_key, q, tk, gqkv[:, c:2 * c], ldx=3 * c, xscale=0.25, ldo=3 * c, order=worder)),
        ("rows grad_k (CSC), length then window order", lambda: be._wa_rows(n, h, d, L, key_off, None, key_q, key_rel, g_key, q, tk, gqkv[:, c:2 * c], ldx=3 * c, xscale=0.25, ldo=3 * c, order=lorder)),
        ("rows grad_v (CSC), length then window order", lambda: be._wa_rows(n, h, d, 0, key_off, None, key_q, None, attn_key, go, None, gqkv[:, 2 * c:], ldo=3 * c, order=lorder)),
        ("table gtv (CSR)", lambda: be._wa_table_grad(n, h, d, L, offsets, None, rel, attn, go, qkv)),
        ("softmax_bwd", lambda: be.segment_softmax_backward(attn, ga, offsets)),
        ("rows grad_q (CSR, rows+table)", lambda: be._wa_rows(n, h, d, L, offsets, None, index1, rel, gsm, k, tq, gqkv[:, :c], ldx=3 * c, ldo=3 * c, oscale=0.25)),
        ("rows grad_k (CSC, rows+table)", lambda: be._wa_rows(n, h, d, L, key_off, None, key_q, key_rel, g_key, q, tk, gqkv[:, c:2 * c], ldx=3 * c, xscale=0.25, ldo=3 * c)),
        ("table gtq (CSR)", lambda: be._wa_table_grad(n, h, d, L, offsets, None, rel, gsm, q, qkv, ldx=3 * c, xscale=0.25)),
        ("table gtk (key order)", lambda: be._wa_table_grad(n, h, d, L, key_off, None, key_rel, g_key, k, qkv, ldx=3 * c)),
    ]
    print(f"level {level}: N={n} M={m} C={c} h={h} L={L} n_max={n_max} mean row {m / n:.1f}  blocks={layer.depth}")
    for name, fn in rows:
        us = t(fn)
        if name not in ("logits_fwd", "rows fwd out (CSR, rows+table)") and "edge ids" not in name and "window order" not in name and "length" not in name:
            total += us * layer.depth
        print(f"    {name:34s} {us:9.1f} us", flush=True)
print(f"backward pieces x blocks per level: {total / 1e3:.2f} ms per step")
