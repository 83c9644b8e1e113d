"""Per-op HBM roofline of the pointops drop-in at the headline level sizes (SURVEY.md 8d byte counts):
2 scenes x 100k points, level 1 (c=32, k=8) and level 2 (c=64, k=16).  Each op is called through the HipBackend
method a `pointops.*` call lands on (allocation of the outputs included, as a user sees it), timed with HIP events on
torch's current stream over `--iters` back-to-back calls.

    python tools/ops_roofline.py [--iters 20] [--json gpurun_out/ops_roofline.json]
"""
import os
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic, _native
from pointcloudpdf_amd.geometry import Geometry

VALU_PEAK_FLOPS = 157.3e12   # fp32 vector peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK = 8000.0


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    rows = run(a.iters, a.only, verbose=True)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        json.dump(dict(peak_GBps=PEAK, iters=a.iters, rows=rows), open(a.json, "w"), indent=1)


def run(iters=20, only="", verbose=False, level2=True, references=True):
    """-> list of dicts (op, us, algorithmic_MB, GBps, frac).  Used by bench.py for its "roofline_ops" entry."""
    class A: pass
    a = A(); a.iters, a.only = iters, only
    be = _native.hip_backend()
    dev = "cuda"
    b = synthetic.make_batch([100000, 100000], first_scene_id=3, device=dev)
    geom = Geometry(b["coord"], b["offset"], b["offset_host"])
    L1 = geom.levels[0]
    l2, _ = geom.down(0, 4)
    L2 = geom.levels[l2]
    B = L1.o.shape[0]
    g = torch.Generator(device=dev); g.manual_seed(7)
    rows = []

    def add(name, nbytes, fn, pairs=None, counted=None):
        if a.only and a.only not in name:
            return
        s = timeit(fn, a.iters)
        gbs = nbytes / s / 1e9
        rows.append(dict(op=name, us=s * 1e6, algorithmic_MB=nbytes / 1e6, GBps=gbs, frac=gbs / PEAK))
        if pairs is not None:
            # kNN is VALU / latency bound, not HBM bound.  The grid prunes by design, so the VALU-side figure is quoted on the
            # candidate distances the kernel REALLY evaluates (a counting build of the same launch, pdf_knn_query_ws_counted: 8 flop
            # each -- 3 sub, 3 mul, 2 add) against the fp32 vector peak (157.3 TFLOP/s, MI355X_MICROARCH.md): a fraction <= 1.  The
            # brute-force definition's m_b * n_b pairs per scene are kept as `bruteforce_pairs` (what the reference kernel evaluates).
            ev = counted() if counted is not None else None
            rows[-1].update(bruteforce_pairs=pairs)
            if ev is not None:
                rows[-1].update(evaluated_pairs=float(ev), pair_evals_per_s=ev / s, valu_frac_evaluated=8.0 * ev / s / VALU_PEAK_FLOPS,
                                pruning_factor=pairs / max(ev, 1))
        if verbose:
            print(f"{name:58s} {s * 1e6:9.1f} us  {nbytes / 1e6:8.1f} MB  {gbs:8.1f} GB/s  {100 * gbs / PEAK:5.1f} % of HBM peak", flush=True)

    # ---- kNN (12N + 12M + 8B + 8Mk)
    def knn_bytes(n, m, k):
        return 12 * n + 12 * m + 8 * B + 8 * m * k
    n1, n2 = L1.p.shape[0], L2.p.shape[0]
    def pair_count(S, Q):   # sum over the scenes of (queries of the scene) x (source points of the scene)
        so, qo = [0] + list(S.o_host), [0] + list(Q.o_host)
        return float(sum((so[i + 1] - so[i]) * (qo[i + 1] - qo[i]) for i in range(len(S.o_host))))
    cnt = lambda k, S, Q: (lambda: be.knn_query_counted(k, S.p, Q.p, S.o, Q.o)[2])
    add("knn_query L1 self k=8", knn_bytes(n1, n1, 8), lambda: be.knn_query(8, L1.p, L1.p, L1.o, L1.o), pair_count(L1, L1), cnt(8, L1, L1))
    add("knn_query L2 self k=16", knn_bytes(n2, n2, 16), lambda: be.knn_query(16, L2.p, L2.p, L2.o, L2.o), pair_count(L2, L2), cnt(16, L2, L2))
    add("knn_query L1->L2 down k=16", knn_bytes(n1, n2, 16), lambda: be.knn_query(16, L1.p, L2.p, L1.o, L2.o), pair_count(L1, L2), cnt(16, L1, L2))
    add("knn_query L2->L1 interp k=3", knn_bytes(n2, n1, 3), lambda: be.knn_query(3, L2.p, L1.p, L2.o, L1.o), pair_count(L2, L1), cnt(3, L2, L1))
    # the tables as the model gets them: from the batch's Geometry (with the query level's Morton visiting order and, on first use
    # in a backward, the inverse table attached to the idx tensor)
    idx1, _ = geom.knn(8, 0, 0)
    idx2, _ = geom.knn(16, l2, l2)
    idxd, _ = geom.knn(16, 0, l2)
    idx3, d3 = geom.knn(3, l2, 0)
    w3 = be.interpolation_weights(d3)

    for (tag, L, idx, c, k) in (("L1 c=32 k=8", L1, idx1, 32, 8), ("L2 c=64 k=16", L2, idx2, 64, 16))[:2 if level2 else 1]:
        n = L.p.shape[0]
        feat = torch.randn(n, c, device=dev, generator=g)
        # grouping fwd: 4Nc + 4Mk + 4Mkc  (+ 12N + 12M + 12Mk with xyz)
        gb = 4 * n * c + 4 * n * k + 4 * n * k * c
        add(f"grouping2 fwd {tag}", gb, lambda: be.grouping_forward(feat, idx))
        go = torch.randn(n, k, c, device=dev, generator=g)
        add(f"grouping2 bwd {tag}", gb, lambda: be.grouping_backward(go, idx, n))
        add(f"grouping(with_xyz) fwd {tag}", gb + 12 * n + 12 * n + 12 * n * k, lambda: be.group_forward(feat, L.p, L.p, idx, True))
        if os.environ.get("PDFOPS_ROOFLINE_EXTRA"):   # the same op visiting the query points in index order (a plain copy of the table carries no order)
            idx_plain = idx.clone()
            add(f"grouping(with_xyz) fwd {tag} [index order]", gb + 12 * n + 12 * n + 12 * n * k, lambda: be.group_forward(feat, L.p, L.p, idx_plain, True))
            add(f"grouping2 fwd {tag} [index order]", gb, lambda: be.grouping_forward(feat, idx_plain))
        gox = torch.randn(n, k, c + 3, device=dev, generator=g)
        add(f"grouping(with_xyz) bwd {tag}", 4 * n * k * (c + 3) + 4 * n * k + 4 * n * c, lambda: be.group_backward(gox, idx, n, c, True))
        # subtraction fwd: 8Nc + 4Nk + 4Nkc ; bwd: 4Nkc + 4Nk + 8Nc
        sb = 8 * n * c + 4 * n * k + 4 * n * k * c
        f2 = torch.randn(n, c, device=dev, generator=g)
        add(f"subtraction fwd {tag}", sb, lambda: be.subtraction_forward(feat, f2, idx))
        add(f"subtraction bwd {tag}", sb, lambda: be.subtraction_backward(idx, go))
        # aggregation fwd: 4Nc + 4Nkc + 4Nkw + 4Nk + 4Nc ; bwd: reads the same + g_out, writes g_in, g_pos, g_w
        wc = c // 8
        pos = torch.randn(n, k, c, device=dev, generator=g)
        wgt = torch.randn(n, k, wc, device=dev, generator=g)
        ab = 4 * n * c + 4 * n * k * c + 4 * n * k * wc + 4 * n * k + 4 * n * c
        add(f"aggregation fwd {tag}", ab, lambda: be.aggregation_forward(feat, pos, wgt, idx))
        gout = torch.randn(n, c, device=dev, generator=g)
        add(f"aggregation bwd {tag}", ab + 4 * n * c + 4 * n * k * c + 4 * n * k * wc, lambda: be.aggregation_backward(feat, pos, wgt, idx, gout))

    # ---- interpolation L2 -> L1 (4 Nc c + 8 Nf k + 4 Nf c)
    for c in (32, 64):
        fc = torch.randn(n2, c, device=dev, generator=g)
        ib = 4 * n2 * c + 8 * n1 * 3 + 4 * n1 * c
        add(f"interpolation2 fwd L2->L1 c={c} k=3", ib, lambda: be.interpolation_forward(fc, idx3, w3))
        gf = torch.randn(n1, c, device=dev, generator=g)
        add(f"interpolation2 bwd L2->L1 c={c} k=3", ib, lambda: be.interpolation_backward(gf, idx3, w3, n2))

    # ---- reference points: device copy / fill / read of 256 MB
    if references:
        src = torch.empty(64 * 1024 * 1024, device=dev); dst = torch.empty_like(src)
        add("copy 256 MB (d2d, read+write = 512 MB)", 2 * src.numel() * 4, lambda: dst.copy_(src))
        add("fill 256 MB (write only)", src.numel() * 4, lambda: dst.zero_())
        add("sum 256 MB (read only)", src.numel() * 4, lambda: src.sum())
    return rows


if __name__ == "__main__":
    main()
