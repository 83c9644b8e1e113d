"""GPU suite: fused BatchNorm(+residual)(+ReLU) kernels (csrc/pointwise.hip) and the split-K Linear against torch."""
import numpy as np
import pytest
import torch

from helpers import l2_rel, max_rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c", [(5000, 32), (3001, 128), (777, 512), (40000, 64), (4096, 256), (13, 128)])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("train", [True, False])
def test_bn_act_matches_torch(n, c, res, relu, train):
    from pointcloudpdf_amd.dense import bn_act

    g = torch.Generator(device="cuda").manual_seed(n + c)
    x0 = torch.randn(n, c, device="cuda", generator=g) * 2 + 0.5
    r0 = torch.randn(n, c, device="cuda", generator=g) if res else None
    go = torch.randn(n, c, device="cuda", generator=g)
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(c).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.1, c)); bn.running_var.copy_(torch.linspace(0.8, 1.2, c))
        bn.train(train)
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        if fused:
            y = bn_act(bn, x, r, relu)
        else:
            y = bn(x)
            if res:
                y = y + r
            if relu:
                y = torch.relu(y)
        y.backward(go)
        outs.append(dict(y=y.detach(), gx=x.grad, gr=r.grad if res else None, gw=bn.weight.grad, gb=bn.bias.grad,
                         rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nb=bn.num_batches_tracked.clone()))
    a, b = outs
    for k in a:
        if a[k] is None:
            continue
        assert max_rel(a[k].float().cpu().numpy(), b[k].float().cpu().numpy()) < 2e-5, k


def test_split_k_linear_matches_torch():
    from pointcloudpdf_amd.dense import linear

    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(50001, 32, device="cuda", generator=g)
    lin = torch.nn.Linear(32, 64).cuda()
    go = torch.randn(50001, 64, device="cuda", generator=g)
    res = []
    for f in (lambda x: linear(lin, x), lambda x: lin(x)):
        lin.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = f(x)
        y.backward(go)
        res.append((y.detach(), x.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    for a, b in zip(*res):
        assert max_rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("n,k,o", [(5000, 32, 32), (3001, 6, 32), (777, 512, 512), (40000, 64, 192), (999, 35, 64), (2500, 32, 13), (130, 256, 1),
                                   (100003, 32, 32), (50001, 64, 64), (12517, 128, 128), (3125, 256, 256), (5, 32, 96), (1000, 128, 48),
                                   (780, 1024, 512), (333, 1024, 64)])   # (k = 1024: two column windows of 512 through the streaming kernel when no statistics are asked for)
@pytest.mark.parametrize("pre", [False, True])
def test_rowlin_forward_dgrad_wgrad(n, k, o, pre):
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + k + o)
    x = torch.randn(n, k, device="cuda", generator=g)
    w = torch.randn(o, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(o, device="cuda", generator=g)
    coef = None
    fx = x
    if pre:
        sc = torch.rand(k, device="cuda", generator=g) + 0.5
        sh = torch.randn(k, device="cuda", generator=g) * 0.3
        coef = torch.cat([sc, sh, torch.zeros(2 * k, device="cuda")])
        fx = torch.relu(x * sc + sh)
    y, partial = be.rowlin(x, w, b, coef=coef, relu=True, stats=True)
    ref = fx.double() @ w.double().t() + b.double()
    assert max_rel(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    rows = partial._pdf_rows
    ps = partial[: rows * 2 * o].view(rows, 2 * o).double().sum(0)
    assert max_rel(ps[:o].cpu().numpy(), ref.sum(0).cpu().numpy()) < 1e-4
    assert max_rel(ps[o:].cpu().numpy(), (ref * ref).sum(0).cpu().numpy()) < 1e-5
    go = torch.randn(n, o, device="cuda", generator=g)
    gx, _ = be.rowlin(go, w, transpose_w=True)
    assert max_rel(gx.cpu().numpy(), (go.double() @ w.double()).cpu().numpy()) < 2e-6
    gx2, _ = be.rowlin(go, w, transpose_w=True, out=gx.clone(), accumulate=True)
    assert max_rel(gx2.cpu().numpy(), (2 * (go.double() @ w.double())).cpu().numpy()) < 2e-6
    dw, db = be.rowlin_wgrad(go, x, coef, True, True)
    assert max_rel(dw.cpu().numpy(), (go.double().t() @ fx.double()).cpu().numpy()) < 1e-5
    assert max_rel(db.cpu().numpy(), go.double().sum(0).cpu().numpy()) < 1e-5
    # strided views (row stride > width)
    big = torch.randn(n, k + 5, device="cuda", generator=g)
    y2, _ = be.rowlin(big[:, :k], w, b)
    assert max_rel(y2.cpu().numpy(), (big[:, :k].double() @ w.double().t() + b.double()).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("n,k,o", [(780, 1024, 512), (333, 1024, 64), (3, 1024, 32), (50, 2048, 32)])
@pytest.mark.parametrize("pre", [False, True])
def test_rowlin_forward_wide_inputs(n, k, o, pre):
    """k = 1024 (the TransitionUp head's Linear(2 * 512, 512)): both 512-wide column windows in one launch of the streaming kernel (one
    accumulation chain, bias at the end), on top of an existing y when asked; with a prologue / other widths: the tiled kernel."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + k + o)
    x = torch.randn(n, k, device="cuda", generator=g)
    w = torch.randn(o, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(o, device="cuda", generator=g)
    coef, fx = None, x
    if pre:
        sc = torch.rand(k, device="cuda", generator=g) + 0.5
        sh = torch.randn(k, device="cuda", generator=g) * 0.3
        coef = torch.cat([sc, sh, torch.zeros(2 * k, device="cuda")])
        fx = torch.relu(x * sc + sh)
    ref = fx.double() @ w.double().t() + b.double()
    y, _ = be.rowlin(x, w, b, coef=coef, relu=True)
    assert max_rel(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    y2, _ = be.rowlin(x, w, b, coef=coef, relu=True, out=y.clone(), accumulate=True)
    assert max_rel(y2.cpu().numpy(), (2 * ref).cpu().numpy()) < 2e-6
    y3, _ = be.rowlin(x, w, None, coef=coef, relu=True)
    assert max_rel(y3.cpu().numpy(), (ref - b.double()).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,k,o", [(5000, 32, 32), (777, 512, 512), (40000, 64, 192), (12517, 128, 128), (3125, 256, 256), (5, 32, 96)])
@pytest.mark.parametrize("pre", [False, True])
def test_rowlin_reduced_precision_operands(n, k, o, pre, dtype):
    """mma_input = 1 | 2 (per-call argument, include/pdfops.h): the streaming Linear products round their OPERANDS to fp16 / bfloat16 in registers (after the folded
    BatchNorm + ReLU prologue, which stays fp32) and accumulate in fp32 -- so the result must equal, to fp32 accumulation error, the
    float64 product of the operands rounded the same way; tensors in memory stay fp32 and mode 0 comes back afterwards."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    mode = _native.MMA_INPUT_OF_DTYPE[dtype]
    rnd = lambda t: t.to(dtype).double()
    g = torch.Generator(device="cuda").manual_seed(n + k + o + mode)
    x = torch.randn(n, k, device="cuda", generator=g)
    w = torch.randn(o, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(o, device="cuda", generator=g)
    coef, fx = None, x
    if pre:
        sc = torch.rand(k, device="cuda", generator=g) + 0.5
        sh = torch.randn(k, device="cuda", generator=g) * 0.3
        coef = torch.cat([sc, sh, torch.zeros(2 * k, device="cuda")])
        fx = torch.relu(x * sc + sh)
    go = torch.randn(n, o, device="cuda", generator=g)
    y32, _ = be.rowlin(x, w, b, coef=coef, relu=True)
    with _native.mma_input(mode):
        assert _native.current_mma_input() == mode
        y, partial = be.rowlin(x, w, b, coef=coef, relu=True, stats=True)
        gx, _ = be.rowlin(go, w, transpose_w=True)
        dw, db = be.rowlin_wgrad(go, x, coef, True, True)
    assert _native.current_mma_input() == 0
    assert y.dtype == torch.float32 and gx.dtype == torch.float32 and dw.dtype == torch.float32
    ref = rnd(fx) @ rnd(w).t() + b.double()
    # (with the prologue the kernel forms x * scale + shift as one FMA, torch as a product and a sum: an operand one fp32 ulp apart now and
    #  then rounds to the neighbouring half-precision value -- a few 2^-11 / 2^-8 flips per row instead of none)
    assert max_rel(y.cpu().numpy(), ref.cpu().numpy()) < (1e-3 if pre else 5e-6)
    assert not torch.equal(y, y32), "the reduced-precision kernel was not the one that ran"
    exact = fx.double() @ w.double().t() + b.double()
    assert max_rel(y.cpu().numpy(), exact.cpu().numpy()) < (4e-3 if dtype == torch.float16 else 3e-2)   # the price of the rounding
    rows = partial._pdf_rows
    ps = partial[: rows * 2 * o].view(rows, 2 * o).double().sum(0)
    assert max_rel(ps[:o].cpu().numpy(), y.double().sum(0).cpu().numpy()) < 1e-4       # statistics of the values actually written
    if o in (32, 64, 128, 256, 512):   # the reduction width of the input gradient: streaming kernel (other widths: tiled kernel, fp32 operands)
        assert max_rel(gx.cpu().numpy(), (rnd(go) @ rnd(w)).cpu().numpy()) < 5e-6
    else:
        assert max_rel(gx.cpu().numpy(), (go.double() @ w.double()).cpu().numpy()) < 2e-6
    assert max_rel(dw.cpu().numpy(), (rnd(go).t() @ rnd(fx)).cpu().numpy()) < (1e-3 if pre else 2e-5)
    assert max_rel(db.cpu().numpy(), go.double().sum(0).cpu().numpy()) < 1e-5   # the bias gradient is a plain fp32 column sum


@pytest.mark.parametrize("n,c", [(30011, 32), (9000, 64), (4097, 128), (3124, 256), (780, 512), (5, 64)])
@pytest.mark.parametrize("mode", [0, 1])
def test_rowlin_wgrad_group(n, c, mode):
    """The five c x c weight gradients of a Bottleneck backward in one launch + one reduction: each with its own gradient rows, its own
    input rows and its own folded BatchNorm + ReLU prologue (or none), biases where asked; against float64 and against the one-by-one
    entry points; deterministic (two calls give identical bits).  mode 1: fp16 operands (the grouped kernel honours the call's mma_input)."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + c)
    gs = [torch.randn(n, c, device="cuda", generator=g) for _ in range(5)]
    xs = [torch.randn(n, c, device="cuda", generator=g) for _ in range(3)]
    coefs = []
    for _ in range(2):
        sc = torch.rand(c, device="cuda", generator=g) + 0.5
        sh = torch.randn(c, device="cuda", generator=g) * 0.3
        coefs.append(torch.cat([sc, sh, torch.zeros(2 * c, device="cuda")]))
    x_of, coef_of = [xs[0], xs[1], xs[1], xs[1], xs[2]], [coefs[0], coefs[1], coefs[1], coefs[1], None]
    relus, need_bias = [True, True, True, True, False], [False, True, True, True, False]
    with _native.mma_input(mode):
        out = be.rowlin_wgrad_group(gs, x_of, coef_of, relus, need_bias)
        again = be.rowlin_wgrad_group(gs, x_of, coef_of, relus, need_bias)
        single = [be.rowlin_wgrad(gs[i], x_of[i], coef_of[i], relus[i], need_bias[i]) for i in range(5)]
    assert out is not None
    dws, dbs = out
    rnd = (lambda t: t.half().double()) if mode == 1 else (lambda t: t.double())
    for i in range(5):
        fx = x_of[i] if coef_of[i] is None else torch.relu(x_of[i] * coef_of[i][:c] + coef_of[i][c:2 * c])
        ref = rnd(gs[i]).t() @ rnd(fx)
        tol = 1e-5 if mode == 0 else 1e-3   # (mode 1 with the prologue: FMA vs product + sum flips a few operand roundings, as in the single form)
        assert max_rel(dws[i].cpu().numpy(), ref.cpu().numpy()) < tol, i
        assert max_rel(dws[i].cpu().numpy(), single[i][0].cpu().numpy()) < (2e-6 if mode == 0 else 1e-3), i
        assert torch.equal(dws[i], again[0][i])
        if need_bias[i]:
            assert max_rel(dbs[i].cpu().numpy(), gs[i].double().sum(0).cpu().numpy()) < 1e-5 and torch.equal(dbs[i], again[1][i])
        else:
            assert dbs[i] is None


def test_rowlin_wgrad_group_outside_the_streaming_shapes():
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = [torch.randn(100, 48, device="cuda") for _ in range(2)]
    x = [torch.randn(100, 48, device="cuda") for _ in range(2)]
    assert be.rowlin_wgrad_group(g, x, [None, None], [False, False], [False, False]) is None   # PDF_ERR_UNSUPPORTED: one product at a time


def test_mma_input_rejects_unknown_modes():
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    x = torch.randn(64, 32, device="cuda")
    w = torch.randn(32, 32, device="cuda")
    _native._MMA.mode = 3   # (bypassing the context manager's own check)
    try:
        with pytest.raises(_native.PdfOpsError, match="status -1"):
            be.rowlin(x, w)
    finally:
        _native._MMA.mode = 0


def test_two_threads_two_modes_run_concurrently():
    """The product-input mode is a per-call argument (ABI 4; round 3 kept it in a process-wide word that dispatch read): two threads on
    their own streams run forward + backward products in DIFFERENT modes at the same time, repeatedly, and every result is bit-identical
    to the same thread's result when it runs alone."""
    import threading

    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(7)
    n, c = 30000, 64
    x = torch.randn(n, c, device="cuda", generator=g)
    w = torch.randn(c, c, device="cuda", generator=g) / 8
    go = torch.randn(n, c, device="cuda", generator=g)

    def work(mode, reps):
        outs = []
        with _native.mma_input(mode):
            for _ in range(reps):
                y, _ = be.rowlin(x, w)
                gx, _ = be.rowlin(go, w, transpose_w=True)
                dw, _ = be.rowlin_wgrad(go, x, None, False, False)
                outs.append((y, gx, dw))
        return outs

    alone = {m: work(m, 1)[0] for m in (0, 1, 2)}
    torch.cuda.synchronize()
    assert not torch.equal(alone[0][0], alone[1][0]) and not torch.equal(alone[1][0], alone[2][0])
    results, errors = {}, []
    start = threading.Barrier(3)

    def thread(mode):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                start.wait()
                results[mode] = work(mode, 40)
                torch.cuda.current_stream().synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=thread, args=(m,)) for m in (0, 1, 2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for m in (0, 1, 2):
        for y, gx, dw in results[m]:
            assert torch.equal(y, alone[m][0]) and torch.equal(gx, alone[m][1]) and torch.equal(dw, alone[m][2]), m
    assert _native.current_mma_input() == 0


@pytest.mark.parametrize("n,c", [(30011, 32), (9000, 64), (4097, 128), (1500, 256), (300, 512), (200, 48)])
@pytest.mark.parametrize("pre", [False, True])
def test_rowlin_multi(n, c, pre):
    """q/k/v projections in one launch, their joint input gradient and the three weight gradients."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + c)
    x = torch.randn(n, c, device="cuda", generator=g)
    ws = [torch.randn(c, c, device="cuda", generator=g) / c ** 0.5 for _ in range(3)]
    bs = [torch.randn(c, device="cuda", generator=g) for _ in range(3)]
    coef, fx = None, x
    if pre:
        sc = torch.rand(c, device="cuda", generator=g) + 0.5
        sh = torch.randn(c, device="cuda", generator=g) * 0.3
        coef = torch.cat([sc, sh, torch.zeros(2 * c, device="cuda")])
        fx = torch.relu(x * sc + sh)
    ys = be.rowlin_multi([x], ws, bs, coef=coef, relu=True, nout=3)
    for y, w, b in zip(ys, ws, bs):
        assert max_rel(y.cpu().numpy(), (fx.double() @ w.double().t() + b.double()).cpu().numpy()) < 2e-6
    gs = [torch.randn(n, c, device="cuda", generator=g) for _ in range(3)]
    (gx,) = be.rowlin_multi(gs, ws, None, transpose_w=True, nout=1)
    ref = sum(gi.double() @ w.double() for gi, w in zip(gs, ws))
    assert max_rel(gx.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    dws, dbs = be.rowlin_wgrad_multi(gs, x, coef, True)
    for gi, dw, db in zip(gs, dws, dbs):
        assert max_rel(dw.cpu().numpy(), (gi.double().t() @ fx.double()).cpu().numpy()) < 1e-5
        assert max_rel(db.cpu().numpy(), gi.double().sum(0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("n,c,nin", [(30011, 32, 3), (30011, 32, 1), (9000, 64, 3), (9001, 64, 1), (4097, 128, 3), (4097, 128, 1),
                                     (1500, 256, 3), (1501, 256, 1), (300, 512, 1), (7, 64, 3)])
@pytest.mark.parametrize("train", [True, False])
def test_dgrad_with_batchnorm_backward_sums(n, c, nin, train):
    """Input gradient of Linear layers that read relu(bn(z)) with bn's backward sums as the product's epilogue
    (pdf_rowlin_dgrad_bstats + pdf_bn_act_backward_presummed) against torch autograd in fp64."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + c + nin)
    z = torch.randn(n, c, device="cuda", generator=g) * 1.7 + 0.4
    gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    gamma[::5] *= -1.0   # negative scales: the ReLU mask must follow the sign of the pre-activation, not of xhat
    beta = torch.randn(c, device="cuda", generator=g) * 0.3
    ws = [torch.randn(c, c, device="cuda", generator=g) / c ** 0.5 for _ in range(nin)]
    gs = [torch.randn(n, c, device="cuda", generator=g) for _ in range(nin)]
    zd = z.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    if train:
        mean, var = zd.mean(0), zd.var(0, unbiased=False)
    else:
        mean, var = torch.full((c,), 0.3, device="cuda", dtype=torch.float64), torch.full((c,), 2.0, device="cuda", dtype=torch.float64)
    rstd = (var + 1e-5).rsqrt()
    y = torch.relu((zd - mean) * rstd * gd + bd)
    sum((y @ w.double().t() * gi.double()).sum() for w, gi in zip(ws, gs)).backward()
    scale = (gamma.double() * rstd.detach()).float()
    coef = torch.cat([scale, (beta.double() - mean.detach() * gamma.double() * rstd.detach()).float(), mean.detach().float(), rstd.detach().float()])
    out = be.rowlin_dgrad_bn_backward(gs, ws, z, coef, training=train, relu=True)
    assert out is not None, "streaming kernels cover every Bottleneck width"
    dz, dgamma, dbeta = out
    assert l2_rel(dz.cpu().numpy(), zd.grad.cpu().numpy()) < 2e-6
    assert l2_rel(dgamma.cpu().numpy(), gd.grad.cpu().numpy()) < 1e-5
    assert l2_rel(dbeta.cpu().numpy(), bd.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("C,K,train", [(32, 8, True), (64, 16, True), (256, 16, True), (512, 16, True), (32, 8, False)])
def test_bottleneck_matrix_core_path(C, K, train):
    """Bottleneck through rowlin / folded norms vs the op-by-op path (same module, switch off)."""
    from pointcloudpdf_amd import synthetic
    from pointcloudpdf_amd.geometry import Geometry
    from pointcloudpdf_amd.point_transformer import Bottleneck

    sizes = [900, 700] if C <= 64 else [300, 260]
    batch = synthetic.make_batch(sizes, first_scene_id=60, grid_size=0.25, device="cuda")
    res = []
    default = Bottleneck.matrix_core
    for mc in (True, False):
        torch.manual_seed(0)
        geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
        blk = Bottleneck(C, C, 8, K).cuda()
        synthetic.fill_parameters_deterministic(blk, seed=7)
        blk.train(train)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(sum(sizes), C, device="cuda", generator=g).requires_grad_(True)
        Bottleneck.matrix_core = mc
        try:
            y = blk([geom.coord(0), x, geom.offset(0)])[1]
            y.backward(torch.randn(y.shape, device="cuda", generator=g))
        finally:
            Bottleneck.matrix_core = default
        out = {"y": y.detach().cpu().numpy(), "gx": x.grad.cpu().numpy()}
        out.update({"g_" + n: p.grad.cpu().numpy() for n, p in blk.named_parameters() if p.grad is not None})
        out.update({"b_" + n: b.detach().float().cpu().numpy() for n, b in blk.named_buffers()})
        res.append(out)
    a, b = res
    gscale = max(abs(v).max() for k, v in b.items() if k.startswith("g_"))
    for k in b:
        if k.startswith("g_") and abs(b[k]).max() < 1e-4 * gscale:
            continue  # analytically-zero gradients (biases in front of a train-mode BatchNorm)
        if k == "y" or k.startswith("b_"):
            assert max_rel(a[k], b[k]) < 2e-5, (k, max_rel(a[k], b[k]))
        else:  # gradients: Frobenius norm (isolated ReLU-kink flips, see helpers.l2_rel)
            # 3-element tensors (BNp) get no averaging: one ReLU-kink flip among 560 x 16 rows moves them by ~5e-3 in EITHER path
            tol = 1e-2 if b[k].size <= 3 else 5e-3
            assert l2_rel(a[k], b[k]) < tol, (k, l2_rel(a[k], b[k]), max_rel(a[k], b[k]))


@pytest.mark.parametrize("n,c", [(200003, 13), (5000, 14), (77, 64), (3, 2)])
def test_fused_cross_entropy(n, c):
    """csrc/loss.hip vs nn.CrossEntropyLoss (mean, ignore_index=-1), value and gradient."""
    from pointcloudpdf_amd.segmentor import CrossEntropyLoss

    g = torch.Generator(device="cuda").manual_seed(n + c)
    logits = (torch.randn(n, c, device="cuda", generator=g) * 3).requires_grad_(True)
    target = torch.randint(-1, c, (n,), device="cuda", generator=g)
    crit = CrossEntropyLoss(loss_weight=0.7, ignore_index=-1)
    loss = crit(logits, target)
    (loss * 1.3).backward()
    ref_in = logits.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, target, ignore_index=-1) * 0.7
    (ref * 1.3).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert max_rel(logits.grad.cpu().numpy(), ref_in.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("cin,sizes,train", [(32, [3000, 2600], True), (64, [900, 40], True), (128, [700], True), (256, [300, 260], True),
                                              (32, [2000, 1500], False), (64, [10, 1200], True)])
def test_fused_transition_down(cin, sizes, train):
    """csrc/transition_down.hip (Gram-matrix BatchNorm statistics, sparse arg-max backward) vs the composed path
    (grouping kernel + Linear + BatchNorm + ReLU + MaxPool1d), incl. scenes shorter than nsample (-1 placeholder rows)."""
    from pointcloudpdf_amd import synthetic
    from pointcloudpdf_amd.geometry import Geometry
    from pointcloudpdf_amd.point_transformer import TransitionDown

    batch = synthetic.make_batch(sizes, first_scene_id=70, grid_size=0.25, device="cuda")
    res = []
    for fused in (True, False):
        geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
        td = TransitionDown(cin, 2 * cin, 4, 16).cuda()
        synthetic.fill_parameters_deterministic(td, seed=11)
        td.train(train)
        g = torch.Generator(device="cuda").manual_seed(2)
        x = torch.randn(sum(sizes), cin, device="cuda", generator=g).requires_grad_(train)
        TransitionDown.fused = fused
        try:
            with torch.set_grad_enabled(train):
                p2, y, o2 = td([geom.coord(0), x, geom.offset(0)])
            out = {"y": y.detach().cpu().numpy(), "p": p2.cpu().numpy(), "o": o2.cpu().numpy()}
            if train:
                y.backward(torch.randn(y.shape, device="cuda", generator=g))
                out["gx"] = x.grad.cpu().numpy()
                out.update({"g_" + n: p.grad.cpu().numpy() for n, p in td.named_parameters()})
            out.update({"b_" + n: b.detach().float().cpu().numpy() for n, b in td.named_buffers()})
        finally:
            TransitionDown.fused = True
        res.append(out)
    a, b = res
    assert np.array_equal(a["p"], b["p"]) and np.array_equal(a["o"], b["o"])
    for k in b:
        if k in ("p", "o"):
            continue
        if k == "y" or k.startswith("b_"):
            assert max_rel(a[k], b[k]) < 2e-5, (k, max_rel(a[k], b[k]))
        else:
            assert l2_rel(a[k], b[k]) < 2e-3, (k, l2_rel(a[k], b[k]), max_rel(a[k], b[k]))


@pytest.mark.parametrize("n,k,o,train", [(5000, 64, 32, True), (3000, 512, 256, True), (777, 32, 32, True), (20000, 32, 32, False), (1500, 256, 512, True)])
def test_linear_bn_relu_node(n, k, o, train):
    """Linear -> BatchNorm1d -> ReLU as one node (csrc/block.hip: pdf_linbn_*) vs the composed ops, through point_transformer._seq."""
    import torch.nn as nn
    from pointcloudpdf_amd import dense, synthetic
    from pointcloudpdf_amd.point_transformer import _seq

    res = []
    for fused in (True, False):
        seq = nn.Sequential(nn.Linear(k, o), nn.BatchNorm1d(o), nn.ReLU(inplace=True)).cuda()
        synthetic.fill_parameters_deterministic(seq, seed=4)
        seq.train(train)
        g = torch.Generator(device="cuda").manual_seed(n + k)
        x = torch.randn(n, k, device="cuda", generator=g).requires_grad_(train)
        dense.LINBN = fused
        try:
            with torch.set_grad_enabled(train):
                y = _seq(seq, x)
            out = {"y": y.detach().cpu().numpy()}
            if train:
                y.backward(torch.randn(y.shape, device="cuda", generator=g))
                out["gx"] = x.grad.cpu().numpy()
                out.update({"g_" + nm: p.grad.cpu().numpy() for nm, p in seq.named_parameters()})
            out.update({"b_" + nm: b.detach().float().cpu().numpy() for nm, b in seq.named_buffers()})
        finally:
            dense.LINBN = True
        res.append(out)
    a, b = res
    gscale = max([abs(v).max() for kk, v in b.items() if kk.startswith("g_")] + [1e-30])
    for kk in b:
        if kk.startswith("g_") and abs(b[kk]).max() < 1e-4 * gscale:
            continue  # the Linear bias in front of a train-mode BatchNorm: analytically zero gradient
        if kk == "y" or kk.startswith("b_"):
            assert max_rel(a[kk], b[kk]) < 2e-5, (kk, max_rel(a[kk], b[kk]))
        else:
            assert l2_rel(a[kk], b[kk]) < 2e-3, (kk, l2_rel(a[kk], b[kk]))


def test_fused_sgd_matches_torch_sgd():
    """engine.FusedSGD (one launch over all tensors, csrc/optim.hip) against torch.optim.SGD over five steps: odd lengths, a
    tensor longer than several chunks, a misaligned view, a parameter that gets no gradient on some steps."""
    from pointcloudpdf_amd import engine

    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(3,), (13, 32), (70001,), (4096,), (512, 513), (9, 3, 3)]
    base = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    flat = torch.randn(1001, device="cuda", generator=g)
    pa = [torch.nn.Parameter(b.clone()) for b in base] + [torch.nn.Parameter(flat.clone()[1:])]   # (4-byte aligned only)
    pb = [torch.nn.Parameter(b.clone()) for b in base] + [torch.nn.Parameter(flat.clone()[1:])]
    oa = engine.FusedSGD(pa, lr=0.05, momentum=0.9, weight_decay=1e-2)
    ob = torch.optim.SGD(pb, lr=0.05, momentum=0.9, weight_decay=1e-2)
    for it in range(5):
        oa.zero_grad(); ob.zero_grad()
        for k, (x, y) in enumerate(zip(pa, pb)):
            if k == 1 and it in (1, 3):
                continue   # no gradient this step: skipped by both
            gr = torch.randn(x.shape, device="cuda", generator=g)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert (x - y).abs().max() <= 1e-6 * (1 + y.abs().max()), (tuple(x.shape), float((x - y).abs().max()))
    # a torch optimizer: LR schedulers attach (the reference steps its scheduler every iteration) and the state moves both ways
    assert isinstance(oa, torch.optim.Optimizer) and oa.param_groups[0]["lr"] == 0.05
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, 1, 0.5), torch.optim.lr_scheduler.StepLR(ob, 1, 0.5)
    pc = [torch.nn.Parameter(y.detach().clone()) for y in pb]
    oc = engine.FusedSGD(pc, lr=123.0, momentum=0.0, weight_decay=0.0)
    import copy
    # torch.optim.SGD's checkpoint: lr / momentum / weight decay and the momentum buffers (deep copy = what torch.save / torch.load
    # hand over; load_state_dict itself aliases same-device tensors)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    sc = torch.optim.lr_scheduler.StepLR(oc, 1, 0.5)
    for it in range(3):
        for x, y, z in zip(pa, pb, pc):
            gr = torch.randn(x.shape, device="cuda", generator=g)
            x.grad, y.grad, z.grad = gr.clone(), gr.clone(), gr.clone()
        oa.step(); ob.step(); oc.step()
        sa.step(); sb.step(); sc.step()
    assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"] == oc.param_groups[0]["lr"] == 0.05 / 8
    for x, y, z in zip(pa, pb, pc):
        assert (x - y).abs().max() <= 1e-6 * (1 + y.abs().max()), (tuple(x.shape), float((x - y).abs().max()))
        assert (z - y).abs().max() <= 1e-6 * (1 + y.abs().max()), (tuple(x.shape), float((z - y).abs().max()))
    ob2 = torch.optim.SGD(pb, lr=1.0, momentum=0.9)
    ob2.load_state_dict(copy.deepcopy(oa.state_dict()))         # and back: FusedSGD's state in torch.optim.SGD
    assert torch.equal(ob2.state[pb[2]]["momentum_buffer"], oa.state[pa[2]]["momentum_buffer"])
    # moved parameters (model.to(), load_state_dict(assign=True)): pointers are read at every step
    with torch.no_grad():
        pa[2].data = pa[2].data.clone()
    before = pa[2].detach().clone()
    pa[2].grad = torch.ones_like(pa[2])
    for k in range(len(pa)):
        if k != 2:
            pa[k].grad = None
    oa.step()
    torch.cuda.synchronize()
    assert not torch.equal(pa[2].detach(), before)


def test_geometry_branch_batchnorm_from_coordinate_sums():
    """pdf_knn_rel_moments against a torch fp64 evaluation, and a Bottleneck whose geometry-branch BatchNorm comes from those sums
    (Geometry.rel_moments, attached to the idx tensor) against the same block running its own statistics pass (P1): outputs, input /
    parameter gradients and the running statistics agree to fp32 rounding."""
    from pointcloudpdf_amd import _native, synthetic
    from pointcloudpdf_amd.geometry import Geometry
    from pointcloudpdf_amd.point_transformer import Bottleneck

    be = _native.hip_backend()
    batch = synthetic.make_batch([5000, 3000, 4100], first_scene_id=70, grid_size=0.1, device="cuda")
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"])
    idx, _ = geom.knn(16, 0, 0)
    mom = geom.rel_moments(16, 0)
    p = geom.coord(0).double()
    rel = torch.where((idx >= 0)[..., None], p[idx.clamp(min=0).long()] - p[:, None, :], torch.zeros((), dtype=torch.float64, device="cuda"))
    scene = torch.bucketize(torch.arange(p.shape[0], device="cuda"), batch["offset"].long(), right=True)
    for s in range(3):
        r = rel[scene == s].reshape(-1, 3)
        M = r.t() @ r
        ref = torch.cat([r.sum(0), torch.stack([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])])
        assert (mom[s] - ref).abs().max() <= 1e-9 * ref.abs().max() + 1e-12, (s, mom[s], ref)
    assert _native.moments_of(idx) is not None
    res = []
    for use in (True, False):
        be.use_moments = use
        try:
            torch.manual_seed(0)
            blk = Bottleneck(64, 64, 8, 16).cuda()
            synthetic.fill_parameters_deterministic(blk, seed=9)
            blk.train()
            g = torch.Generator(device="cuda").manual_seed(2)
            x = torch.randn(p.shape[0], 64, device="cuda", generator=g).requires_grad_(True)
            y = blk([geom.coord(0), x, geom.offset(0)])[1]
            y.backward(torch.randn(y.shape, device="cuda", generator=g))
        finally:
            be.use_moments = True
        out = {"y": y.detach(), "gx": x.grad}
        out.update({"g_" + n: q.grad for n, q in blk.named_parameters() if q.grad is not None})
        out.update({"b_" + n: b.detach().float() for n, b in blk.named_buffers()})
        res.append(out)
    a, b = res
    gscale = max(v.abs().max().item() for k, v in b.items() if k.startswith("g_"))
    for k in b:
        scale = b[k].abs().max().item() + 1e-12
        if k.startswith("g_") and scale < 1e-4 * gscale:
            continue   # analytically-zero gradients (biases in front of a train-mode BatchNorm): rounding noise in both runs
        assert (a[k] - b[k]).abs().max().item() <= (2e-5 if (k == "y" or k.startswith("b_")) else 2e-3) * scale, (k, (a[k] - b[k]).abs().max().item(), scale)


@pytest.mark.parametrize("n,c", [(3648, 32), (8192, 32), (40000, 32), (9000, 64), (700, 256)])
def test_block_halves_stay_inside_their_partial_buffer(n, c):
    """The Bottleneck halves' own nodes (dense._BlockPre / _BlockPost: the path of a block whose attention layer is not fused) hand the
    C calls a `partial` scratch sized max(pdf_bn_partial_floats, pdf_rowlin_partial_floats): the dgrad epilogue
    (pdf_rowlin_dgrad_bstats) writes pdf_rowlin_partial_rows rows of 2c floats, more than the BatchNorm passes' rows at c = 32 and
    n < 65536 (round-2 advisor finding: heap overrun).  Run both backward calls with a guard region behind `partial`."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator(device="cuda").manual_seed(n + c)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    e = lambda *s: torch.empty(*s, device="cuda")
    need = max(int(be.lib.pdf_bn_partial_floats(n, c)), int(be.lib.pdf_rowlin_partial_floats(n, c)))
    GUARD, SENT = 1 << 16, 12345.0
    cc = c * c
    # pre half
    x, W1, Wq, Wk, Wv = r(n, c), r(c, c) / c ** 0.5, r(c, c) / c ** 0.5, r(c, c) / c ** 0.5, r(c, c) / c ** 0.5
    g1, b1, bq = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    z1, coef1, xq, xk, xv = e(n, c), e(4 * c), e(n, c), e(n, c), e(n, c)
    part = torch.full((need + GUARD,), SENT, device="cuda")
    be.block_call("pre_forward", n, c, [x, W1, g1, b1, rm, rv, Wq, bq, Wk, bq, Wv, bq, z1, coef1, xq, xk, xv, part], True, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert bool((part[need:] == SENT).all()), "pre_forward wrote behind its partial buffer"
    gx, grads, dy = e(n, c), e(cc + 2 * c + 3 * (cc + c)), e(n, c)
    part.fill_(SENT)
    ws = be.wgrad_workspace(n, c, c, 3, x.device)
    be.block_call("pre_backward", n, c, [x, z1, coef1, W1, Wq, Wk, Wv, r(n, c), r(n, c), r(n, c), gx, grads, dy, part, ws], True)
    torch.cuda.synchronize()
    assert bool((part[need:] == SENT).all()), "pre_backward wrote behind its partial buffer"
    assert bool(torch.isfinite(gx).all()) and bool(torch.isfinite(grads).all())
    # post half
    t, W3 = r(n, c), r(c, c) / c ** 0.5
    rm2, rv2, rm3, rv3 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    coef2, z3, coef3, y = e(4 * c), e(n, c), e(4 * c), e(n, c)
    part.fill_(SENT)
    be.block_call("post_forward", n, c, [t, x, g1, b1, rm2, rv2, W3, g1, b1, rm3, rv3, coef2, z3, coef3, y, part], True, 1e-5, 0.1)
    torch.cuda.synchronize()
    assert bool((part[need:] == SENT).all()), "post_forward wrote behind its partial buffer"
    gt, gres, grads2, da = e(n, c), e(n, c), e(cc + 4 * c), e(n, c)
    part.fill_(SENT)
    be.block_call("post_backward", n, c, [r(n, c), t, x, z3, coef2, coef3, W3, gt, gres, grads2, da, part, ws], True)
    torch.cuda.synchronize()
    assert bool((part[need:] == SENT).all()), "post_backward wrote behind its partial buffer"
    assert bool(torch.isfinite(gt).all()) and bool(torch.isfinite(grads2).all())
