"""GPU suite: the HIP kernels behind the C ABI (include/pdfops.h) against the CPU oracle on identical seeded inputs.
kNN / FPS indices: bit-exact.  Gathers/subtractions: bit-exact.  FMA-reordered sums and atomic scatters: 1e-5/1e-6."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from pointcloudpdf_amd import _native

    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    be = _native.hip_backend()  # raises if libpdfops.so is missing: no fallback
    assert be.lib.pdf_abi_version() >= 1
    return be


def cloud(n, seed, snap=0):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)).astype(np.float32) * np.array([8, 6, 3], dtype=np.float32)
    if snap:
        xyz = (np.floor(xyz * snap) / snap).astype(np.float32)
    return torch.from_numpy(xyz)


def offs(sizes):
    return torch.tensor(np.cumsum(sizes), dtype=torch.int32)


def both_knn(hip, oracle_backend, k, xyz, new_xyz, off, noff):
    i_o, d_o = oracle_backend.knn_query(k, xyz, new_xyz, off, noff)
    i_h, d_h = hip.knn_query(k, xyz.to(DEV), new_xyz.to(DEV), off.to(DEV), noff.to(DEV))
    torch.cuda.synchronize()
    return i_o, d_o, i_h.cpu(), d_h.cpu()


@pytest.mark.parametrize("k", [1, 3, 8, 16, 33, 128])
@pytest.mark.parametrize("snap", [0, 5])
def test_knn_bit_exact(hip, oracle_backend, k, snap):
    sizes = [1500, 37, 2900, 600]
    xyz = cloud(sum(sizes), 11 + k, snap)
    off = offs(sizes)
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, k, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h), f"kNN idx mismatch rows: {(i_o != i_h).any(1).sum().item()}"
    assert torch.equal(d_o, d_h)


def test_knn_cross_queries_and_placeholders(hip, oracle_backend):
    sizes, qsizes = [900, 5, 2000], [200, 5, 333]
    xyz = cloud(sum(sizes), 3)
    sel = torch.cat([torch.arange(0, 200), torch.arange(900, 905), 905 + torch.arange(0, 333) * 5])
    new_xyz = xyz[sel].contiguous()
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, 16, xyz, new_xyz, offs(sizes), offs(qsizes))
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)
    assert (i_h[200:205, 5:] == -1).all() and (d_h[200:205, 5:] == 1e10).all()


def test_knn_many_tiny_scenes_in_one_wave(hip, oracle_backend):
    sizes = [7, 1, 12, 3, 30, 2, 9, 64, 5, 11] * 6
    xyz = cloud(sum(sizes), 21, snap=3)
    off = offs(sizes)
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, 8, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)


def test_knn_long_and_short_redo_lists(hip, oracle_backend):
    """Grid path with ties: a snapped cloud of 11,000 points sends every query to the exact re-scan (list longer than the
    wave-per-query limit -> lane-per-query kernel); 3 duplicated points in a real-valued cloud give a short list (wave-per-
    query kernel).  Both must reproduce the reference's tie order bit for bit."""
    sizes = [11000]
    xyz = cloud(sum(sizes), 5, snap=2)
    off = offs(sizes)
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, 16, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)
    xyz = cloud(6000, 9)
    xyz[100], xyz[2500], xyz[5999] = xyz[7], xyz[7], xyz[4000]   # duplicates -> equal distances for a handful of queries
    off = offs([6000])
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, 8, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)


def test_knn_rejects_bad_arguments(hip):
    from pointcloudpdf_amd._native import PdfOpsError

    xyz = torch.rand(10, 3, device=DEV)
    off = torch.tensor([10], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):
        hip.knn_query(129, xyz, xyz, off, off)
    with pytest.raises(PdfOpsError):
        hip.knn_query(3, xyz.cpu(), xyz.cpu(), off.cpu(), off.cpu())  # CPU tensors: loud failure, no fallback


def test_knn_full_size_properties(hip, oracle_backend):
    """BASELINE config-2 size (2 x 100k, k=8): structure everywhere + oracle equality on a query sample."""
    from pointcloudpdf_amd import synthetic

    batch = synthetic.make_batch([100000, 100000], first_scene_id=0)
    xyz, off = batch["coord"], batch["offset"]
    idx, d2 = hip.knn_query(8, xyz.to(DEV), xyz.to(DEV), off.to(DEV), off.to(DEV))
    idx, d2 = idx.cpu(), d2.cpu()
    assert (d2[:, 1:] >= d2[:, :-1]).all()  # ascending
    assert (idx[:100000] < 100000).all() and (idx[100000:] >= 100000).all() and (idx >= 0).all()
    assert (idx[:, 0] == torch.arange(200000)).float().mean() > 0.999  # self is the nearest (d=0) unless duplicates
    assert (d2[:, 0] == 0).all()
    q = torch.cat([torch.arange(0, 100000, 997), torch.arange(100000, 200000, 991)])
    noff = torch.tensor([(q < 100000).sum().item(), q.numel()], dtype=torch.int32)
    i_o, d_o = oracle_backend.knn_query(8, xyz, xyz[q].contiguous(), off, noff)
    assert torch.equal(i_o, idx[q]) and torch.equal(d_o, d2[q])


# ----------------------------------------------------------------------------------------------- FPS
def both_fps(hip, oracle_backend, xyz, sizes, msizes):
    off, noff = offs(sizes), offs(msizes)
    n_max, m_total = max(sizes), sum(msizes)
    f_o = oracle_backend.farthest_point_sampling(xyz, off, noff, n_max, m_total)
    res = {}
    for mode in ("plain", "bucketed"):
        hip.fps_mode = mode
        res[mode] = hip.farthest_point_sampling(xyz.to(DEV), off.to(DEV), noff.to(DEV), n_max, m_total).cpu()
    hip.fps_mode = "bucketed"
    return f_o, res


def test_knn_counted_launch_returns_the_same_tables_and_the_evaluated_pairs():
    """pdf_knn_query_ws_counted (measurement aid behind the `valu_frac_evaluated` figure of the bench line): identical idx / dist2 and a
    count of evaluated candidate distances that is positive and far below the brute-force m * n pairs the reference kernel evaluates."""
    from pointcloudpdf_amd import _native, synthetic

    be = _native.hip_backend()
    b = synthetic.make_batch([30000, 21000], first_scene_id=11, device="cuda")
    for k in (3, 8, 16):
        idx, d2 = be.knn_query(k, b["coord"], b["coord"], b["offset"], b["offset"])
        idx2, d22, pairs = be.knn_query_counted(k, b["coord"], b["coord"], b["offset"], b["offset"])
        assert torch.equal(idx, idx2) and torch.equal(d2, d22)
        brute = 30000 * 30000 + 21000 * 21000
        assert (k + 1) * 51000 <= pairs < brute / 20, (k, pairs, brute)


@pytest.mark.parametrize("sizes", [[1000], [4096 + 7], [37], [300, 1500, 64], [2048, 1600], [25000], [6250, 6100]])
@pytest.mark.parametrize("snap", [0, 6])
def test_fps_bit_exact(hip, oracle_backend, sizes, snap):
    xyz = cloud(sum(sizes), 5 + len(sizes), snap)
    msizes = [max(s // 4, 1) for s in sizes]
    f_o, res = both_fps(hip, oracle_backend, xyz, sizes, msizes)
    for mode, f_h in res.items():
        bad = (f_o != f_h).nonzero()
        assert torch.equal(f_o, f_h), f"{mode}: first mismatch at sample {bad[0].item() if len(bad) else -1} of {f_o.numel()}"


def test_fps_single_sample_and_duplicates(hip, oracle_backend):
    xyz = torch.cat([cloud(50, 1), cloud(50, 1)])  # every point duplicated: zero distances, ties everywhere
    f_o, res = both_fps(hip, oracle_backend, xyz, [100], [100])
    for f_h in res.values():
        assert torch.equal(f_o, f_h)
    f_o, res = both_fps(hip, oracle_backend, cloud(10, 2), [10], [1])
    for f_h in res.values():
        assert torch.equal(f_o, f_h) and f_h.tolist() == [0]


def test_fps_full_size_100k(hip, oracle_backend):
    """Level-1 FPS of the headline config (100k -> 25k) against the oracle's lock-step emulation."""
    from pointcloudpdf_amd import synthetic

    xyz = torch.from_numpy(synthetic.make_scene(100000, 7)["coord"])
    f_o, res = both_fps(hip, oracle_backend, xyz, [100000], [25000])
    for mode, f_h in res.items():
        assert torch.equal(f_o, f_h), mode
    # size-independent properties: unique samples, first sample is point 0, min pairwise distance decreases
    f = res["bucketed"].long()
    assert f[0] == 0 and f.unique().numel() == f.numel()


def test_fps_config4_size_and_kernel_variants(hip):
    """ScanNet-shaped 150k -> 37.5k (16-wave multi-sample kernel, LDS-resident records near the 160 KB limit) and a 3-scene batch whose
    scenes fall into the three kernel classes (>= 16k points: 16 waves, >= 3k: 8 waves, below: the one-sample kernel -- chosen by
    the LARGEST scene) against the plain reference-shaped kernel; every PDFOPS_FPS_MW / PDFOPS_FPS_K variant gives the same indices."""
    from pointcloudpdf_amd import synthetic

    xyz = torch.from_numpy(synthetic.make_scene(150000, 11, kind="scannet")["coord"]).to(DEV)
    off, noff = offs([150000]).to(DEV), offs([37500]).to(DEV)
    hip.fps_mode = "plain"
    want = hip.farthest_point_sampling(xyz, off, noff, 150000, 37500)
    hip.fps_mode = "bucketed"
    try:
        assert torch.equal(hip.farthest_point_sampling(xyz, off, noff, 150000, 37500), want)
        sizes, msizes = [20000, 5000, 900], [5000, 1250, 225]
        pts = torch.cat([torch.from_numpy(synthetic.make_scene(n, 20 + i)["coord"]) for i, n in enumerate(sizes)]).to(DEV)
        o3, n3 = offs(sizes).to(DEV), offs(msizes).to(DEV)
        hip.fps_mode = "plain"
        want3 = hip.farthest_point_sampling(pts, o3, n3, max(sizes), sum(msizes))
        hip.fps_mode = "bucketed"
        for env in ({}, {"PDFOPS_FPS_MW": "8"}, {"PDFOPS_FPS_MW": "16"}, {"PDFOPS_FPS_MW": "0"}, {"PDFOPS_FPS_K": "1"}, {"PDFOPS_FPS_MW": "0", "PDFOPS_FPS_K": "4"}):
            os.environ.update(env)
            try:
                got = hip.farthest_point_sampling(pts, o3, n3, max(sizes), sum(msizes))
            finally:
                for k_ in env:
                    os.environ.pop(k_, None)
            assert torch.equal(got, want3), env
        # 170k points: the one-centre-per-wave records no longer fit the LDS -> k_fps_multi; 200k: bucket records do not fit -> plain kernel
        for n in (170000, 200000):
            big = torch.from_numpy(synthetic.make_scene(n, 12, kind="scannet")["coord"]).to(DEV)
            ob, nb = offs([n]).to(DEV), offs([n // 4]).to(DEV)
            hip.fps_mode = "plain"
            wb = hip.farthest_point_sampling(big, ob, nb, n, n // 4)
            hip.fps_mode = "bucketed"
            assert torch.equal(hip.farthest_point_sampling(big, ob, nb, n, n // 4), wb), n
    finally:
        hip.fps_mode = "bucketed"


# ----------------------------------------------------------------------------------------------- ball queries
def both_ball(hip, oracle_backend, ns, rmax, rmin, xyz, new_xyz, off, noff, order=None):
    i_o, d_o = oracle_backend.ball_query(ns, rmax, rmin, xyz, new_xyz, off, noff, order=order)
    i_h, d_h = hip.ball_query(ns, rmax, rmin, xyz.to(DEV), new_xyz.to(DEV), off.to(DEV), noff.to(DEV),
                              order=None if order is None else order.to(DEV))
    torch.cuda.synchronize()
    return i_o, d_o, i_h.cpu(), d_h.cpu()


@pytest.mark.parametrize("ns,rmax,rmin", [(16, 0.4, 0.0), (8, 1.2, 0.5), (64, 0.3, 0.0), (5, 2.5, 0.0), (100, 0.8, 0.1)])
@pytest.mark.parametrize("snap", [0, 4])
def test_ball_query_bit_exact(hip, oracle_backend, ns, rmax, rmin, snap):
    """ball_query_cuda_kernel.cu:58-123: shell test, index-ordered candidates, the un-heapified heap_sort permutation,
    padding / strided pick -- bit for bit, incl. a scene shorter than nsample, snapped clouds (duplicates: d2 == 0) and
    rmax = 2.5 on 2900 points (more than 2048 candidates: the list stops at the first 2048 on both sides)."""
    sizes = [1500, 3, 2900, 600]
    xyz = cloud(sum(sizes), 31 + ns, snap)
    off = offs(sizes)
    i_o, d_o, i_h, d_h = both_ball(hip, oracle_backend, ns, rmax, rmin, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h), f"ball idx mismatch rows: {(i_o != i_h).any(1).sum().item()}"
    assert torch.equal(d_o, d_h)
    qsel = torch.cat([torch.arange(0, 1500, 7), torch.arange(1500, 1503), torch.arange(1503, 4403, 11), torch.arange(4403, 5003, 3)])
    qoff = offs([215, 3, 264, 200])
    i_o, d_o, i_h, d_h = both_ball(hip, oracle_backend, ns, rmax, rmin, xyz, xyz[qsel].contiguous(), off, qoff)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)


@pytest.mark.parametrize("ns,rmax,rmin", [(16, 0.4, 0.0), (8, 1.2, 0.5), (70, 0.6, 0.0)])
def test_random_ball_query_bit_exact(hip, oracle_backend, ns, rmax, rmin):
    sizes = [1500, 3, 2900, 600]
    xyz = cloud(sum(sizes), 77, snap=0)
    off = offs(sizes)
    g = torch.Generator().manual_seed(ns)
    order, start = [], 0
    for n in sizes:
        order.append(torch.randperm(n, generator=g, dtype=torch.int32) + start); start += n
    order = torch.cat(order)
    i_o, d_o, i_h, d_h = both_ball(hip, oracle_backend, ns, rmax, rmin, xyz, xyz, off, off, order=order)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)


def test_ball_query_python_goldens_on_gpu(hip, golden_dir):
    """Fixtures produced by the reference's own BallQuery / ball_query_and_group wrappers, replayed through the HIP path."""
    from pointcloudpdf_amd import pointops

    g = np.load(os.path.join(golden_dir, "ops_ball_ref.npz"))
    T = lambda a: torch.from_numpy(np.array(a)).to(DEV)
    for tag, (ns, rmax, rmin) in {"a": (16, 0.5, 0.0), "b": (8, 0.9, 0.3), "c": (32, 0.25, 0.0)}.items():
        i, d = pointops.ball_query(ns, rmax, rmin, T(g["xyz"]), T(g["offset"]), T(g["new_xyz"]), T(g["new_offset"]))
        assert np.array_equal(i.cpu().numpy(), g[f"bq_{tag}_idx"])
        assert_close(d.cpu(), g[f"bq_{tag}_dist"], 1e-6, "ball dist")   # device sqrt vs host sqrt: an ulp
        i, _ = pointops.random_ball_query(ns, rmax, rmin, T(g["xyz"]), T(g["offset"]), T(g["new_xyz"]), T(g["new_offset"]))
        ref = g[f"bq_{tag}_idx"]   # a different permutation picks different members of the same shell
        assert ((i.cpu().numpy() >= 0).sum(1) == (ref >= 0).sum(1)).all()
    out, idx = pointops.ball_query_and_group(T(g["feat"]), T(g["xyz"]), T(g["offset"]), T(g["new_xyz"]), T(g["new_offset"]),
                                             max_radio=0.5, min_radio=0.0, nsample=16, with_xyz=True)
    assert np.array_equal(idx.cpu().numpy(), g["bqg_idx"]) and np.array_equal(out.cpu().numpy(), g["bqg_out"])


def test_ball_query_rejects_bad_arguments(hip):
    xyz = cloud(100, 1).to(DEV)
    off = offs([100]).to(DEV)
    with pytest.raises(ValueError):
        hip.ball_query(8, 0.5, 0.5, xyz, xyz, off, off)
    with pytest.raises(ValueError):
        hip.ball_query(4096, 0.5, 0.0, xyz, xyz, off, off)


# ----------------------------------------------------------------------------------------------- gather family
@pytest.fixture(scope="module")
def table(hip, oracle_backend):
    sizes = [700, 1300]
    xyz = cloud(sum(sizes), 9)
    off = offs(sizes)
    idx, _ = oracle_backend.knn_query(16, xyz, xyz, off, off)
    idx_pad = idx.clone()
    idx_pad[::7, -3:] = -1
    return xyz, off, idx, idx_pad


@pytest.mark.parametrize("c", [32, 6, 3, 1, 64])
def test_grouping2_fwd_bwd(hip, oracle_backend, table, c):
    xyz, off, idx, _ = table
    g = torch.Generator().manual_seed(c)
    x = torch.randn(xyz.shape[0], c, generator=g)
    y_o = oracle_backend.grouping_forward(x, idx)
    y_h = hip.grouping_forward(x.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(y_o, y_h)
    go = torch.randn(y_o.shape, generator=g)
    g_o = oracle_backend.grouping_backward(go, idx, x.shape[0])
    g_h = hip.grouping_backward(go.to(DEV), idx.to(DEV), x.shape[0]).cpu()
    assert_close(g_h, g_o, 1e-5, "grouping bwd")


@pytest.mark.parametrize("c,with_xyz", [(32, True), (32, False), (6, True), (5, False)])
def test_group_fused_twin(hip, oracle_backend, table, c, with_xyz):
    """pdf_group_forward/backward == the reference's python grouping() (composition, run on the CPU through torch)."""
    from pointcloudpdf_amd import pointops, _native

    xyz, off, _, idx_pad = table
    g = torch.Generator().manual_seed(c)
    feat = torch.randn(xyz.shape[0], c, generator=g)
    new_xyz = xyz + 0.01
    prev = _native._set_backend_for_testing(oracle_backend)
    try:
        f_c = feat.clone().requires_grad_(True)
        y_c = pointops.grouping(idx_pad, f_c, xyz, new_xyz, with_xyz=with_xyz)
        go = torch.randn(y_c.shape, generator=g)
        y_c.backward(go)
    finally:
        _native._set_backend_for_testing(prev)
    f_d = feat.to(DEV).requires_grad_(True)
    y_d = pointops.grouping(idx_pad.to(DEV), f_d, xyz.to(DEV), new_xyz.to(DEV), with_xyz=with_xyz)
    assert torch.equal(y_d.detach().cpu(), y_c.detach())
    y_d.backward(go.to(DEV))
    assert_close(f_d.grad.cpu(), f_c.grad, 1e-5, "group bwd")


@pytest.mark.parametrize("c,k,ordered", [(32, 8, False), (32, 8, True), (64, 16, True), (8, 4, False), (32, 12, True), (16, 64, False)])
def test_group_with_xyz_rows_staged_through_lds(hip, oracle_backend, c, k, ordered):
    """grouping(with_xyz) for c % 4 == 0, nsample % 4 == 0 assembles the rows of a few query points in LDS and writes them as whole
    aligned lines (gather_ops.hip, group_fwd_lds): bit-identical to the oracle's rows, with and without a visiting order on the
    table, -1 placeholders included, a last block with fewer points than the others."""
    from pointcloudpdf_amd import _native

    sizes = [611, 1302]
    n = sum(sizes)
    xyz = cloud(n, 21 + c)
    off = offs(sizes)
    idx, _ = oracle_backend.knn_query(k, xyz, xyz, off, off)
    idx[::5, -2:] = -1
    g = torch.Generator().manual_seed(c + k)
    feat = torch.randn(n, c, generator=g)
    new_xyz = xyz + 0.02
    from pointcloudpdf_amd import pointops

    prev = _native._set_backend_for_testing(oracle_backend)
    try:
        want = pointops.grouping(idx, feat, xyz, new_xyz, with_xyz=True)   # the reference's composition, on the CPU
    finally:
        _native._set_backend_for_testing(prev)
    idx_d = idx.to(DEV)
    if ordered:
        _native.attach_order(idx_d, torch.randperm(n, generator=g).to(torch.int32).to(DEV))
    got = hip.group_forward(feat.to(DEV), xyz.to(DEV), new_xyz.to(DEV), idx_d, True).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("c,k", [(32, 3), (64, 3), (7, 5), (512, 3)])
def test_interpolation2_fwd_bwd(hip, oracle_backend, c, k):
    xyz = cloud(3000, 4)
    off = offs([1000, 2000])
    coarse = xyz[::4].contiguous()
    coff = offs([250, 500])
    idx, d2 = oracle_backend.knn_query(k, coarse, xyz, coff, off)
    w = 1.0 / (torch.sqrt(d2) + 1e-8)
    w = w / w.sum(1, keepdim=True)
    w_h = hip.interpolation_weights(d2.to(DEV)).cpu()
    assert_close(w_h, w, 1e-6, "interp weights")
    g = torch.Generator().manual_seed(c)
    x = torch.randn(coarse.shape[0], c, generator=g)
    y_o = oracle_backend.interpolation_forward(x, idx, w)
    y_h = hip.interpolation_forward(x.to(DEV), idx.to(DEV), w.to(DEV)).cpu()
    assert_close(y_h, y_o, 1e-6, "interp fwd")
    go = torch.randn(y_o.shape, generator=g)
    g_o = oracle_backend.interpolation_backward(go, idx, w, x.shape[0])
    g_h = hip.interpolation_backward(go.to(DEV), idx.to(DEV), w.to(DEV), x.shape[0]).cpu()
    assert_close(g_h, g_o, 1e-5, "interp bwd")


@pytest.mark.parametrize("c", [32, 10])
def test_subtraction_fwd_bwd(hip, oracle_backend, table, c):
    xyz, off, idx, _ = table
    g = torch.Generator().manual_seed(c)
    a, b = torch.randn(xyz.shape[0], c, generator=g), torch.randn(xyz.shape[0], c, generator=g)
    y_o = oracle_backend.subtraction_forward(a, b, idx)
    y_h = hip.subtraction_forward(a.to(DEV), b.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(y_o, y_h)
    go = torch.randn(y_o.shape, generator=g)
    g1_o, g2_o = oracle_backend.subtraction_backward(idx, go)
    g1_h, g2_h = hip.subtraction_backward(idx.to(DEV), go.to(DEV))
    assert_close(g1_h.cpu(), g1_o, 1e-5, "sub g1")
    assert_close(g2_h.cpu(), g2_o, 1e-5, "sub g2")
    # the one-walk form (self table, c % 4 == 0: own-row sums + inverse-segment sums together) visits the points in the table's order
    # when one is attached: same sums, bit for bit; and it equals the two-pass form within rounding
    from pointcloudpdf_amd import _native

    idx_o = idx.to(DEV)
    _native.attach_order(idx_o, torch.randperm(idx.shape[0], generator=g).to(torch.int32).to(DEV), torch.randperm(idx.shape[0], generator=g).to(torch.int32).to(DEV))
    g1_p, g2_p = hip.subtraction_backward(idx_o, go.to(DEV))
    assert torch.equal(g1_p, g1_h) and torch.equal(g2_p, g2_h)
    hip.fuse_own_rows = False
    try:
        g1_t, g2_t = hip.subtraction_backward(idx.to(DEV), go.to(DEV))
    finally:
        hip.fuse_own_rows = True
    assert_close(g1_t.cpu(), g1_h.cpu(), 1e-6, "sub g1 two-pass")
    assert_close(g2_t.cpu(), g2_h.cpu(), 1e-6, "sub g2 two-pass")


@pytest.mark.parametrize("c,w_c", [(32, 4), (64, 8), (24, 24), (128, 16), (16, 2), (40, 5)])
def test_aggregation_fwd_bwd(hip, oracle_backend, table, c, w_c):
    xyz, off, idx, _ = table
    n, ns = idx.shape
    g = torch.Generator().manual_seed(c)
    x, pos, w = torch.randn(n, c, generator=g), torch.randn(n, ns, c, generator=g), torch.randn(n, ns, w_c, generator=g)
    y_o = oracle_backend.aggregation_forward(x, pos, w, idx)
    y_h = hip.aggregation_forward(x.to(DEV), pos.to(DEV), w.to(DEV), idx.to(DEV)).cpu()
    assert_close(y_h, y_o, 1e-6, "agg fwd")
    go = torch.randn(n, c, generator=g)
    o = oracle_backend.aggregation_backward(x, pos, w, idx, go)
    h = hip.aggregation_backward(x.to(DEV), pos.to(DEV), w.to(DEV), idx.to(DEV), go.to(DEV))
    for a, b, nm in zip(h, o, ["gi", "gp", "gw"]):
        assert_close(a.cpu(), b, 1e-5, "agg " + nm)


def test_attention_steps_fwd_bwd(hip, oracle_backend):
    g = torch.Generator().manual_seed(1)
    n, E, G, C = 600, 5000, 4, 8
    q, k, v = (torch.randn(n, G, C, generator=g) for _ in range(3))
    aw = torch.randn(C, generator=g)
    it = torch.randint(0, n, (E,), generator=g, dtype=torch.int32)
    ir = torch.randint(0, n, (E,), generator=g, dtype=torch.int32)
    d = lambda t: t.to(DEV)
    y_o = oracle_backend.attention_relation_step_forward(q, k, aw, it, ir)
    y_h = hip.attention_relation_step_forward(d(q), d(k), d(aw), d(it), d(ir)).cpu()
    assert_close(y_h, y_o, 1e-6, "rel fwd")
    go = torch.randn(E, G, generator=g)
    o = oracle_backend.attention_relation_step_backward(q, k, aw, it, ir, go)
    h = hip.attention_relation_step_backward(d(q), d(k), d(aw), d(it), d(ir), d(go))
    for a, b, nm in zip(h, o, ["gq", "gk", "gw"]):
        assert_close(a.cpu(), b, 2e-5, "rel " + nm)
    ew = torch.randn(E, G, generator=g)
    y_o = oracle_backend.attention_fusion_step_forward(ew, v, it, ir)
    y_h = hip.attention_fusion_step_forward(d(ew), d(v), d(it), d(ir)).cpu()
    assert_close(y_h, y_o, 1e-5, "fus fwd")
    go = torch.randn(n, G, C, generator=g)
    o = oracle_backend.attention_fusion_step_backward(ew, v, it, ir, go)
    h = hip.attention_fusion_step_backward(d(ew), d(v), d(it), d(ir), d(go))
    for a, b, nm in zip(h, o, ["gw", "gv"]):
        assert_close(a.cpu(), b, 1e-5, "fus " + nm)


def test_python_level_goldens_on_gpu(hip, golden_dir):
    """The fixtures produced by the reference's Python wrappers, replayed through the HIP path."""
    from pointcloudpdf_amd import pointops

    g = np.load(os.path.join(golden_dir, "ops_python_ref.npz"))
    T = lambda a, grad=False: torch.from_numpy(np.array(a)).to(DEV).requires_grad_(grad)
    idx, dist = pointops.knn_query(8, T(g["xyz"]), T(g["offset"]), T(g["new_xyz"]), T(g["new_offset"]))
    assert np.array_equal(idx.cpu().numpy(), g["knn_idx"])
    # dist = sqrt(dist2): dist2 is bit-exact (test_knn_bit_exact); the device sqrt may differ from the host's by an ulp
    assert_close(dist.cpu(), g["knn_dist"], 1e-6, "knn dist")
    feat = T(g["feat"], True)
    y = pointops.grouping(T(g["idx_pad"]), feat, T(g["xyz"]), T(g["new_xyz"]), with_xyz=True)
    assert np.array_equal(y.detach().cpu().numpy(), g["grouping_xyz"])
    y.backward(T(g["grouping_go"]))
    assert_close(feat.grad.cpu(), g["grouping_gfeat"], 1e-5)
    cf = T(g["interp_feat"], True)
    yi = pointops.interpolation(T(g["new_xyz"]), T(g["xyz"]), cf, T(g["new_offset"]), T(g["offset"]))
    assert_close(yi.cpu(), g["interp_out"], 1e-6)
    yi.backward(T(g["interp_go"]))
    assert_close(cf.grad.cpu(), g["interp_gfeat"], 1e-5)
    inp, pos, w = T(g["agg_in"], True), T(g["agg_pos"], True), T(g["agg_w"], True)
    ya = pointops.aggregation(inp, pos, w, T(g["self_idx"]))
    assert_close(ya.cpu(), g["agg_out"], 1e-6)
    ya.backward(T(g["agg_go"]))
    assert_close(inp.grad.cpu(), g["agg_gin"], 1e-5)
    assert_close(w.grad.cpu(), g["agg_gw"], 1e-5)
    a, b = T(g["sub_a"], True), T(g["sub_b"], True)
    ys = pointops.subtraction(a, b, T(g["self_idx"]))
    assert np.array_equal(ys.detach().cpu().numpy(), g["sub_out"])


def test_empty_inputs_are_no_ops(hip):
    """Zero queries / zero rows: every entry point returns empty outputs instead of launching an empty grid (0-size tensors carry
    null data pointers)."""
    xyz = cloud(300, 2).to(DEV)
    off = offs([300]).to(DEV)
    q0 = torch.zeros(0, 3, device=DEV)
    qoff = torch.zeros(1, dtype=torch.int32, device=DEV)
    idx, d2 = hip.knn_query(8, xyz, q0, off, qoff)
    assert idx.shape == (0, 8) and d2.shape == (0, 8)
    bi, bd = hip.ball_query(8, 0.5, 0.0, xyz, q0, off, qoff)
    assert bi.shape == (0, 8)
    feat = torch.randn(300, 16, device=DEV)
    e_idx = torch.zeros(0, 8, dtype=torch.int32, device=DEV)
    assert hip.grouping_forward(feat, e_idx).shape == (0, 8, 16)
    assert hip.group_forward(feat, xyz, q0, e_idx, True).shape == (0, 8, 19)
    assert torch.count_nonzero(hip.grouping_backward(torch.zeros(0, 8, 16, device=DEV), e_idx, 300)) == 0
    e3 = torch.zeros(0, 3, dtype=torch.int32, device=DEV)
    assert hip.interpolation_forward(feat, e3, torch.zeros(0, 3, device=DEV)).shape == (0, 16)
    assert hip.subtraction_forward(torch.zeros(0, 16, device=DEV), feat, e_idx).shape == (0, 8, 16)
    assert hip.aggregation_forward(feat, torch.zeros(0, 8, 16, device=DEV), torch.zeros(0, 8, 2, device=DEV), e_idx).shape == (0, 16)
    torch.cuda.synchronize()


@pytest.mark.parametrize("radius,ns,snap", [(0.1, 64, 0), (0.25, 16, 0), (0.3, 64, 4), (1.5, 32, 0), (0.05, 8, 0)])
def test_radius_neighbors_grid_matches_in_order_scan(hip, radius, ns, snap):
    """pdf_radius_neighbors_self (27 grid cells, rank selection; radius 1.5 overflows the candidate list -> in-wave scan) against
    pdf_random_ball_query walked along the identity permutation: indices and squared distances bit for bit, incl. duplicated points
    (snapped cloud), a 3-point scene and empty balls beyond the point itself."""
    sizes = [6000, 3, 9000, 800]
    xyz = cloud(sum(sizes), 51, snap).to(DEV)
    off = offs(sizes).to(DEV)
    order = torch.arange(sum(sizes), dtype=torch.int32, device=DEV)
    i_s, d_s = hip.ball_query(ns, radius, 0.0, xyz, xyz, off, off, order=order)
    i_g, d_g = hip.radius_neighbors_self(ns, radius, xyz, off)
    torch.cuda.synchronize()
    assert torch.equal(i_s, i_g), f"rows differing: {(i_s != i_g).any(1).sum().item()}"
    assert torch.equal(d_s, d_g)


def test_more_than_64_scenes_take_the_scan_paths(hip, oracle_backend):
    """The grid workspaces are sized for <= 64 scenes per call; beyond that kNN and the radius table fall back to the exact scans."""
    sizes = [40 + (i % 7) * 13 for i in range(70)]
    xyz = cloud(sum(sizes), 61)
    off = offs(sizes)
    i_o, d_o, i_h, d_h = both_knn(hip, oracle_backend, 8, xyz, xyz, off, off)
    assert torch.equal(i_o, i_h) and torch.equal(d_o, d_h)
    order = torch.arange(sum(sizes), dtype=torch.int32)
    r_o, _ = oracle_backend.ball_query(16, 0.8, 0.0, xyz, xyz, off, off, order=order)
    r_h, _ = hip.radius_neighbors_self(16, 0.8, xyz.to(DEV), off.to(DEV))
    assert torch.equal(r_o, r_h.cpu())


# ---------------------------------------------------------------- scatter-adds as segmented gathers (csrc/seg_gather.hip)
def test_inverse_table_matches_numpy():
    """inverse_table(idx, n): every destination's segment lists exactly the entries that gather it, in ascending entry order;
    entries with idx < 0 belong to no segment."""
    from pointcloudpdf_amd import _native

    g = torch.Generator().manual_seed(3)
    n, m, k = 700, 900, 8
    idx = torch.randint(-1, n, (m, k), generator=g, dtype=torch.int32)
    idx[::11] = -1
    off, ent, base = _native.inverse_table(idx.cuda(), n)
    off, ent = off.cpu().numpy(), ent.cpu().numpy()
    flat = idx.numpy().reshape(-1)
    assert base == 0 and off.shape == (n + 1,) and off[-1] == flat.size and off[0] == (flat < 0).sum()
    for v in range(0, n, 37):
        assert np.array_equal(ent[off[v]:off[v + 1]], np.nonzero(flat == v)[0]), v


@pytest.mark.parametrize("c,k,wc", [(32, 8, 4), (64, 16, 8), (6, 5, 3), (256, 16, 32)])
def test_segmented_gather_backward_matches_atomic_kernels(oracle_backend, c, k, wc):
    """grouping2 / interpolation2 / subtraction / aggregation backward through the inverse-table gathers: equal to the reference-shaped
    atomic kernels (HIP) and to the oracle (summation order differs: 1e-6), bit-reproducible from run to run, -1 rows skipped."""
    from pointcloudpdf_amd import _native

    be = _native.hip_backend()
    g = torch.Generator().manual_seed(c + k)
    n, m = 3000, 2600
    idx = torch.randint(0, n, (m, k), generator=g, dtype=torch.int32)
    idx_pad = idx.clone()
    idx_pad[::9, -2:] = -1     # placeholder rows: HIP paths only (the reference kernel, hence the oracle, dereferences idx = -1)
    go3 = torch.randn(m, k, c, generator=g)
    go2 = torch.randn(m, c, generator=g)
    w3 = torch.rand(m, 3, generator=g)
    idx3 = torch.randint(0, n, (m, 3), generator=g, dtype=torch.int32)
    inp, pos, wt = torch.randn(n, c, generator=g), torch.randn(n, k, c, generator=g), torch.randn(n, k, wc, generator=g)
    idxs = torch.randint(0, n, (n, k), generator=g, dtype=torch.int32)
    gon = torch.randn(n, c, generator=g)
    D = lambda t: t.cuda()

    def run_all():
        out = {"grouping": be.grouping_backward(D(go3), D(idx), n), "grouping_pad": be.grouping_backward(D(go3), D(idx_pad), n),
               "interp": be.interpolation_backward(D(go2), D(idx3), D(w3), n)}
        out["sub1"], out["sub2"] = be.subtraction_backward(D(idxs), D(pos), n)
        out["agg_in"], out["agg_pos"], out["agg_w"] = be.aggregation_backward(D(inp), D(pos), D(wt), D(idxs), D(gon))
        return {k_: v.cpu() for k_, v in out.items()}

    assert be.use_inverse
    a, b = run_all(), run_all()
    for key in a:
        if key != "agg_w":   # (grad_weight at c > 64 still combines 64-channel spans with atomics; not one of the scatters moved here)
            assert torch.equal(a[key], b[key]), f"{key}: not bit-reproducible"
    be.use_inverse = False
    try:
        atomic = run_all()
    finally:
        be.use_inverse = True
    ref = {"grouping": oracle_backend.grouping_backward(go3, idx, n), "interp": oracle_backend.interpolation_backward(go2, idx3, w3, n)}
    ref["sub1"], ref["sub2"] = oracle_backend.subtraction_backward(idxs, pos, n)
    ref["agg_in"], ref["agg_pos"], ref["agg_w"] = oracle_backend.aggregation_backward(inp, pos, wt, idxs, gon)
    for key in a:
        if key in ref:
            assert_close(a[key], ref[key], 2e-6, f"{key} vs oracle")
        assert_close(a[key], atomic[key], 2e-6, f"{key} vs atomic kernels")


def test_forward_gathers_with_a_visiting_order_are_bit_identical():
    """grouping2 / grouping(with_xyz) / subtraction / aggregation / interpolation forward with the Morton visiting order a Geometry
    attaches to its kNN tables (XCD-chunked *_ord kernels) against the same ops on an untagged copy of the table (storage order): the
    outputs are bit-identical -- the order only changes WHEN a query is processed."""
    from pointcloudpdf_amd import _native, synthetic
    from pointcloudpdf_amd.geometry import Geometry

    be = _native.hip_backend()
    b = synthetic.make_batch([30000, 17000], first_scene_id=12, device="cuda")
    geom = Geometry(b["coord"], b["offset"], b["offset_host"])
    lvl2, _ = geom.down(0, 4)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for (k, src, qry, c) in ((8, 0, 0, 32), (16, lvl2, lvl2, 64), (16, 0, lvl2, 32), (3, lvl2, 0, 64)):
        idx, d2 = geom.knn(k, src, qry)
        order = _native.order_of(idx)
        assert order is not None and sorted(order.tolist()) == list(range(idx.shape[0]))
        plain = idx.clone()                                        # no order attached
        assert _native.order_of(plain) is None
        ns, nq = geom.coord(src).shape[0], idx.shape[0]
        feat = torch.randn(ns, c, device="cuda", generator=g)
        assert torch.equal(be.grouping_forward(feat, idx), be.grouping_forward(feat, plain))
        assert torch.equal(be.group_forward(feat, geom.coord(src), geom.coord(qry), idx, True), be.group_forward(feat, geom.coord(src), geom.coord(qry), plain, True))
        if k == 3:
            w = be.interpolation_weights(d2)
            assert torch.equal(be.interpolation_forward(feat, idx, w), be.interpolation_forward(feat, plain, w))
        if src == qry:
            f1 = torch.randn(nq, c, device="cuda", generator=g)
            assert torch.equal(be.subtraction_forward(f1, feat, idx), be.subtraction_forward(f1, feat, plain))
            pos = torch.randn(nq, k, c, device="cuda", generator=g); wt = torch.randn(nq, k, c // 8, device="cuda", generator=g)
            assert torch.equal(be.aggregation_forward(feat, pos, wt, idx), be.aggregation_forward(feat, pos, wt, plain))
