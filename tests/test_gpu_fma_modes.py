"""GPU suite: the FMA-distance builds of the HIP library (libpdfops_fma1.so / libpdfops_fma2.so, PDFOPS_DIST_FMA=1|2) against the oracle
in the matching arithmetic (oracle_set_dist_mode): kNN (grid path, exact re-scan of ties, scan kernel), FPS (bucketed multi-sample
kernels and the plain kernel) and the ball queries, bit for bit.  A user validating against an `nvcc -O2` build of libs/pointops
(FMA contraction on, libs/pointops/setup.py:29) picks the mode that matches their build; the default stays as written."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", params=[1, 2])
def fma(request, oracle_backend):
    from pointcloudpdf_amd import _native

    assert torch.cuda.is_available()
    be = _native.HipBackend(_native.load_library(_native.library_path(request.param)))   # raises if the variant is not built
    prev = oracle_backend.set_dist_mode(request.param)
    yield be, oracle_backend, request.param
    oracle_backend.set_dist_mode(prev)


def cloud(n, seed, snap=0):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)).astype(np.float32) * np.array([8, 6, 3], dtype=np.float32)
    if snap:
        xyz = (np.floor(xyz * snap) / snap).astype(np.float32)
    return torch.from_numpy(xyz)


def offs(sizes):
    return torch.tensor(np.cumsum(sizes), dtype=torch.int32)


@pytest.mark.parametrize("k", [3, 8, 16, 33])
@pytest.mark.parametrize("snap", [0, 5])
def test_knn_fma_modes_bit_exact(fma, k, snap):
    be, orc, mode = fma
    sizes = [1500, 37, 2900, 600]
    xyz, off = cloud(sum(sizes), 11 + k, snap), offs(sizes)
    i_o, d_o = orc.knn_query(k, xyz, xyz, off, off)
    i_h, d_h = be.knn_query(k, xyz.to(DEV), xyz.to(DEV), off.to(DEV), off.to(DEV))
    assert torch.equal(i_o, i_h.cpu()) and torch.equal(d_o, d_h.cpu())
    if not snap:   # and the mode is not a no-op: some distances differ from the as-written ones
        prev = orc.set_dist_mode(0)
        try:
            _, d_w = orc.knn_query(k, xyz, xyz, off, off)
        finally:
            orc.set_dist_mode(prev)
        assert not torch.equal(d_w, d_o)


def test_knn_fma_modes_full_scene_sample(fma):
    from pointcloudpdf_amd import synthetic

    be, orc, mode = fma
    xyz = torch.from_numpy(synthetic.make_scene(100000, 3)["coord"])
    off = offs([100000])
    idx, d2 = be.knn_query(16, xyz.to(DEV), xyz.to(DEV), off.to(DEV), off.to(DEV))
    q = torch.arange(0, 100000, 499)
    i_o, d_o = orc.knn_query(16, xyz, xyz[q].contiguous(), off, offs([q.numel()]))
    assert torch.equal(i_o, idx.cpu()[q]) and torch.equal(d_o, d2.cpu()[q])


@pytest.mark.parametrize("sizes", [[1000], [4096 + 7], [300, 1500, 64], [25000], [6250, 6100]])
@pytest.mark.parametrize("snap", [0, 6])
def test_fps_fma_modes_bit_exact(fma, sizes, snap):
    be, orc, mode = fma
    xyz = cloud(sum(sizes), 5 + len(sizes), snap)
    msizes = [max(s // 4, 1) for s in sizes]
    off, noff = offs(sizes), offs(msizes)
    f_o = orc.farthest_point_sampling(xyz, off, noff, max(sizes), sum(msizes))
    for kind in ("plain", "bucketed"):
        be.fps_mode = kind
        f_h = be.farthest_point_sampling(xyz.to(DEV), off.to(DEV), noff.to(DEV), max(sizes), sum(msizes)).cpu()
        bad = (f_o != f_h).nonzero()
        assert torch.equal(f_o, f_h), f"mode {mode} {kind}: first mismatch at sample {bad[0].item() if len(bad) else -1} of {f_o.numel()}"
    be.fps_mode = "bucketed"


def test_fps_fma_modes_full_size_100k(fma):
    from pointcloudpdf_amd import synthetic

    be, orc, mode = fma
    xyz = torch.from_numpy(synthetic.make_scene(100000, 7)["coord"])
    off, noff = offs([100000]), offs([25000])
    f_o = orc.farthest_point_sampling(xyz, off, noff, 100000, 25000)
    f_h = be.farthest_point_sampling(xyz.to(DEV), off.to(DEV), noff.to(DEV), 100000, 25000).cpu()
    assert torch.equal(f_o, f_h)


@pytest.mark.parametrize("ns,rmax,rmin", [(16, 0.4, 0.0), (8, 1.2, 0.5)])
def test_ball_query_fma_modes_bit_exact(fma, ns, rmax, rmin):
    be, orc, mode = fma
    sizes = [700, 1300]
    xyz, off = cloud(sum(sizes), 31), offs(sizes)
    i_o, d_o = orc.ball_query(ns, rmax, rmin, xyz, xyz, off, off)
    i_h, d_h = be.ball_query(ns, rmax, rmin, xyz.to(DEV), xyz.to(DEV), off.to(DEV), off.to(DEV))
    assert torch.equal(i_o, i_h.cpu()) and torch.equal(d_o, d_h.cpu())
