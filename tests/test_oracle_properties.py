"""CPU suite: properties of the C oracle itself (the kernel restatement) on edge cases the reference's
kernels define: short scenes (-1 / 1e10 placeholders), uneven scenes, duplicate points / ties, FPS tie rule."""
import numpy as np
import pytest
import torch


def brute_knn(xyz, q, k):
    d = ((q[:, None, :] - xyz[None, :, :]) ** 2).sum(-1)
    return np.sort(d, 1)[:, :k]


def test_knn_sorted_distances_match_bruteforce(oracle_backend):
    rng = np.random.default_rng(0)
    xyz = rng.random((900, 3)).astype(np.float32)
    off = np.array([250, 900], dtype=np.int32)
    idx, d2 = oracle_backend.knn_query(16, torch.from_numpy(xyz), torch.from_numpy(xyz), torch.from_numpy(off), torch.from_numpy(off))
    idx, d2 = idx.numpy(), d2.numpy()
    assert (np.diff(d2, axis=1) >= 0).all()
    for s, e in [(0, 250), (250, 900)]:
        assert ((idx[s:e] >= s) & (idx[s:e] < e)).all()  # never crosses a scene boundary
        x = xyz[s:e].astype(np.float32)
        # as-written fp32 arithmetic
        dx = x[:, None, :] - x[None, :, :]
        ref = np.sort((dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2], 1)[:, :16]
        assert np.array_equal(ref, d2[s:e])


def test_knn_placeholders_for_short_scene(oracle_backend):
    xyz = torch.rand(12, 3)
    off = torch.tensor([5, 12], dtype=torch.int32)
    idx, d2 = oracle_backend.knn_query(8, xyz, xyz, off, off)
    assert (idx[:5, 5:] == -1).all() and (d2[:5, 5:] == 1e10).all()
    assert (idx[:5, :5] >= 0).all() and (idx[5:, :7] >= 5).all() and (idx[5:, 7] == -1).all()


def test_knn_rejects_bad_nsample(oracle_backend):
    xyz = torch.rand(4, 3)
    off = torch.tensor([4], dtype=torch.int32)
    with pytest.raises(ValueError):
        oracle_backend.knn_query(129, xyz, xyz, off, off)


def fps_closed_form(xyz, start, end, m, bs):
    """Independent numpy model of the reference's arg-max: max tmp, ties -> min (bitrev(slot), k)."""
    lg = int(np.log2(bs))
    n = end - start
    rel = np.arange(n)
    slot = rel % bs
    rev = np.array([int(format(s, f"0{lg}b")[::-1], 2) if lg else 0 for s in slot])
    tmp = np.full(n, 1e10, dtype=np.float32)
    out = [start]
    old = 0
    P = xyz[start:end]
    for _ in range(1, m):
        dx = P - P[old]
        d = (dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1]) + dx[:, 2] * dx[:, 2]
        tmp = np.minimum(d.astype(np.float32), tmp)
        best = tmp.max()
        cand = np.nonzero(tmp == best)[0]
        order = np.lexsort((cand, rev[cand]))
        old = int(cand[order[0]])
        out.append(start + old)
    return np.array(out, dtype=np.int32)


@pytest.mark.parametrize("n,snap", [(1000, False), (4096 + 7, False), (300, True), (1500, True), (37, True)])
def test_fps_matches_closed_form_tie_rule(oracle_backend, n, snap):
    rng = np.random.default_rng(n)
    xyz = rng.random((n, 3)).astype(np.float32)
    if snap:  # grid-snapped coordinates: many exactly equal distances
        xyz = (np.floor(xyz * 6) / 6).astype(np.float32)
    m = max(n // 4, 1)
    off = torch.tensor([n], dtype=torch.int32)
    noff = torch.tensor([m], dtype=torch.int32)
    idx = oracle_backend.farthest_point_sampling(torch.from_numpy(xyz), off, noff, n, m).numpy()
    bs = oracle_backend.opt_n_threads(n)
    assert np.array_equal(idx, fps_closed_form(xyz, 0, n, m, bs))


def test_fps_block_size_comes_from_largest_scene(oracle_backend):
    # two scenes: the small one is sampled with the block size of the big one (sampling.py:15-17)
    rng = np.random.default_rng(5)
    xyz = (np.floor(rng.random((1400, 3)) * 5) / 5).astype(np.float32)
    off = torch.tensor([200, 1400], dtype=torch.int32)
    noff = torch.tensor([50, 350], dtype=torch.int32)
    idx = oracle_backend.farthest_point_sampling(torch.from_numpy(xyz), off, noff, 1200, 350).numpy()
    bs = oracle_backend.opt_n_threads(1200)
    assert bs == 1024
    assert np.array_equal(idx[:50], fps_closed_form(xyz, 0, 200, 50, bs))
    assert np.array_equal(idx[50:], fps_closed_form(xyz, 200, 1400, 300, bs))


def test_opt_n_threads(oracle_backend):
    # libs/pointops/src/cuda_utils.h:11-14
    for n, want in [(1, 1), (2, 2), (3, 2), (63, 32), (64, 64), (1000, 512), (1024, 1024), (100000, 1024), (1 << 20, 1024)]:
        assert oracle_backend.opt_n_threads(n) == want
