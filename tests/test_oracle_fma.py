"""CPU suite: how much of the "bit-exact kNN / FPS indices" claim depends on FMA contraction.

The reference builds libs/pointops with `nvcc -O2` (libs/pointops/setup.py:29), i.e. with fmad on: its
`d2 = (new_x - x)*(new_x - x) + ...` (knn_query_cuda_kernel.cu:92, sampling_cuda_kernel.cu:54) is contracted to FMAs in an order that
cannot be observed without an NVIDIA toolchain.  The oracle (and the HIP library) therefore carry three arithmetics of that
expression -- as written (default), and the two possible FMA chains -- and this file (a) pins the FMA modes against exact rational
arithmetic, (b) counts how many kNN rows / FPS picks move between the arithmetics on a BASELINE config-2 scene and on a grid-snapped
scene.  The counts are written to DESIGN.md section 3 / INTEGRATION.md by tools/fma_sensitivity.py (same functions)."""
import sys
import os
from fractions import Fraction

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _round_f32(fr):
    """Exact rational -> nearest float32 (candidates around the double approximation compared exactly)."""
    c = np.float32(float(fr))
    cands = [c, np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))]
    return min(cands, key=lambda v: abs(Fraction(float(v)) - fr))


def _fma(a, b, c):
    return _round_f32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def _dist_exact_chain(q, x, mode):
    d = [np.float32(q[i]) - np.float32(x[i]) for i in range(3)]
    if mode == 0:
        return np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    if mode == 1:
        return _fma(d[2], d[2], _fma(d[1], d[1], np.float32(d[0] * d[0])))
    return _fma(d[2], d[2], _fma(d[0], d[0], np.float32(d[1] * d[1])))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_distance_modes_against_exact_rational_arithmetic(oracle_backend, mode):
    """k = 1 kNN of one query over a scene returns that arithmetic's minimum distance; every (query, point) distance is exposed by
    making each point its own one-point scene."""
    rng = np.random.default_rng(5 + mode)
    n = 400
    pts = (rng.random((n, 3)) * np.array([8, 6, 3])).astype(np.float32)
    qs = (pts + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    off = torch.arange(1, n + 1, dtype=torch.int32)
    prev = oracle_backend.set_dist_mode(mode)
    try:
        idx, d2 = oracle_backend.knn_query(1, torch.from_numpy(pts), torch.from_numpy(qs), off, off)
    finally:
        oracle_backend.set_dist_mode(prev)
    want = np.array([_dist_exact_chain(qs[i], pts[i], mode) for i in range(n)], dtype=np.float32)
    assert np.array_equal(d2.numpy()[:, 0], want)
    if mode:   # the modes are not all the same function
        other = np.array([_dist_exact_chain(qs[i], pts[i], 0) for i in range(n)], dtype=np.float32)
        assert (other != want).any()


def test_fma_sensitivity_of_knn_and_fps_indices(oracle_backend):
    """Counts on one 100k-point S3DIS-shaped scene (BASELINE config 2's unit) and on a 2 cm grid-snapped copy.  Bounds are loose
    versions of the measured numbers (tools/fma_sensitivity.py prints them)."""
    import fma_sensitivity as fs

    res = fs.measure(oracle_backend, n_points=100000, n_queries=8000, fps_levels=1)
    real, snapped = res["scene"], res["snapped"]
    for mode in ("fma1", "fma2"):
        # un-snapped data: distances move in most rows, the neighbour SETS and their order in (almost) none
        assert real[mode]["knn_k8_dist_rows_differ"] > 0
        assert real[mode]["knn_k8_idx_rows_differ"] <= 8, real
        assert real[mode]["knn_k16_idx_rows_differ"] <= 16, real
        # FPS is a chain: one flipped arg-max shifts later picks; measured 6-10 of 25,000
        assert real[mode]["fps_picks_differ"] <= 250, real
        # snapped data is tie-heavy: equal as-written distances can separate under FMA (and vice versa), so rows do move
        assert snapped[mode]["knn_k8_idx_rows_differ"] <= snapped["queries"]
    assert res["oracle_mode_after"] == 0
