"""CPU restatement (numpy) of the reference's GridSample key / partition logic -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows pointcept/datasets/transform.py:813-823 (grid coordinates, key, argsort, unique) and :911-925 (fnv_hash_vec) for ONE scene.
Pinned against the reference itself: tests/golden/ops_gridsample_ref.npz is produced by running the reference's own GridSample
(tests/golden/make_golden.py::run_gridsample_cases)."""
import numpy as np


def fnv_hash_vec(arr):
    """transform.py:911-925 (multiply, then xor, per axis; uint64 wrap-around)"""
    arr = arr.astype(np.uint64, copy=True)
    hashed = np.uint64(14695981039346656037) * np.ones(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1]):
        hashed *= np.uint64(1099511628211)
        hashed = np.bitwise_xor(hashed, arr[:, j])
    return hashed


def grid_partition(coord, grid_size, float32_division=False):
    """-> key (n) uint64, grid_coord (n,3) int64 (minus the scene minimum), inverse (n), count (V), idx_sort (n)   (transform.py:813-823, 838-840)"""
    g = np.array(grid_size, dtype=np.float32 if float32_division else np.float64)
    scaled = coord / g
    grid_coord = np.floor(scaled).astype(int)
    grid_coord -= grid_coord.min(0)
    key = fnv_hash_vec(grid_coord)
    idx_sort = np.argsort(key, kind="stable")   # upstream: default (unstable) kind -- the order inside a voxel is unspecified there
    key_sort = key[idx_sort]
    _, inverse_sorted, count = np.unique(key_sort, return_inverse=True, return_counts=True)
    inverse = np.zeros_like(inverse_sorted)
    inverse[idx_sort] = inverse_sorted
    return key, grid_coord, inverse, count, idx_sort
