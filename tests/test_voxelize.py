"""GridSample voxel keys / partition.  CPU: the numpy restatement (oracle/voxel.py) against fixtures produced by
the REFERENCE's own GridSample (tests/golden/ops_gridsample_ref.npz).  GPU: pointcloudpdf_amd.voxelize.grid_sample (pdf_grid_hash +
device sorts) over a batch of scenes against the same fixtures and the oracle -- keys, grid coordinates, voxel partition and counts
bit-exact; the kept point per voxel is unspecified upstream (unstable argsort + np.random) and is checked for membership."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

GRID_CASES = {"a": (31, 60000, 0.05), "b": (32, 25000, 0.02), "c": (33, 7, 0.5)}


def dense_scene(seed, n, room=(6.0, 4.0, 2.5)):
    """Same generator as tests/golden/make_golden.py::dense_scene."""
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3)) * np.array(room) - np.array([1.0, 0.5, 0.25])
    face = rng.integers(0, 3, n)
    c[np.arange(n), face] = np.where(rng.random(n) < 0.5, -np.array([1.0, 0.5, 0.25])[face], (np.array(room) - np.array([1.0, 0.5, 0.25]))[face])
    return c.astype(np.float32)


@pytest.fixture(scope="module")
def gg(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_gridsample_ref.npz"))


@pytest.mark.parametrize("tag", sorted(GRID_CASES))
def test_numpy_oracle_matches_reference_gridsample(gg, tag):
    from oracle import voxel

    seed, n, gs = GRID_CASES[tag]
    coord = dense_scene(seed, n)
    assert str(gg[f"{tag}_division_dtype"]) == "float64"   # NumPy >= 2: float32 / 0-d float64 array promotes
    key, grid, inverse, count, idx_sort = voxel.grid_partition(coord, gs)
    assert np.array_equal(key, gg[f"{tag}_key"])
    assert np.array_equal(inverse, gg[f"{tag}_inverse"])
    kept = gg[f"{tag}_kept_index"]                         # the reference's own pick: one point of every voxel, in key order
    assert np.array_equal(inverse[kept], np.arange(count.shape[0]))
    assert np.array_equal(grid[kept], gg[f"{tag}_kept_grid"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "test"])
def test_gpu_grid_sample_batch_matches_reference(gg, mode):
    from pointcloudpdf_amd import voxelize

    tags = sorted(GRID_CASES)
    scenes = [dense_scene(*GRID_CASES[t][:2]) for t in tags]
    sizes = [s.shape[0] for s in scenes]
    # one grid size per call: batch the two scenes that share none -> run per grid size, each as a 2-scene batch with itself shifted
    for t, sc in zip(tags, scenes):
        gs = GRID_CASES[t][2]
        other = (sc + np.float32(3.7)).astype(np.float32)              # a second scene in the same batch: same shape, other place
        coord = torch.from_numpy(np.concatenate([sc, other])).cuda()
        off = torch.tensor([sc.shape[0], 2 * sc.shape[0]], dtype=torch.int32, device="cuda")
        out = voxelize.grid_sample(coord, off, gs, mode=mode, generator=torch.Generator(device="cuda").manual_seed(3))
        n = sc.shape[0]
        key = out["key"][:n].cpu().numpy().view(np.uint64)
        assert np.array_equal(key, gg[f"{t}_key"])
        inv = out["inverse"][:n].cpu().numpy()
        assert np.array_equal(inv, gg[f"{t}_inverse"])
        kept = gg[f"{t}_kept_index"]
        assert np.array_equal(out["grid_coord"][:n].cpu().numpy()[kept], gg[f"{t}_kept_grid"])
        nv0 = int(out["voxel_offset"][0])
        assert nv0 == kept.shape[0] and int(out["count"][:nv0].sum()) == n
        if mode == "train":
            sel = out["idx_unique"].cpu().numpy()
            assert np.array_equal(np.sort(inv[sel[:nv0]]), np.arange(nv0))          # exactly one point per voxel of scene 0
            assert (sel[:nv0] < n).all() and (sel[nv0:] >= n).all()                   # scenes never mix
        else:
            frags = [f.cpu().numpy() for f in out["fragments"]]
            assert len(frags) == int(out["count"].max())
            seen = np.zeros(2 * n, dtype=bool)
            for f in frags:
                assert np.array_equal(np.sort(out["inverse"].cpu().numpy()[f[:nv0]]), np.arange(nv0))
                seen[f] = True
            assert seen.all()                                                          # every point is in some fragment


@pytest.mark.gpu
def test_gpu_float32_division_variant_and_size_properties():
    """NumPy 1.x promotion (float32 division) against the oracle, and a 2 x 1M-point batch through size-independent properties."""
    from oracle import voxel
    from pointcloudpdf_amd import voxelize

    sc = dense_scene(41, 30000)
    k32, g32, inv32, cnt32, _ = voxel.grid_partition(sc, 0.04, float32_division=True)
    out = voxelize.grid_sample(torch.from_numpy(sc).cuda(), torch.tensor([30000], dtype=torch.int32, device="cuda"), 0.04, float32_division=True)
    assert np.array_equal(out["key"].cpu().numpy().view(np.uint64), k32) and np.array_equal(out["inverse"].cpu().numpy(), inv32)
    assert np.array_equal(out["count"].cpu().numpy(), cnt32)
    big = torch.from_numpy(np.concatenate([dense_scene(50, 1000000), dense_scene(51, 1000000)])).cuda()
    off = torch.tensor([1000000, 2000000], dtype=torch.int32, device="cuda")
    o = voxelize.grid_sample(big, off, 0.04)
    assert int(o["count"].sum()) == 2000000 and int(o["voxel_offset"][-1]) == o["count"].shape[0]
    ks = o["key"][o["order"]] ^ (-(2 ** 63))
    sc_id = torch.bucketize(o["order"], off.long(), right=True)
    assert bool(((ks[1:] >= ks[:-1]) | (sc_id[1:] != sc_id[:-1])).all()) and bool((sc_id[1:] >= sc_id[:-1]).all())   # sorted per scene
    kept_grid = o["grid_coord"][o["idx_unique"]]
    assert kept_grid.shape[0] == o["count"].shape[0] and bool((kept_grid >= 0).all())
