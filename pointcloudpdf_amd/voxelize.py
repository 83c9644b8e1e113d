"""GridSample on the GPU for a whole batch of scenes (pointcept/datasets/transform.py:786-925).

The reference voxelises every scene on the CPU inside the DataLoader workers: floor(coord / grid_size), a 64-bit FNV key per
point, ``np.argsort`` + ``np.unique`` and one random point per occupied voxel ("train" mode) or count.max() interleaved fragments
("test" mode).  Here the key pass is one HIP kernel over all scenes of the batch (``pdf_grid_hash``), the ordering is two stable
device sorts (key, then scene) and everything downstream is segmented arithmetic on the sorted keys -- no host round trip except
the per-scene sizes.

What is bit-identical to the reference (tests/golden/ops_gridsample_ref.npz, made by running the reference's own GridSample):
the uint64 keys, ``grid_coord``, the voxel partition (``inverse`` = rank of the point's voxel among the scene's voxels sorted by
key) and ``count``.  What is not defined upstream and therefore only checked as a property: WHICH point of a voxel is kept --
``np.argsort`` is not stable, and train mode adds ``np.random.randint``; here the order inside a voxel is the original point
order (stable sorts) and the draw comes from a ``torch.Generator``.
"""
import torch

from . import _native

_SIGN = -(2 ** 63)


def grid_sample(coord, offset, grid_size, mode="train", generator=None, float32_division=False, offset_host=None):
    """coord (N,3) f32 (device), offset (B) cumulative ends ->  dict with
         key (N) int64 (uint64 FNV key bits), grid_coord (N,3) int64 scene-relative, order (N) point ids scene-major / key-sorted,
         inverse (N) voxel id of every point (per-scene numbering like upstream: rank among the scene's sorted unique keys),
         count (V) points per voxel, voxel_offset (B) cumulative voxel counts,
         train: idx_unique (V) one kept point per voxel;   test: fragments = list of (V,) index tensors (transform.py:858-884).
    ``float32_division`` evaluates coord / grid_size in float32 (NumPy 1.x promotion) instead of float64 (NumPy >= 2)."""
    assert mode in ("train", "test")
    be = _native.backend_for(coord)
    n, b = coord.shape[0], offset.shape[0]
    ends = offset_host if offset_host is not None else [int(v) for v in offset.tolist()]
    gs = [float(grid_size)] * 3 if not hasattr(grid_size, "__len__") else [float(g) for g in grid_size]
    starts = [0] + ends[:-1]
    mins = torch.stack([coord[s:e].amin(0) if e > s else coord.new_zeros(3) for s, e in zip(starts, ends)])   # (B,3)
    g = torch.tensor(gs, dtype=torch.float32 if float32_division else torch.float64, device=coord.device)
    min_grid = torch.floor(mins.to(g.dtype) / g).long().contiguous()          # floor is monotone: = min over the scene of floor(c / g)
    grid, key = be.grid_hash(coord.contiguous(), offset.int().contiguous(), gs, min_grid, float32_division)
    scene = torch.bucketize(torch.arange(n, device=coord.device), offset.long(), right=True)
    skey = key ^ _SIGN                                   # unsigned order under a signed sort
    o1 = torch.sort(skey, stable=True)[1]
    o2 = torch.sort(scene[o1], stable=True)[1]
    order = o1[o2]                                       # scene-major, key ascending, original order inside a voxel
    ks, ss = key[order], scene[order]
    new_voxel = torch.ones(n, dtype=torch.bool, device=coord.device)
    if n > 1:
        new_voxel[1:] = (ks[1:] != ks[:-1]) | (ss[1:] != ss[:-1])
    vid = torch.cumsum(new_voxel, 0) - 1                 # global voxel id along `order`
    nv = int(vid[-1].item()) + 1 if n else 0
    count = torch.bincount(vid, minlength=nv)
    vstart = torch.cumsum(count, 0) - count              # first position (in `order`) of every voxel
    vscene = ss[vstart]
    voxel_offset = torch.cumsum(torch.bincount(vscene, minlength=b), 0)
    first_voxel = torch.cat([voxel_offset.new_zeros(1), voxel_offset[:-1]])
    inverse = torch.empty(n, dtype=torch.long, device=coord.device)
    inverse[order] = vid - first_voxel[ss]               # per-scene voxel rank, like np.unique(..., return_inverse) per scene
    out = dict(key=key, grid_coord=grid, order=order, inverse=inverse, count=count, voxel_offset=voxel_offset.int(), voxel_scene=vscene)
    if mode == "train":                                  # transform.py:824-829
        cmax = int(count.max().item()) if nv else 1
        dice = torch.randint(0, max(cmax, 1), (nv,), generator=generator, device=coord.device) % count
        out["idx_unique"] = order[vstart + dice]
    else:                                                # transform.py:858-861
        cmax = int(count.max().item()) if nv else 0
        out["fragments"] = [order[vstart + (i % count)] for i in range(cmax)]
    return out
