"""Data contract either side of the hot path, on the device (second half): ``SphereCrop`` and the offset
collate -- pointcept/datasets/transform.py:929-1025 and pointcept/datasets/utils.py:15-56.

With GridSample (voxelize.py) this closes the per-batch input pipeline voxelise -> crop -> collate without leaving the GPU: at
9 M points/s per GPU the numpy pipeline upstream (argsort of every scene per sample) cannot feed eight GPUs from one host.

* ``sphere_crop`` crops a whole batch of scenes in one pass: squared distances to the scene's centre point, one stable device sort
  by (scene, distance), the first ``point_max`` rows of every scene kept.  Upstream's ``np.argsort`` is unstable, so WHICH of
  several equidistant points survives at the cut is undefined there; everything else (the kept set, the ascending-distance order
  of the kept points, "random" / "center" centre choice, scenes at or below ``point_max`` left untouched) is reproduced.
* ``collate_fn`` / ``point_collate_fn`` are upstream's recursion (tensors concatenated, every key containing "offset" turned into
  cumulative ends, Mix3D merge of neighbouring scenes).
Plumbing on torch device ops; the arithmetic is a three-term fp32 sum per point -- nothing here is worth a hand-written kernel.
"""
import random
from collections.abc import Mapping, Sequence

import torch

_CROP_KEYS = ("coord", "origin_coord", "grid_coord", "color", "normal", "segment", "instance", "displacement", "strength")


def sphere_crop(data, offset, point_max=80000, sample_rate=None, mode="random", generator=None, centers=None):
    """data: dict of (N, ...) device tensors with "coord" (N, 3); offset (B,) cumulative ends (host list or tensor).
    -> (cropped dict, new offset int32 tensor, kept row indices int64).  ``mode``: "random" | "center" (transform.py:995-1000);
    ``centers`` (B,) row indices override the choice (tests)."""
    coord = data["coord"]
    dev = coord.device
    ends = [int(v) for v in (offset.tolist() if isinstance(offset, torch.Tensor) else offset)]
    starts = [0] + ends[:-1]
    sizes = [e - s for s, e in zip(starts, ends)]
    limits = [int(sample_rate * n) if sample_rate is not None else int(point_max) for n in sizes]
    if centers is None:
        if mode == "center":
            centers = [s + n // 2 for s, n in zip(starts, sizes)]
        elif mode == "random":
            centers = [s + int(torch.randint(n, (1,), generator=generator).item()) for s, n in zip(starts, sizes)]
        else:
            raise NotImplementedError(f"sphere_crop: mode {mode!r} (the sliding 'all' mode of the test pipeline is not a per-batch op)")
    sizes_t = torch.tensor(sizes, device=dev)
    scene = torch.repeat_interleave(torch.arange(len(sizes), device=dev), sizes_t, output_size=coord.shape[0])
    c = coord[torch.tensor(centers, device=dev)][scene]
    d2 = torch.sum(torch.square(coord - c), 1)                                   # np.sum(np.square(coord - center), 1)
    crop = torch.tensor([n > lim for n, lim in zip(sizes, limits)], device=dev)[scene]
    rows = torch.arange(coord.shape[0], device=dev)
    # scenes that are cropped: ascending distance (stable: ties by row); scenes at or below the limit: untouched, original order
    key = torch.where(crop, d2, torch.zeros_like(d2))
    order = torch.argsort(key, stable=True)
    order = order[torch.argsort(scene[order], stable=True)]
    start_t = torch.tensor(starts, device=dev)[scene[order]]
    rank = torch.arange(coord.shape[0], device=dev) - start_t
    keep = rank < torch.tensor(limits, device=dev)[scene[order]]
    kept = order[keep]
    # untouched scenes keep their original order (argsort of the all-zero key is the identity under a stable sort)
    out = {k: (v[kept] if (k in _CROP_KEYS and isinstance(v, torch.Tensor)) else v) for k, v in data.items()}
    new_sizes = [min(n, lim) for n, lim in zip(sizes, limits)]
    new_offset = torch.cumsum(torch.tensor(new_sizes, device=dev), 0).int()
    return out, new_offset, kept


def collate_fn(batch):
    """pointcept/datasets/utils.py:15-41: tensors are concatenated; a list sample gets its length appended and the last column
    becomes the cumulative offset; a dict is collated key by key and every key containing "offset" is accumulated."""
    if not isinstance(batch, Sequence):
        raise TypeError(f"{type(batch)} is not supported.")
    if isinstance(batch[0], torch.Tensor):
        return torch.cat(list(batch))
    if isinstance(batch[0], str):
        return list(batch)
    if isinstance(batch[0], Sequence):
        for data in batch:
            data.append(torch.tensor([data[0].shape[0]], device=data[0].device))
        batch = [collate_fn(samples) for samples in zip(*batch)]
        batch[-1] = torch.cumsum(batch[-1], dim=0).int()
        return batch
    if isinstance(batch[0], Mapping):
        out = {key: collate_fn([d[key] for d in batch]) for key in batch[0]}
        for key in out.keys():
            if "offset" in key:
                out[key] = torch.cumsum(out[key], dim=0)
        return out
    from torch.utils.data.dataloader import default_collate

    return default_collate(batch)


def point_collate_fn(batch, mix_prob=0):
    """pointcept/datasets/utils.py:44-56 (Mix3D: neighbouring scenes merged pairwise with probability ``mix_prob``)."""
    assert isinstance(batch[0], Mapping)
    batch = collate_fn(batch)
    if "offset" in batch.keys():
        if random.random() < mix_prob:
            batch["offset_ori"] = batch["offset"].clone()
            batch["offset"] = torch.cat([batch["offset"][1:-1:2], batch["offset"][-1].unsqueeze(0)], dim=0)
    return batch
