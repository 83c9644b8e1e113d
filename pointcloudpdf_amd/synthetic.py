"""Synthetic voxelised indoor scenes + deterministic weights (no datasets / checkpoints: there is no network).

Scene recipe: an axis-aligned room (S3DIS-like 10 x 8 x 3 m or ScanNet-like 6 x 5 x 2.7 m) with
floor, ceiling, four walls and 6-12 furniture boxes; surfaces are sampled densely with +-2 mm jitter and then pushed
through the reference's voxelisation semantics -- ``grid = floor(coord / grid_size)``, 64-bit FNV key over the three
ints (multiply-then-xor order, pointcept/datasets/transform.py:911-925), ONE random real-valued point per occupied
voxel (:826-830, train mode) -- so coordinates stay un-snapped and fp32 distance ties are as rare as in real data.
Then: nearest-to-centre crop to exactly ``n`` points (SphereCrop, :1004-1006), shuffle (:1029-1048), positive shift
(:139-144); colour ~ U[0,1]^3, normals = face normals, label = surface id mod num_classes.
``snap=True`` snaps coordinates to voxel centres instead (forces fp32 ties; parity tests only, never timed).
"""
import zlib

import numpy as np
import torch

_FNV_OFFSET = np.uint64(14695981039346656037)
_FNV_PRIME = np.uint64(1099511628211)


def fnv_hash_vec(arr):
    """transform.py:911-925 (docstring there says FNV64-1A; the loop multiplies THEN xors -- reproduce the code)."""
    arr = np.asarray(arr).astype(np.uint64, copy=True)
    h = _FNV_OFFSET * np.ones(arr.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(arr.shape[1]):
            h = h * _FNV_PRIME
            h = np.bitwise_xor(h, arr[:, j])
    return h


def _sample_rect(rng, origin, u, v, density):
    """Points on the parallelogram origin + a*u + b*v, a,b in [0,1], about `density` points per m^2."""
    area = np.linalg.norm(np.cross(u, v))
    cnt = max(int(area * density), 8)
    ab = rng.random((cnt, 2))
    return origin[None, :] + ab[:, :1] * u[None, :] + ab[:, 1:] * v[None, :]


def make_scene(n_points, scene_id=0, kind="s3dis", grid_size=None, num_classes=None, snap=False, seed=2024):
    """-> dict(coord (n,3) f32, color (n,3) f32, normal (n,3) f32, segment (n,) int64). 2024 = reference default seed."""
    rng = np.random.default_rng(seed + scene_id)
    if kind == "s3dis":
        room, gs, nc = np.array([10.0, 8.0, 3.0]), 0.04, 13
    elif kind == "scannet":
        room, gs, nc = np.array([6.0, 5.0, 2.7]), 0.02, 20
    else:
        raise ValueError(kind)
    gs = grid_size or gs
    nc = num_classes or nc
    # one point per occupied voxel on ~2*(xy+xz+yz) m^2 of surface: pick the density so enough voxels are hit
    faces = []  # (origin, u, v, normal)
    X, Y, Z = room
    ex, ey, ez = np.eye(3)
    faces += [(np.zeros(3), X * ex, Y * ey, ez), (Z * ez, X * ex, Y * ey, -ez)]
    faces += [(np.zeros(3), X * ex, Z * ez, ey), (Y * ey, X * ex, Z * ez, -ey)]
    faces += [(np.zeros(3), Y * ey, Z * ez, ex), (X * ex, Y * ey, Z * ez, -ex)]
    for _ in range(int(rng.integers(6, 13))):
        size = rng.uniform([0.4, 0.4, 0.3], [2.0, 1.6, 1.4])
        org = rng.uniform([0.1, 0.1, 0.0], np.maximum(room - size - 0.1, 0.2))
        sx, sy, sz = size
        faces += [(org + sz * ez, sx * ex, sy * ey, ez)]
        faces += [(org, sx * ex, sz * ez, -ey), (org + sy * ey, sx * ex, sz * ez, ey)]
        faces += [(org, sy * ey, sz * ez, -ex), (org + sx * ex, sy * ey, sz * ez, ex)]
    total_area = sum(np.linalg.norm(np.cross(u, v)) for _, u, v, _ in faces)
    need_voxels = n_points * 1.35
    if total_area / (gs * gs) < need_voxels:  # room too small for the request at this grid: shrink the grid
        gs = float(np.sqrt(total_area / need_voxels))
    density = 2.5 / (gs * gs)
    pts, nrm, lab = [], [], []
    for sid, (o, u, v, nvec) in enumerate(faces):
        q = _sample_rect(rng, o, u, v, density)
        q = q + rng.uniform(-0.002, 0.002, q.shape)
        pts.append(q)
        nrm.append(np.repeat(nvec[None, :], q.shape[0], 0))
        lab.append(np.full(q.shape[0], sid % nc, dtype=np.int64))
    coord = np.concatenate(pts)
    normal = np.concatenate(nrm)
    segment = np.concatenate(lab)
    # GridSample(mode="train"): one random point per voxel
    grid = np.floor(coord / gs).astype(np.int64)
    grid -= grid.min(0)
    key = fnv_hash_vec(grid)
    order = np.argsort(key, kind="stable")
    _, count = np.unique(key[order], return_counts=True)
    pick = np.cumsum(np.insert(count, 0, 0)[:-1]) + rng.integers(0, count.max(), count.size) % count
    sel = order[pick]
    coord, normal, segment, grid = coord[sel], normal[sel], segment[sel], grid[sel]
    if snap:
        coord = (grid.astype(np.float64) + 0.5) * gs
    if coord.shape[0] < n_points:
        raise RuntimeError(f"synthetic scene has only {coord.shape[0]} voxels, {n_points} requested")
    # SphereCrop(mode="center"): n nearest to the centre point
    centre = coord[coord.shape[0] // 2]
    keep = np.argsort(np.sum((coord - centre) ** 2, 1), kind="stable")[:n_points]
    keep = keep[rng.permutation(n_points)]  # ShufflePoint
    coord, normal, segment = coord[keep], normal[keep], segment[keep]
    coord = coord - coord.min(0)  # PositiveShift
    color = rng.random((n_points, 3))
    return dict(coord=coord.astype(np.float32), color=color.astype(np.float32),
                normal=normal.astype(np.float32), segment=segment)


def make_batch(sizes, first_scene_id=0, kind="s3dis", device="cpu", snap=False, unknown=(5, 9), grid_size=None):
    """Collate scenes the way pointcept/datasets/utils.py:34-39 does: concatenate, offset = cumsum(n_i).
    feat = cat(coord, color[, normal]); segment_known-style labels: ``unknown`` classes -> -1."""
    scenes = [make_scene(n, first_scene_id + i, kind=kind, snap=snap, grid_size=grid_size) for i, n in enumerate(sizes)]
    coord = np.concatenate([s["coord"] for s in scenes])
    feats = [coord, np.concatenate([s["color"] for s in scenes])]
    if kind == "scannet":
        feats.append(np.concatenate([s["normal"] for s in scenes]))
    segment = np.concatenate([s["segment"] for s in scenes])
    segment = np.where(np.isin(segment, list(unknown)), -1, segment)
    offset = np.cumsum([int(n) for n in sizes]).astype(np.int32)
    return dict(
        coord=torch.from_numpy(coord).to(device),
        feat=torch.from_numpy(np.concatenate(feats, 1)).to(device),
        segment=torch.from_numpy(segment).to(device),
        offset=torch.from_numpy(offset).to(device),
        offset_host=[int(v) for v in offset],
    )


def fill_parameters_deterministic(module, seed=0):
    """Closed-form, name-keyed values for every parameter and buffer (no dependence on torch's init RNG stream):
    weights ~ U(+-1/sqrt(fan_in)); 1-D '.weight' (norm scales) ~ 1 + 0.1 U; biases ~ 0.1 U;
    running_mean ~ 0.1 U, running_var ~ 1 + 0.1 |U|, counters 0.  The golden fixtures were produced by applying the
    same function to the reference modules (tests/golden/make_golden.py)."""
    state = module.state_dict()
    new = {}
    for name, t in state.items():
        if name.endswith("K_points"):   # geometric constants (KPConv kernel-point layout), not weights
            new[name] = t
            continue
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        u = rs.uniform(-1.0, 1.0, size=tuple(t.shape)) if t.numel() else np.zeros(tuple(t.shape))
        if name.endswith("num_batches_tracked"):
            v = np.zeros(tuple(t.shape))
        elif name.endswith("running_mean"):
            v = 0.1 * u
        elif name.endswith("running_var"):
            v = 1.0 + 0.1 * np.abs(u)
        elif t.dim() >= 2:
            v = u / np.sqrt(t.shape[1])
        elif name.endswith(".weight"):
            v = 1.0 + 0.1 * u
        else:
            v = 0.1 * u
        new[name] = torch.from_numpy(np.asarray(v)).to(dtype=t.dtype, device=t.device)
    module.load_state_dict(new)
    return module
