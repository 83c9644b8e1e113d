"""CPU oracle of the StratifiedTransformer's window edge tables -- TEST INFRASTRUCTURE ONLY.

The reference's own construction, restated statement by statement on torch CPU tensors (the reference is torch code):
pointcept/models/stratified_transformer/stratified_transformer_v1m1_origin.py:103-127 (``grid_sample``: voxel -> member table padded to
the fullest voxel), :45-100 (``get_indice_pairs``: dense pair expansion through boolean masks), :500-536 (``BasicLayer.forward``: the
stable sort of the edges by query, ``index_0_offsets``, ``n_max``) and :282-292 (``WindowAttention``: quantised relative positions).
The product builds the same tables per query on the device (pointcloudpdf_amd/csrc/window_edges.hip); tests compare the two bit for bit.
Pinned by: tests/golden/model_stratified.npz (the reference's own StratifiedTransformer run by tests/golden/make_golden.py).
"""
import torch


def p2v_from_keys(cluster_key):
    """:103-127 after ``voxel_grid``: (voxel -> member points) table padded to the largest voxel, member counts."""
    unique, cluster, counts = torch.unique(cluster_key, sorted=True, return_inverse=True, return_counts=True)
    n, k = unique.shape[0], int(counts.max().item())
    p2v_map = cluster.new_zeros(n, k)
    mask = torch.arange(k).unsqueeze(0) < counts.unsqueeze(-1)
    p2v_map[mask] = torch.argsort(cluster, stable=True)
    return p2v_map, counts


def get_indice_pairs(p2v_map, counts, new_p2v_map, new_counts, downsample_idx, n_points, window_cell_key):
    """:45-100 -- edge list (index_0 = query, index_1 = key): all pairs inside a window of the fine partition, plus, from the coarse
    (2 x window) partition, the pairs whose key is an FPS-downsampled point lying in a DIFFERENT fine window (``window_cell_key``: the
    reference's ``window_coord`` triple of :91-94 packed into one integer per point)."""
    n, k = p2v_map.shape
    mask = torch.arange(k).unsqueeze(0) < counts.unsqueeze(-1)
    mask_mat = mask.unsqueeze(-1) & mask.unsqueeze(-2)
    index_0 = p2v_map.unsqueeze(-1).expand(-1, -1, k)[mask_mat]
    index_1 = p2v_map.unsqueeze(1).expand(-1, k, -1)[mask_mat]

    downsample_mask = torch.zeros(n_points, dtype=torch.bool)
    downsample_mask[downsample_idx.long()] = True
    downsample_mask = downsample_mask[new_p2v_map]
    n, k = new_p2v_map.shape
    mask = torch.arange(k).unsqueeze(0) < new_counts.unsqueeze(-1)
    downsample_mask = downsample_mask & mask
    mask_mat = mask.unsqueeze(-1) & downsample_mask.unsqueeze(-2)
    key = window_cell_key[new_p2v_map]
    mask_mat_prev = key.unsqueeze(2) != key.unsqueeze(1)
    mask_mat = mask_mat & mask_mat_prev
    new_index_0 = new_p2v_map.unsqueeze(-1).expand(-1, -1, k)[mask_mat]
    new_index_1 = new_p2v_map.unsqueeze(1).expand(-1, k, -1)[mask_mat]
    return torch.cat([index_0, new_index_0], 0), torch.cat([index_1, new_index_1], 0)


def window_edges(xyz, kf, kc, wk, downsample_idx, c2w, qs, vmax):
    """Same interface as ``HipBackend.window_edges`` (pointcloudpdf_amd/_native.py)."""
    n = xyz.shape[0]
    p2v, cnt = p2v_from_keys(kf)
    new_p2v, new_cnt = p2v_from_keys(kc)
    index_0, index_1 = get_indice_pairs(p2v, cnt, new_p2v, new_cnt, downsample_idx, n, wk)
    index_0, indices = torch.sort(index_0, stable=True)   # :507 (CSR by query)
    index_1 = index_1[indices]
    counts = torch.bincount(index_0, minlength=n)
    n_max = int(counts.max()) if n else 0
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(dim=-1)], 0).int()
    rel = xyz[index_0] - xyz[index_1]                                                        # :282-292
    rel = torch.round(rel * 100000) / rel.new_tensor(100000.0)
    rel = torch.div(rel + float(c2w) - 1e-4, rel.new_tensor(float(qs)), rounding_mode="trunc")
    flag = torch.tensor([int(not ((rel >= 0).all() and (rel <= vmax).all()))], dtype=torch.int32)
    return index_0, index_1.int().contiguous(), offsets.contiguous(), n_max, rel.int().contiguous(), flag
