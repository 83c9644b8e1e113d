"""The PDF pseudo-label pass (pointcept/recognizers/ours/pointpdf_v1m1_base.py:118-382, helpers
ours/utils.py:7-43, 87-131) on device tensors.

Per scene: seeds are drawn among the least-confident points, the seed set grows over a fixed-radius neighbour table ("heuristic
search": candidates = unvisited neighbours, ranked by 0.4 * closeness to the region's centroid + 0.6 * similarity of their score to
the region's, the best 40 % join) until the region's mean score passes mean - beta * std; the region is then pruned through a
minimum spanning tree over neighbour similarities (weak edges = left outliers of the heavier GMM component are cut, only the
outlier-LARGE connected components survive).  The result is the boolean ``pseudo_mask`` that ``PointPdfV1.forward`` turns into the
extra "unknown" label.

What runs where: the neighbour table comes from the HIP radius query (``radius_neighbors``); region growing is torch indexing on the
device; MST / GMM / connected components stay on scipy / sklearn on the host exactly as upstream (it moves the tensors with .cpu()
there too) -- they see a few thousand edges.  Upstream spreads scenes over joblib workers; here they run on worker threads.

Parity: ``pseudo_labeling`` is pinned against the reference's OWN static method (tests/golden/ops_pseudo_label_ref.npz, same
neighbour table, same torch / numpy seeds).  The neighbour table itself replaces ``torch_points_kernels.ball_query(radius,
max_neighbor, x, x, mode="partial_dense")`` -- an unvendored, unversioned dependency (README.md:105), absent here: its documented
behaviour (the first ``max_neighbor`` points of the same scene, in index order, with d2 < radius^2, padded with -1) is what
``radius_neighbors`` implements; that part is "parity unpinned".
"""
import numpy as np
import torch

from . import _native


def radius_neighbors(coord, offset, radius, max_neighbor):
    """-> (N, max_neighbor) int64 neighbour ids (global rows, the point itself included), -1 padded."""
    be = _native.backend_for(coord)
    off = offset.int().contiguous()
    if hasattr(be, "radius_neighbors_self"):       # HIP: 27 grid cells around every point instead of the whole scene (same results)
        idx, _ = be.radius_neighbors_self(int(max_neighbor), float(radius), coord.contiguous(), off)
    else:
        order = torch.arange(coord.shape[0], dtype=torch.int32, device=coord.device)   # identity permutation: index order
        idx, _ = be.ball_query(int(max_neighbor), float(radius), 0.0, coord.contiguous(), coord.contiguous(), off, off, order=order)
    return idx.long()


def _pair_similarity(node, node_nn, coord, score):
    """ours/utils.py:7-43: per (node, neighbour) 0.4 * distance similarity + 0.6 * confidence similarity; -10 marks invalid pairs."""
    valid = (node_nn != -1) & (node_nn != node[:, None])
    dist = torch.norm(coord[node_nn] - coord[node, None], dim=-1)
    masked = torch.where(valid, dist, torch.zeros((), device=dist.device, dtype=dist.dtype))
    dmin, dmax = masked.min(-1)[0][:, None], masked.max(-1)[0][:, None]
    dist_sim = torch.where(valid, 1 - (dist - dmin) / (dmax - dmin + 1e-3), torch.full((), -10.0, device=dist.device))
    conf_sim = torch.where(valid, torch.exp(-torch.abs(score[node_nn] - score[node, None])), torch.full((), -10.0, device=dist.device))
    return 0.4 * dist_sim + 0.6 * conf_sim


def _grow_region(coord, score, neighbors, seeds, stop, slide_window):
    """pointpdf_v1m1_base.py:233-305"""
    graph = seeds
    n = coord.shape[0]
    while True:
        g_coord, g_score = coord[graph], score[graph]
        if g_score.mean(0) > stop and len(graph) > 0.01 * n and len(graph) > 50:
            break
        cand = torch.unique(neighbors[graph])
        cand = cand[(cand != -1) & ~torch.isin(cand, graph)]
        dist = torch.norm(coord[cand] - g_coord.mean(0), dim=-1)
        dist_sim = 1 - (dist - dist.min()) / (dist.max() - dist.min() + 1e-3)
        if slide_window:
            lo = torch.kthvalue(g_score, int(len(g_score) * 0.1)).values
            hi = torch.kthvalue(g_score, int(len(g_score) * 0.6)).values
        else:
            lo, hi = g_score.min(), g_score.max()
        conf_sim = torch.exp(-torch.abs(score[cand] - g_score[(g_score >= lo) & (g_score <= hi)].mean(0)))
        sim = 0.4 * dist_sim + 0.6 * conf_sim
        take = torch.topk(sim.view(-1), k=int(sim.numel() * 0.4))[1]
        grown = torch.unique(torch.cat([graph, cand.view(-1)[take]]))
        grown = grown[grown != -1]
        if grown.shape[0] == graph.shape[0]:
            break
        graph = grown
    return graph


def _prune_by_spanning_tree(coord, msp, neighbors, node):
    """pointpdf_v1m1_base.py:309-380: MST over neighbour similarities inside the region, weak edges cut, large components kept."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components, minimum_spanning_tree
    from sklearn.mixture import GaussianMixture

    node_nn = neighbors[node]
    sim = _pair_similarity(node, node_nn, coord, msp)
    n = coord.shape[0]
    keep = ((node_nn != -1) & torch.isin(node_nn, node) & (node_nn != node[:, None])).flatten()
    rows = node.repeat_interleave(node_nn.shape[1])[keep].cpu()
    cols = node_nn.reshape(-1)[keep].cpu()
    adj = csr_matrix((sim.reshape(-1)[keep].cpu(), (rows, cols)), shape=(n, n))
    mst = minimum_spanning_tree(adj)
    w = mst.data
    gmm = GaussianMixture(n_components=2).fit(w.reshape(-1, 1))
    means, covs = gmm.means_.flatten(), gmm.covariances_.flatten()
    top = int(np.argmax(means))
    lower = means[top] - 2.0 * covs[top]                       # z_score_filter_np(..., "left", 2.0) with std := the covariance, as upstream
    mst.data[~(w < lower)] = 0                                  # upstream zeroes the edges that are NOT left outliers
    mst.eliminate_zeros()
    _, label = connected_components(mst, directed=False)
    touched = torch.unique(torch.cat([node, node_nn.reshape(-1)]))[1:].cpu()     # [1:]: drops the -1 padding (first after the sort)
    labels, sizes = np.unique(label[touched], return_counts=True)
    big = (sizes - sizes.mean()) / sizes.std() > 2.0            # z_score_mask_np(..., area="right", score=2.0)
    accept = np.where(np.isin(label, labels[big]))[0]
    mask = torch.zeros(n, dtype=torch.bool)
    mask[accept] = True
    return mask


@torch.no_grad()
def pseudo_labeling(coord, logits, neighbors, condition_from="msp", beta=1.5, seed_from="ml", seed_range=0.15, num_seed=100,
                    slide_window=True, generator=None):
    """One scene (local neighbour ids) -> bool mask (n,) on the host, like upstream's static method (:187-382).  The seed draw uses
    a CPU generator (upstream: the global one), the GMM numpy's global state."""
    msp = torch.softmax(logits, dim=-1).max(dim=-1)[0]
    ml = logits.max(dim=-1)[0]
    ml = (ml - ml.min()) / (ml.max() - ml.min() + 1e-6)
    score = msp if condition_from == "msp" else ml
    stop = torch.mean(score) - beta * torch.std(score)
    src = msp if seed_from == "msp" else ml
    dice = torch.randint(0, int(seed_range * len(src)), [num_seed], generator=generator)
    seeds = torch.sort(src, dim=-1)[1][dice.to(src.device)]
    region = _grow_region(coord, score, neighbors, seeds, stop, slide_window)
    return _prune_by_spanning_tree(coord, msp, neighbors, region)


@torch.no_grad()
def get_pseudo_mask(coord, seg_logits, offset, radius=0.1, max_neighbor=64, neighbors=None, offset_host=None, generator=None, workers=4, **kw):
    """pointpdf_v1m1_base.py:118-185 for a batch: neighbour table once, scenes one by one; -> bool (N,) on coord's device."""
    if neighbors is None:
        neighbors = radius_neighbors(coord, offset, radius, max_neighbor)
    ends = offset_host if offset_host is not None else [int(v) for v in offset.tolist()]
    starts = [0] + ends[:-1]
    stream = torch.cuda.current_stream() if coord.is_cuda else None

    def one(se):
        s0, e = se
        nn = neighbors[s0:e].clone()
        nn[nn != -1] -= s0
        return pseudo_labeling(coord[s0:e], seg_logits[s0:e], nn, generator=generator, **kw)

    def one_on_stream(se):   # worker threads start on the default stream: keep them on the caller's
        with torch.cuda.stream(stream):
            return one(se)

    scenes = list(zip(starts, ends))
    if workers > 1 and len(scenes) > 1 and generator is None:
        # upstream fans the scenes out over joblib workers (parallel_processing, n_jobs=4); threads here: the scipy / sklearn stage
        # and the device syncs release the GIL.  Not used with an explicit generator (the draw order would depend on scheduling).
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(workers, len(scenes))) as ex:
            masks = list(ex.map(one_on_stream if stream is not None else one, scenes))
    else:
        masks = [one(se) for se in scenes]
    return torch.cat(masks).to(coord.device)


def make_pseudo_mask_fn(radius=0.1, max_neighbor=64, **kw):
    """A ``pseudo_mask_fn(coord, seg_logits, offset)`` for ``recognizer.PointPdfV1`` / ``engine.OpenSegStep`` built from the
    recognizer section of configs/scannet/openseg-pt-v1-0-pointpdf-v1m1-base.py:40-58 (kp_ball_radius, kp_max_neighbor,
    condition_from, beta, seed_from, seed_range, num_seed, slide_window)."""
    def fn(coord, seg_logits, offset):
        return get_pseudo_mask(coord, seg_logits.detach(), offset, radius=radius, max_neighbor=max_neighbor, **kw)
    return fn
