"""The PDF pseudo-label pass (pointcept/recognizers/ours/pointpdf_v1m1_base.py:118-382, helpers
ours/utils.py:7-43, 87-131) on device tensors.

Per scene: seeds are drawn among the least-confident points, the seed set grows over a fixed-radius neighbour table ("heuristic
search": candidates = unvisited neighbours, ranked by 0.4 * closeness to the region's centroid + 0.6 * similarity of their score to
the region's, the best 40 % join) until the region's mean score passes mean - beta * std; the region is then pruned through a
minimum spanning tree over neighbour similarities (weak edges = left outliers of the heavier GMM component are cut, only the
outlier-LARGE connected components survive).  The result is the boolean ``pseudo_mask`` that ``PointPdfV1.forward`` turns into the
extra "unknown" label.

What runs where: the neighbour table comes from the HIP radius query (``radius_neighbors``); region growing is torch indexing on the
device; spanning tree / mixture fit / connected components are single-workgroup HIP kernels for device tensors
(``_prune_by_spanning_tree_hip``, csrc/graph_prune.hip) and scipy / sklearn on the host exactly as upstream for CPU tensors (the
reference-parity path).  Upstream spreads scenes over joblib workers; here they run on worker threads.

Parity: ``pseudo_labeling`` is pinned against the reference's OWN static method (tests/golden/ops_pseudo_label_ref.npz, same
neighbour table, same torch / numpy seeds).  The neighbour table itself replaces ``torch_points_kernels.ball_query(radius,
max_neighbor, x, x, mode="partial_dense")`` -- an unvendored, unversioned dependency (README.md:105), absent here: its documented
behaviour (the first ``max_neighbor`` points of the same scene, in index order, with d2 < radius^2, padded with -1) is what
``radius_neighbors`` implements; that part is "parity unpinned".
The device pruning path's mixture fit (``pdf_gmm2_1d``: EM in double) is **mixture unpinned vs sklearn** in one respect only: its 2-means
start is deterministic (quartiles) where upstream's ``GaussianMixture(n_components=2)`` (:334-336, no random state) draws a k-means++ start
from numpy's global generator.  Everything else is sklearn's: k-means to its fixed point, the (E-step, M-step, test) loop, tol 1e-3, at most
100 iterations, covariance floor 1e-6 (round 6; rounds 1-5 ran the EM to 1e-6, which on trees of a few thousand edges ends at ANOTHER cut
than sklearn's loosely converged one -- found with the round-6 fixtures whose regions really grow: IoU 0.83 against the reference's mask,
0.99+ now).  Upstream's own result moves with the numpy seed (cut 0.7383 or 1.2394 on the 78-edge tree of fixture s2); the host path (CPU
tensors, scipy / sklearn as upstream) is the upstream-exact one.
"""
import ctypes
import os

import numpy as np
import torch

from . import _native


def radius_neighbors(coord, offset, radius, max_neighbor, raw=False):
    """-> (N, max_neighbor) int64 neighbour ids (global rows, the point itself included), -1 padded.  ``raw``: the backend's int32 table
    as it is (what ``get_pseudo_mask_static`` reads; saves two passes over 8 N max_neighbor bytes)."""
    be = _native.backend_for(coord)
    off = offset.int().contiguous()
    if hasattr(be, "radius_neighbors_self"):       # HIP: 27 grid cells around every point instead of the whole scene (same results)
        idx, _ = be.radius_neighbors_self(int(max_neighbor), float(radius), coord.contiguous(), off)
    else:
        order = torch.arange(coord.shape[0], dtype=torch.int32, device=coord.device)   # identity permutation: index order
        idx, _ = be.ball_query(int(max_neighbor), float(radius), 0.0, coord.contiguous(), coord.contiguous(), off, off, order=order)
    return idx if raw else idx.long()


def _pair_similarity(node, node_nn, coord, score):
    """ours/utils.py:7-43: per (node, neighbour) 0.4 * distance similarity + 0.6 * confidence similarity; -10 marks invalid pairs."""
    valid = (node_nn != -1) & (node_nn != node[:, None])
    dist = torch.norm(coord[node_nn] - coord[node, None], dim=-1)
    masked = torch.where(valid, dist, torch.zeros((), device=dist.device, dtype=dist.dtype))
    dmin, dmax = masked.min(-1)[0][:, None], masked.max(-1)[0][:, None]
    dist_sim = torch.where(valid, 1 - (dist - dmin) / (dmax - dmin + 1e-3), torch.full((), -10.0, device=dist.device))
    conf_sim = torch.where(valid, torch.exp(-torch.abs(score[node_nn] - score[node, None])), torch.full((), -10.0, device=dist.device))
    return 0.4 * dist_sim + 0.6 * conf_sim


def _grow_region(coord, score, neighbors, seeds, stop, slide_window, with_flag=False):
    """pointpdf_v1m1_base.py:233-305.  ``with_flag``: also return whether the region grew (it is then sorted and free of repeats).

    Upstream forms the candidate set as ``unique(neighbors[graph])`` minus ``-1`` minus ``isin(., graph)`` and the grown region as
    ``unique(cat(graph, chosen))``: three sorts of up to |graph| x 64 ids per round.  The same SETS, in the same ascending order, come
    out of a membership mask over the scene's points (scatter the ids, clear the members, ``nonzero``): one host read per set instead
    of a sort + a read, every other line as upstream (``_grow_region_reference`` is the line-by-line form; the tests compare them)."""
    graph = seeds
    n = coord.shape[0]
    grew = False
    while True:
        g_coord, g_score = coord[graph], score[graph]
        if g_score.mean(0) > stop and len(graph) > 0.01 * n and len(graph) > 50:
            break
        seen = torch.zeros(n + 1, dtype=torch.bool, device=coord.device)   # (slot n: the -1 padding)
        seen[neighbors[graph].reshape(-1)] = True
        seen[graph] = False
        cand = torch.nonzero(seen[:n]).flatten()
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
        seen.zero_()
        seen[graph] = True
        seen[cand[take]] = True
        grown = torch.nonzero(seen[:n]).flatten()
        if grown.shape[0] == graph.shape[0]:
            break
        graph, grew = grown, True
    return (graph, grew) if with_flag else graph


def _grow_region_reference(coord, score, neighbors, seeds, stop, slide_window, with_flag=False):
    """pointpdf_v1m1_base.py:233-305 line by line (``unique`` / ``isin`` per round): what ``_grow_region`` is tested against."""
    graph = seeds
    n = coord.shape[0]
    grew = False
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
        graph, grew = grown, True
    return (graph, grew) if with_flag else graph


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


# ------------------------------------------------------------------------------------------------------------------
# The pruning stage on the device (round 2): spanning tree and connected components as torch device ops, the 2-component 1-D
# mixture as a closed-form EM on the (few thousand) tree weights.  Upstream moves everything to the host here and calls
# scipy.sparse.csgraph.minimum_spanning_tree / connected_components and sklearn.mixture.GaussianMixture (whose k-means
# initialisation draws from numpy's GLOBAL random state, so upstream's own result is not reproducible run to run).
# ------------------------------------------------------------------------------------------------------------------
def _pointer_jump(parent):
    while True:
        nxt = parent[parent]
        if torch.equal(nxt, parent):
            return parent
        parent = nxt


def minimum_spanning_forest(n, eu, ev, ew):
    """Boruvka on the undirected graph given by directed entries (eu[e], ev[e], ew[e]) -- a pair listed in both directions with two
    weights counts with the smaller one, as scipy's Kruskal over the stored entries does.  Returns the indices (into eu / ev / ew) of the
    chosen entries.  Ties are ordered by (weight, entry index), so equal weights cannot close a cycle; with distinct weights the tree
    is THE minimum spanning forest, i.e. scipy.sparse.csgraph.minimum_spanning_tree's."""
    dev = eu.device
    E = eu.shape[0]
    if E == 0:
        return eu.new_zeros(0)
    u2, v2 = torch.cat([eu, ev]), torch.cat([ev, eu])          # entry e and its mirror e + E
    w2 = torch.cat([ew, ew]).double()
    eid = torch.arange(2 * E, device=dev)
    order_rank = torch.empty(2 * E, dtype=torch.long, device=dev)
    canon = eid % E
    # total order (weight, canonical entry): two stable sorts; rank of every directed copy
    by_canon = torch.argsort(canon, stable=True)
    srt = by_canon[torch.argsort(w2[by_canon], stable=True)]
    order_rank[srt] = torch.arange(2 * E, device=dev)
    # the mirror must share its entry's rank so that both endpoints agree on "the" lightest edge between two components
    order_rank = torch.minimum(order_rank, torch.cat([order_rank[E:], order_rank[:E]]))
    comp = torch.arange(n, device=dev)
    chosen = torch.zeros(E, dtype=torch.bool, device=dev)
    big = 4 * E + 4
    while True:
        cu, cv = comp[u2], comp[v2]
        live = cu != cv
        if not bool(live.any()):
            break
        key = torch.where(live, order_rank, torch.full_like(order_rank, big))
        best = torch.full((n,), big, dtype=torch.long, device=dev).scatter_reduce(0, cu, key, "amin", include_self=True)
        pick = live & (key == best[cu])                         # the lightest outgoing copy of every component (one per component)
        chosen[canon[pick]] = True
        # hook every component onto the component at the other end of its lightest edge; 2-cycles keep the smaller id as root
        parent = torch.arange(n, device=dev)
        parent[cu[pick]] = cv[pick]
        back = parent[parent] == torch.arange(n, device=dev)    # c -> d -> c
        root_fix = back & (torch.arange(n, device=dev) < parent)
        parent = torch.where(root_fix, torch.arange(n, device=dev), parent)
        comp = _pointer_jump(parent)[comp]
    return torch.nonzero(chosen).flatten()


def connected_labels(n, eu, ev):
    """Connected components of an undirected edge list as min-label propagation with pointer jumping -> label (n,) = smallest node id."""
    lab = torch.arange(n, device=eu.device)
    if eu.numel() == 0:
        return lab
    while True:
        lu, lv = lab[eu], lab[ev]
        m = torch.minimum(lu, lv)
        new = lab.scatter_reduce(0, lu, m, "amin", include_self=True).scatter_reduce(0, lv, m, "amin", include_self=True)
        new = _pointer_jump(new)
        if torch.equal(new, lab):
            return lab
        lab = new


def gmm2_1d(x, iters=100, tol=1e-3, reg=1e-6):
    """Two-component 1-D Gaussian mixture by EM (numpy, a few thousand tree weights) with sklearn.mixture.GaussianMixture's own
    defaults and loop -- k-means start run to its fixed point, then (E-step, M-step, stop when the mean log-likelihood moved < tol = 1e-3,
    at most 100 iterations), covariance floor 1e-6 -- and ONE deviation: the 2-means starts from the quartiles instead of sklearn's
    randomly seeded k-means++ draw.  Upstream calls ``GaussianMixture(n_components=2)`` without a random state
    (pointpdf_v1m1_base.py:334-336), so its own fit moves with numpy's global generator; on the reference's fixture trees this fit lands
    inside that spread (cut 0.52871 vs 0.52857 ... 0.52877 over five sklearn seeds; rounds 1-5 ran EM to 1e-6, which ends at ANOTHER cut,
    0.570: sklearn's loose tolerance stops ~5 steps after the k-means start).  -> (means (2,), variances (2,), weights (2,))"""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if x.size < 2 or np.ptp(x) == 0.0:
        return np.array([x.mean() if x.size else 0.0] * 2), np.array([reg, reg]), np.array([0.5, 0.5])
    q1, q3 = np.quantile(x, [0.25, 0.75])
    mu = np.array([q1, q3]) if q3 > q1 else np.array([x.min(), x.max()])
    for _ in range(300):                                         # 2-means from the quartiles, to its fixed point
        a = np.abs(x[:, None] - mu[None, :]).argmin(1)
        if a.min() == a.max():
            break
        new = np.array([x[a == 0].mean(), x[a == 1].mean()])
        fixed = np.array_equal(new, mu)
        mu = new
        if fixed:
            break
    a = np.abs(x[:, None] - mu[None, :]).argmin(1)
    r = np.stack([a == 0, a == 1], 1).astype(np.float64)

    def m_step(r):
        nk = r.sum(0) + 1e-300
        mu = (r * x[:, None]).sum(0) / nk
        return nk / x.size, mu, (r * (x[:, None] - mu[None, :]) ** 2).sum(0) / nk + reg

    pi, mu, var = m_step(r)
    prev = -np.inf
    for _ in range(iters):
        logp = -0.5 * (np.log(2 * np.pi * var)[None, :] + (x[:, None] - mu[None, :]) ** 2 / var[None, :]) + np.log(pi)[None, :]
        mx = logp.max(1, keepdims=True)
        lse = mx[:, 0] + np.log(np.exp(logp - mx).sum(1))
        pi, mu, var = m_step(np.exp(logp - lse[:, None]))
        ll = lse.mean()
        if abs(ll - prev) < tol:
            break
        prev = ll
    return mu, var, pi


def _prune_by_spanning_tree_device(coord, msp, neighbors, node):
    """``_prune_by_spanning_tree`` (pointpdf_v1m1_base.py:309-380) without leaving the device for the graph work: Boruvka spanning
    forest over the neighbour similarities inside the region, the weak-edge threshold from the mixture of the tree weights (one
    small device -> host copy), connected components by label propagation, component sizes by a device unique."""
    n = coord.shape[0]
    node_nn = neighbors[node]
    sim = _pair_similarity(node, node_nn, coord, msp)
    keep = (node_nn != -1) & torch.isin(node_nn, node) & (node_nn != node[:, None])
    eu = node[:, None].expand_as(node_nn)[keep]
    ev = node_nn[keep]
    ew = sim[keep]
    # upstream builds a scipy csr_matrix from these triplets, which SUMS repeated (row, col) entries -- and `node` does hold
    # repeats when the region is still the seed list (seeds are drawn with replacement, :206): same arithmetic here
    key, inv = torch.unique(eu * n + ev, return_inverse=True)
    ew = torch.zeros(key.shape[0], dtype=ew.dtype, device=ew.device).scatter_add_(0, inv, ew)
    eu, ev = torch.div(key, n, rounding_mode="floor"), key % n
    tree = minimum_spanning_forest(n, eu, ev, ew)
    w = ew[tree]
    means, var, _ = gmm2_1d(w.cpu().numpy())
    top = int(np.argmax(means))
    lower = means[top] - 2.0 * var[top]                          # the "std" of upstream's z-score filter is the covariance, as there
    weak = w.double() < lower                                    # upstream keeps exactly the left outliers of the tree
    lab = connected_labels(n, eu[tree][weak], ev[tree][weak])
    touched = torch.unique(torch.cat([node, node_nn.reshape(-1)]))[1:]   # ([1:]: upstream drops the first entry, the -1 padding)
    labels, sizes = torch.unique(lab[touched], return_counts=True)
    sz = sizes.double()
    big = (sz - sz.mean()) / sz.std(unbiased=False) > 2.0        # z_score_mask_np(area="right", score=2.0); numpy's std is the population one
    return torch.isin(lab, labels[big])


def _prune_by_spanning_tree_hip(coord, msp, neighbors, node, distinct=False):
    """``_prune_by_spanning_tree_device`` with the three graph steps as single-workgroup HIP kernels (csrc/graph_prune.hip): the spanning
    forest (``pdf_graph_forest``: the same forest as ``minimum_spanning_forest`` -- under the strict order (weight, entry) it is unique),
    the mixture fit (``pdf_gmm2_1d``: ``gmm2_1d``'s EM in double, on the device -- the threshold never visits the host) and the
    connected components of the weak tree edges (``pdf_graph_forest`` without weights).  One host read remains (the tree's size).
    ``distinct``: ``node`` is sorted and free of repeats (a region that grew at least once)."""
    be = _native.backend_for(coord)
    n = coord.shape[0]
    node_nn = neighbors[node]
    sim = _pair_similarity(node, node_nn, coord, msp)
    member = torch.zeros(n + 1, dtype=torch.bool, device=coord.device)     # (slot n: the -1 padding)
    member[node] = True
    keep = member[node_nn] & (node_nn != -1) & (node_nn != node[:, None])
    eu = node[:, None].expand_as(node_nn)[keep]
    ev = node_nn[keep]
    ew = sim[keep]
    if not distinct:
        # repeated (row, col) entries are SUMMED, as scipy's csr_matrix does -- `node` holds repeats only while the region is still the
        # seed list (seeds are drawn with replacement, :206).  A grown region is a sorted `unique` and a neighbour row lists ids in
        # ascending order: its entries are distinct and already in this (row, col) order
        key, inv = torch.unique(eu * n + ev, return_inverse=True)
        ew = torch.zeros(key.shape[0], dtype=ew.dtype, device=ew.device).scatter_add_(0, inv, ew)
        eu, ev = torch.div(key, n, rounding_mode="floor"), key % n
    if eu.numel() == 0:
        return torch.zeros(n, dtype=torch.bool, device=coord.device)
    chosen, _ = be.graph_forest(n, eu, ev, node, weight=ew)
    tree = torch.nonzero(chosen).flatten()
    w, tu, tv = ew[tree], eu[tree], ev[tree]
    fit = be.gmm2_1d(w)                                                    # means (2), variances (2), ... on the device
    top = (fit[1] > fit[0]).long()                                         # np.argmax(means)
    lower = fit[top] - 2.0 * fit[2 + top]                                  # the "std" of upstream's z-score filter is the covariance, as there
    weak = w.double() < lower
    _, lab = be.graph_forest(n, tu, tv, node, active=weak, want_chosen=False)
    lab = lab.long()
    touched = torch.unique(torch.cat([node, node_nn.reshape(-1)]))[1:]     # ([1:]: upstream drops the first entry, the -1 padding)
    labels, sizes = torch.unique(lab[touched], return_counts=True)
    sz = sizes.double()
    big = (sz - sz.mean()) / sz.std(unbiased=False) > 2.0
    return torch.isin(lab, labels[big])


@torch.no_grad()
def pseudo_labeling(coord, logits, neighbors, condition_from="msp", beta=1.5, seed_from="ml", seed_range=0.15, num_seed=100,
                    slide_window=True, generator=None, prune="auto"):
    """One scene (local neighbour ids) -> bool mask (n,) on the host, like upstream's static method (:187-382).  The seed draw uses
    a CPU generator (upstream: the global one), the GMM numpy's global state."""
    if coord.is_cuda and prune in ("auto", "hip") and os.environ.get("PDFOPS_PL_STATIC", "1") != "0":
        # device tensors: the sync-free form (all growth rounds in one kernel, sizes kept on the device); PDFOPS_PL_STATIC=0: rounds 1-4's
        return get_pseudo_mask_static(coord, logits, [coord.shape[0]], neighbors, condition_from=condition_from, beta=beta, seed_from=seed_from,
                                      seed_range=seed_range, num_seed=num_seed, slide_window=slide_window, generator=generator).cpu()
    msp = torch.softmax(logits, dim=-1).max(dim=-1)[0]
    ml = logits.max(dim=-1)[0]
    ml = (ml - ml.min()) / (ml.max() - ml.min() + 1e-6)
    score = msp if condition_from == "msp" else ml
    stop = torch.mean(score) - beta * torch.std(score)
    src = msp if seed_from == "msp" else ml
    dice = torch.randint(0, int(seed_range * len(src)), [num_seed], generator=generator)
    seeds = torch.sort(src, dim=-1)[1][dice.to(src.device)]
    grow = _grow_region_reference if os.environ.get("PDFOPS_PL_GROW") == "reference" else _grow_region   # (A/B knob)
    region, grew = grow(coord, score, neighbors, seeds, stop, slide_window, with_flag=True)
    # prune: "host" = scipy / sklearn exactly as upstream (CPU tensors: the reference-parity path); "device" = the same stage as
    # torch graph ops + a deterministic mixture fit (any device); "hip" = that stage as HIP kernels (csrc/graph_prune.hip);
    # "auto" = hip for device tensors
    if prune == "hip" or (prune == "auto" and coord.is_cuda):
        return _prune_by_spanning_tree_hip(coord, msp, neighbors, region, distinct=grew).cpu()
    if prune == "device":
        return _prune_by_spanning_tree_device(coord, msp, neighbors, region).cpu()
    return _prune_by_spanning_tree(coord, msp, neighbors, region)


# ------------------------------------------------------------------------------------------------------------------
# The whole pass without a host read (round 5): region growing as ONE kernel per batch (csrc/region_grow.hip: all rounds on the device),
# the region's graph entries by a second kernel, spanning forest / mixture fit / components with their sizes read from device memory
# (pdf_graph_forest_dev / pdf_gmm2_1d_dev), everything else as fixed-shape torch ops.  Nothing waits for the device, so the step around
# it keeps running ahead (rounds 1-4: ~10 host reads per scene, the device idle for the ~7 ms the host needed per 150k-point scene) and
# the pass can be captured into the step's graph.
# ------------------------------------------------------------------------------------------------------------------
def _scene_ranges(dev, starts, sizes, offset=None):
    """(starts, sizes) of the scenes as int32 device tensors, owned by the CALLER's step (round 6; rounds 1-5 kept them in a
    process-wide cache whose eviction freed tensors that captured graphs still pointed at).
    With the batch's device ``offset`` (scene ends) they are derived from it by two elementwise kernels: graph-safe, allocated from the
    capturing graph's own pool, hence alive exactly as long as the graph.  Without it (eager callers that only hold host sizes) they
    travel through a fresh pinned buffer and a non-blocking copy, as ``_draw_seeds`` does: nothing waits for the device.  A capture
    without ``offset`` is refused: a host buffer cannot be read by a replay."""
    if offset is not None and offset.device == dev and offset.numel() == len(sizes):
        ends = offset.detach().to(torch.int32)
        sizes_d = ends.clone()
        sizes_d[1:] -= ends[:-1]
        return ends - sizes_d, sizes_d
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("pseudo-label pass: a captured step needs the batch's device `offset` tensor for its scene ranges")
    host = torch.empty((2, len(sizes)), dtype=torch.int32, pin_memory=dev.type == "cuda")
    host[0] = torch.tensor(starts, dtype=torch.int32)
    host[1] = torch.tensor(sizes, dtype=torch.int32)
    r = host.to(dev, non_blocking=True)
    return r[0], r[1]


def _draw_seeds(num_seed, hi, generator, dev):
    """``torch.randint(0, hi, [num_seed])`` of upstream's seed draw (:206) on the device without waiting for it.
    ONE source per configuration, the same whether the step runs eagerly or is replayed (round 6; rounds 1-5 drew eager steps from the
    CPU generator and captured ones from the device generator, so a trainer that mixes replayed and eager batches used two unrelated
    streams and a capture's warm-up advanced the CPU generator):
    * no explicit generator, device tensors -> the DEVICE generator (graph-safe Philox offsets; `torch.cuda.manual_seed` seeds it).
      Deviation from upstream, which draws from the global CPU generator: same distribution, different stream.
    * an explicit (CPU) generator -> upstream's CPU draw through a FRESH pinned buffer per call (a pageable host -> device copy would
      wait for the stream; the pinned allocator recycles a buffer only after its copy ran).  Not capturable: a replayed graph cannot
      re-read a host buffer the host has refilled for a later step."""
    if generator is None and dev.type == "cuda":
        return torch.randint(0, hi, [num_seed], device=dev)
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("pseudo-label pass: a captured step draws its seeds with the device generator (no explicit CPU generator)")
    dice = torch.empty((num_seed,), dtype=torch.int64, pin_memory=dev.type == "cuda")
    dice.copy_(torch.randint(0, hi, [num_seed], generator=generator))
    return dice.to(dev, non_blocking=True)


@torch.no_grad()
def get_pseudo_mask_static(coord, seg_logits, offset_host, neighbors, condition_from="msp", beta=1.5, seed_from="ml", seed_range=0.15,
                           num_seed=100, slide_window=True, generator=None, max_rounds=4096, info=None, offset=None):
    """pointpdf_v1m1_base.py:118-382 for a batch on the device, free of host reads.  ``neighbors``: (N, k) GLOBAL ids, -1 padded: the int32
    table of ``radius_neighbors(..., raw=True)`` as it is, or the int64 one (converted); ``offset_host``: the scenes' end positions as
    Python ints; ``offset``: the same ends as the batch's device tensor (required while a step is being captured: the scene ranges
    the kernels read are derived from it inside the graph, ``_scene_ranges``).  -> bool (N,) on the device.
    Torch does the row-wise work on the logits (softmax, row maxima) and draws the seed ranks; every reduction over a scene, the seed
    lookup, the growth, the region's graph, both labellings, the mixture fit and the component statistics are kernels of
    csrc/region_grow.hip / csrc/graph_prune.hip, one launch for all scenes wherever the stage allows.
    ``info`` (dict, optional): receives the device tensors with the per-scene round counts / region sizes (diagnostics: reading them syncs)."""
    be = _native.backend_for(coord)
    dev = coord.device
    N, ns = neighbors.shape
    ends = [int(v) for v in offset_host]
    starts = [0] + ends[:-1]
    sizes = [e - s for s, e in zip(starts, ends)]
    B = len(ends)
    if B == 0 or N == 0:
        return torch.zeros(N, dtype=torch.bool, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    u8 = dict(dtype=torch.uint8, device=dev)
    starts_d, sizes_d = _scene_ranges(dev, starts, sizes, offset)
    nn = neighbors if neighbors.dtype == torch.int32 else neighbors.int()
    nn = nn.contiguous()
    coord = coord.contiguous()
    logits = seg_logits.float()
    msp = torch.softmax(logits, dim=-1).max(dim=-1)[0].contiguous()
    ml_raw = logits.max(dim=-1)[0].contiguous()
    ml = torch.empty(N, **f32)
    stop = torch.empty(B, **f32)
    mult = torch.empty(N, **i32)
    be._call("region_stats", B, starts_d, sizes_d, msp, ml_raw, int(condition_from != "msp"), float(beta), ml, stop, mult)
    score = msp if condition_from == "msp" else ml
    dice = torch.stack([_draw_seeds(num_seed, int(seed_range * n), generator, dev) for n in sizes]).contiguous()
    be._call("region_seeds", B, starts_d, sizes_d, msp if seed_from == "msp" else ml, dice, int(num_seed), mult)
    lists = torch.empty(2 * N, **i32)
    simbuf = torch.empty(N, **f32)
    ginfo = torch.empty((B, 4), **i32)
    # (PDFOPS_GROW_SCAN=1: the growth kernel's first form -- every stage a pass over all points -- which scenes beyond ~600k points fall
    #  back to; the size handed down only has to exceed what the LDS form can hold.  A/B and test knob.)
    scan = os.environ.get("PDFOPS_GROW_SCAN") == "1"
    be._call("region_grow", B, starts_d, sizes_d, (1 << 30) if scan else max(sizes), coord, score, nn, ns, stop, int(bool(slide_window)),
             int(max_rounds), mult, lists, simbuf, ginfo)
    nodes = torch.empty(N, dtype=torch.int64, device=dev)
    eu = torch.empty(N * ns, dtype=torch.int64, device=dev)
    ev = torch.empty(N * ns, dtype=torch.int64, device=dev)
    ew = torch.empty(N * ns, **f32)
    touched = torch.empty(N, **u8)
    comp = torch.empty(N, **i32)
    lab = torch.empty(N, **i32)
    counts = torch.empty((B, 4), **i32)
    rows_ws = torch.empty(3 * N, **i32)
    listed = not scan and max(sizes) <= int(be.lib.pdf_region_grow_list_points())   # (the growth left the ascending member list behind)
    be._call("region_edges", B, starts_d, sizes_d, coord, msp, nn, ns, mult, lists if listed else None, ginfo if listed else None, nodes, eu, ev, ew,
             touched, comp, lab, counts, rows_ws, N)
    chosen = torch.empty(N * ns, **u8)
    starts_h, sizes_h = (ctypes.c_int * B)(*starts), (ctypes.c_int * B)(*sizes)
    ws1_bytes = sum((int(be.lib.pdf_graph_forest_workspace_bytes(n, n * ns, n)) + 7) & ~7 for n in sizes)
    ws = torch.empty((ws1_bytes // 8 + 1,), dtype=torch.int64, device=dev)
    # the spanning forest of every scene (one workgroup each, all scenes in one launch)
    be._call("graph_forest_batch_dev", B, starts_h, sizes_h, ns, eu, ev, ew, None, nodes, counts, 4, comp, chosen, ws, ws.numel() * 8)
    tu = torch.empty(N, dtype=torch.int64, device=dev)
    tv = torch.empty(N, dtype=torch.int64, device=dev)
    tw = torch.empty(N, **f32)
    tdev = torch.empty((B, 2), **i32)
    be._call("region_tree", B, starts_d, sizes_d, ns, counts, chosen, eu, ev, ew, tu, tv, tw, tdev)
    xs = torch.empty(N, **f32)
    be._call("sort_floats_dev", B, starts_d, sizes_d, tdev, tw, xs, simbuf.view(torch.int32))
    resp = torch.empty((2 * N,), dtype=torch.float64, device=dev)
    fit = torch.empty((B, 8), dtype=torch.float64, device=dev)
    weak = torch.empty(N, **u8)
    be._call("gmm2_weak_dev", B, starts_d, sizes_d, tdev, xs, tw, resp, fit, weak, 100, 1e-3, 1e-6)   # (sklearn's defaults: gmm2_1d)
    # the components the tree falls into without its weak edges
    be._call("graph_forest_batch_dev", B, starts_h, sizes_h, 1, tu, tv, None, weak, nodes, tdev, 2, lab, None, ws, ws.numel() * 8)
    mask = torch.empty(N, **u8)
    be._call("region_mask", B, starts_d, sizes_d, lab, touched, counts, comp, mask)   # (comp: free again -- the counts' workspace)
    if info is not None:
        info.update(grow=ginfo, counts=counts, tree=tdev, stop=stop, fit=fit, tree_weights=tw)
    return mask.view(torch.bool)


@torch.no_grad()
def get_pseudo_mask(coord, seg_logits, offset, radius=0.1, max_neighbor=64, neighbors=None, offset_host=None, generator=None, workers=None, **kw):
    """pointpdf_v1m1_base.py:118-185 for a batch: neighbour table once, scenes one by one; -> bool (N,) on coord's device."""
    if workers is None:
        # device tensors: the scenes one after the other (the HIP pruning stage leaves ~3 ms of short launches per scene: worker threads
        # only fight over the interpreter lock, 46.4 vs 42.7 ms per step with 4 vs 1); host tensors: upstream's 4 workers (scipy / sklearn)
        workers = int(os.environ.get("PDFOPS_PL_WORKERS", "1" if coord.is_cuda else "4"))
    static = coord.is_cuda and os.environ.get("PDFOPS_PL_STATIC", "1") != "0" and kw.get("prune", "auto") in ("auto", "hip")
    static = static and (offset_host is not None or torch.cuda.is_current_stream_capturing() is False)
    if neighbors is None:
        neighbors = radius_neighbors(coord, offset, radius, max_neighbor, raw=static)
    elif not static and neighbors.dtype != torch.int64:
        neighbors = neighbors.long()
    if static:
        # the sync-free form (get_pseudo_mask_static): scene ends from the host copy when the caller has one (one read of `offset` else)
        ends = offset_host if offset_host is not None else [int(v) for v in offset.tolist()]
        kw2 = {k: v for k, v in kw.items() if k != "prune"}
        return get_pseudo_mask_static(coord, seg_logits, ends, neighbors, generator=generator, offset=offset, **kw2)
    ends = offset_host if offset_host is not None else [int(v) for v in offset.tolist()]
    starts = [0] + ends[:-1]
    stream = torch.cuda.current_stream() if coord.is_cuda else None

    def one(se):
        s0, e = se
        nn = neighbors[s0:e]
        nn = torch.where(nn != -1, nn - s0, nn)
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
    def fn(coord, seg_logits, offset, offset_host=None, geometry=None):
        # the neighbour table from the batch's coordinate pre-pass when it made one (geometry.Geometry.radius: `prepass_plan` below)
        table = geometry.radius_cached(radius, max_neighbor) if geometry is not None and hasattr(geometry, "radius_cached") else None
        if table is not None and table.shape[0] != coord.shape[0]:
            table = None
        return get_pseudo_mask(coord, seg_logits.detach(), offset, radius=radius, max_neighbor=max_neighbor, neighbors=table,
                               offset_host=offset_host, **kw)

    fn.accepts_geometry = True      # (PointPdfV1 hands over input_dict["pdf_geometry"])
    fn.prepass_plan = dict(radius=(float(radius), int(max_neighbor)))   # -> engine.GroupedGeometryLoader(..., **fn.prepass_plan) / Geometry.precompute
    fn.accepts_offset_host = True   # (PointPdfV1 hands the host copy of the scene ends over when the batch carries one: no read of `offset`)
    # device tensors take the sync-free form: the pass can be recorded into the step's graph (engine.CapturedStep captures ONE graph then)
    fn.capturable = os.environ.get("PDFOPS_PL_STATIC", "1") != "0" and kw.get("prune", "auto") in ("auto", "hip") and kw.get("generator") is None

    if os.environ.get("PDFOPS_PL_TRACE"):   # diagnostics: host wall time of the wait for the forward and of the pass, per call
        import atexit
        import time
        log = []

        def traced(coord, seg_logits, offset, offset_host=None, geometry=None):
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mask = fn(coord, seg_logits, offset, offset_host=offset_host, geometry=geometry)
            torch.cuda.synchronize()
            st = torch.cuda.memory_stats()
            log.append((t1 - t0, time.perf_counter() - t1, st.get("num_device_alloc", 0), st.get("num_alloc_retries", 0)))
            return mask

        def report():
            tail = log[len(log) // 2:]
            if tail:
                passes = sorted(x[1] for x in tail)
                print(f"[PDFOPS_PL_TRACE] {len(log)} calls; second half: wait for the forward {1e3 * sum(x[0] for x in tail) / len(tail):.2f} ms, "
                      f"pass {1e3 * sum(passes) / len(tail):.2f} ms per call (min {1e3 * passes[0]:.2f}, median {1e3 * passes[len(passes) // 2]:.2f}, "
                      f"max {1e3 * passes[-1]:.2f}); device allocations in that half: {tail[-1][2] - tail[0][2]}, allocator retries: "
                      f"{tail[-1][3] - tail[0][3]}", file=__import__("sys").stderr)
        atexit.register(report)
        traced.accepts_offset_host = True
        return traced
    return fn
