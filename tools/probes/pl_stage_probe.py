"""Where the pseudo-label pass of config 4 spends its time: the (coord, seg_logits, offset) that bench.py's ScanNet-shaped step hands
``pseudo_mask_fn`` (random-init weights, 2 x 150k points), then every stage of the pass timed with a device sync on both sides:
radius table, seed draw, region growing (rounds counted), pruning.  usage: PYTHONPATH=. python tools/probes/pl_stage_probe.py"""
import time
import numpy as np
import torch
from pointcloudpdf_amd import engine, pseudo_label, synthetic

dev = torch.device("cuda", 0)
seen = {}


def tap(coord, seg_logits, offset):
    seen["args"] = (coord.detach().clone(), seg_logits.detach().clone(), offset.clone())
    return torch.zeros(coord.shape[0], dtype=torch.bool, device=coord.device)


step = engine.OpenSegStep(in_channels=9, num_classes=20, loss_weight=0.04, pseudo_mask_fn=tap).to(dev)
synthetic.fill_parameters_deterministic(step, seed=1)
step.train()
b = synthetic.make_batch([150000, 150000], first_scene_id=0, device=dev, kind="scannet", unknown=(4, 7, 14, 16))
step({k: b[k] for k in ("coord", "feat", "offset", "offset_host", "segment")})
coord, logits, offset = seen["args"]
print("logits", tuple(logits.shape), "offset", offset.tolist())


def timed(fn, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(*a, **k)
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


for rep in range(3):
    nn, t_nn = timed(pseudo_label.radius_neighbors, coord, offset, 0.1, 64)
    _, t_all = timed(pseudo_label.get_pseudo_mask, coord, logits, offset, radius=0.1, max_neighbor=64, condition_from="msp", beta=1.5,
                     seed_from="ml", seed_range=0.15, num_seed=100, slide_window=True)
    print(f"rep {rep}: radius table {t_nn:.2f} ms, whole pass (4 worker threads) {t_all:.2f} ms")

# one scene, stage by stage
s0, e = 0, int(offset[0])
c, lg = coord[s0:e], logits[s0:e]
local = nn[s0:e].clone()
local[local != -1] -= s0
for rep in range(2):
    def seeds_fn():
        msp = torch.softmax(lg, dim=-1).max(dim=-1)[0]
        ml = lg.max(dim=-1)[0]
        ml = (ml - ml.min()) / (ml.max() - ml.min() + 1e-6)
        stop = torch.mean(msp) - 1.5 * torch.std(msp)
        dice = torch.randint(0, int(0.15 * len(ml)), [100], generator=torch.Generator().manual_seed(rep))
        return msp, stop, torch.sort(ml, dim=-1)[1][dice.to(ml.device)]
    (msp, stop, seeds), t_seed = timed(seeds_fn)
    rounds = [0]
    orig_unique = torch.nonzero

    def counting_unique(*a, **k):
        rounds[0] += 1
        return orig_unique(*a, **k)
    torch.nonzero = counting_unique
    region, t_grow = timed(pseudo_label._grow_region, c, msp, local, seeds, stop, True)
    torch.nonzero = orig_unique
    mask, t_prune = timed(pseudo_label._prune_by_spanning_tree_device, c, msp, local, region)
    print(f"scene 0 rep {rep}: seeds {t_seed:.2f} ms, growing {t_grow:.2f} ms ({rounds[0] // 2} rounds, region {region.numel()}), "
          f"pruning {t_prune:.2f} ms (mask {int(mask.sum())})")

# the pruning stage piece by piece (scene 0, the last region)
n = c.shape[0]
for rep in range(2):
    def edges():
        node_nn = local[region]
        sim = pseudo_label._pair_similarity(region, node_nn, c, msp)
        keep = (node_nn != -1) & torch.isin(node_nn, region) & (node_nn != region[:, None])
        eu = region[:, None].expand_as(node_nn)[keep]
        ev = node_nn[keep]
        ew = sim[keep]
        key, inv = torch.unique(eu * n + ev, return_inverse=True)
        ew = torch.zeros(key.shape[0], dtype=ew.dtype, device=ew.device).scatter_add_(0, inv, ew)
        return torch.div(key, n, rounding_mode="floor"), key % n, ew, node_nn
    (eu, ev, ew, node_nn), t_edges = timed(edges)
    tree, t_mst = timed(pseudo_label.minimum_spanning_forest, n, eu, ev, ew)
    w = ew[tree]
    wh, t_copy = timed(lambda: w.cpu().numpy())
    t0 = time.perf_counter()
    means, var, _ = pseudo_label.gmm2_1d(wh)
    t_gmm = (time.perf_counter() - t0) * 1e3
    top = int(np.argmax(means))
    weak = w.double() < (means[top] - 2.0 * var[top])
    lab, t_cc = timed(pseudo_label.connected_labels, n, eu[tree][weak], ev[tree][weak])

    def tail():
        touched = torch.unique(torch.cat([region, node_nn.reshape(-1)]))[1:]
        labels, sizes = torch.unique(lab[touched], return_counts=True)
        sz = sizes.double()
        big = (sz - sz.mean()) / sz.std(unbiased=False) > 2.0
        return torch.isin(lab, labels[big]).cpu()
    _, t_tail = timed(tail)
    print(f"prune rep {rep}: edges {eu.numel()} in {t_edges:.2f} ms, spanning forest ({tree.numel()} edges) {t_mst:.2f} ms, copy {t_copy:.2f} ms, "
          f"mixture fit {t_gmm:.2f} ms, components ({int(weak.sum())} weak edges) {t_cc:.2f} ms, sizes + mask {t_tail:.2f} ms")

# the same stage as HIP kernels (csrc/graph_prune.hip)
from pointcloudpdf_amd import _native
be = _native.hip_backend()
for rep in range(3):
    _, t_hip = timed(pseudo_label._prune_by_spanning_tree_hip, c, msp, local, region)
    (chosen, _), t_f = timed(be.graph_forest, n, eu, ev, region, weight=ew)
    tr, t_nz = timed(lambda: torch.nonzero(chosen).flatten())
    fit, t_g = timed(be.gmm2_1d, ew[tr])
    wk = ew[tr].double() < (fit[1] - 2 * fit[3])
    _, t_c = timed(be.graph_forest, n, eu[tr], ev[tr], region, active=wk, want_chosen=False)
    print(f"hip prune rep {rep}: whole {t_hip:.2f} ms; forest kernel {t_f:.2f} ms, nonzero {t_nz:.2f} ms, sort + mixture kernel {t_g:.2f} ms "
          f"({int(fit[6])} iterations), components kernel {t_c:.2f} ms")
for rep in range(2):
    _, t1 = timed(pseudo_label.pseudo_labeling, c, lg, local, condition_from="msp", beta=1.5, seed_from="ml", seed_range=0.15, num_seed=100, slide_window=True)
    print(f"pseudo_labeling, one scene, one thread: {t1:.2f} ms")
