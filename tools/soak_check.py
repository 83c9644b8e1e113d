"""Long-run agreement of the execution modes of engine.TrainStep on one box: K optimisation steps replayed back to back (the bench's
schedule), replayed with a device synchronisation after every step, and issued eagerly -- same batches, same initial parameters.  The three
must end in bit-identical parameters (every kernel on the path sums in a fixed order).  Prints one line per configuration; exit code 1 on
any difference.  Usage on the GPU box: python tools/soak_check.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointcloudpdf_amd import engine, pseudo_label, synthetic

dev = torch.device("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 120
keys = ("coord", "feat", "offset", "offset_host", "segment")


def run(cfg, mode):
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    kw, bkw = {}, {}
    if cfg["scannet"]:
        kw, bkw = dict(in_channels=9, num_classes=20, loss_weight=0.04), dict(kind="scannet", unknown=(4, 7, 14, 16))
        if cfg["pass"]:
            kw["pseudo_mask_fn"] = pseudo_label.make_pseudo_mask_fn(radius=0.1, max_neighbor=64, condition_from="msp", beta=1.5, seed_from="ml",
                                                                    seed_range=0.15, num_seed=100, slide_window=True)
    step = engine.OpenSegStep(**kw).to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    train = engine.TrainStep(step, opt, graph=mode != "eager")
    pool = [synthetic.make_batch([cfg["n"]] * 2, first_scene_id=10 * i, device=dev, **bkw) for i in range(3)]
    it = iter(engine.GroupedGeometryLoader(({k: pool[j % 3][k] for k in keys} for j in range(K)), group=cfg.get("group", 8)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        out = train(next(it))
        if mode != "queued" or i % 40 == 39:   # (queued: a synchronisation every 40 steps, as between a bench's warm-up and timed region)
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(out["loss"])
    params = [p.detach().clone() for p in step.parameters()]
    engine.release_autograd_state(step)
    del train, opt, step
    torch.cuda.empty_cache()
    return loss, params, dt


bad = 0
for cfg in (dict(name="S3DIS 2 x 100k", n=100000, scannet=False, **{"pass": False}),
            dict(name="S3DIS 2 x 131,200 (level 5 = 2 x 512 rows)", n=131200, scannet=False, **{"pass": False}),
            dict(name="ScanNet 2 x 150k + pseudo-label pass", n=150000, scannet=True, **{"pass": True})):
    res = {m: run(cfg, m) for m in (("queued", "synchronised") if cfg["pass"] else ("queued", "synchronised", "eager"))}
    ref = res["synchronised"]
    line = [f"{cfg['name']}, {K} steps:"]
    for m, (loss, params, dt) in res.items():
        same = all(torch.equal(a, b) for a, b in zip(params, ref[1]))
        bad += 0 if same else 1
        line.append(f"{m} loss {loss:.7f} {1e3 * dt / K:.1f} ms/step {'== synchronised' if same else 'DIFFERS'}")
    print("  ".join(line), flush=True)
sys.exit(1 if bad else 0)
