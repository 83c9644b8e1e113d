"""Does one whole training step (forward + backward + SGD) replay as ONE captured hipGraph on this stack, and what does a replay cost?
Round 1 measured 187 ms per replay against 68 ms eager (~3,000 launches then); the step is host-bound now (1,235 launches, host enqueue
~ step time), so the question is worth asking again.  Static inputs: batch tensors + a static-address Geometry (Geometry.load).

    python tools/graph_probe.py [--points 100000] [--steps 12]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    from pointcloudpdf_amd import engine, synthetic
    from pointcloudpdf_amd.geometry import Geometry

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=0.0, momentum=0.9, weight_decay=0.0)   # lr 0: eager and replayed losses comparable
    batches = [synthetic.make_batch([a.points] * 2, first_scene_id=10 * i, device=dev) for i in range(3)]
    geoms = [Geometry(b["coord"], b["offset"], b["offset_host"]).precompute() for b in batches]
    keys = ["coord", "feat", "offset", "segment"]
    static = {k: batches[0][k].clone() for k in keys}
    static["offset_host"] = batches[0]["offset_host"]
    sgeom = Geometry(static["coord"], static["offset"], static["offset_host"]).precompute()

    def run(data, geom):
        opt.zero_grad(set_to_none=True)
        out = step(dict(data, pdf_geometry=geom))
        out["loss"].backward()
        opt.step()
        return out

    def load(i):
        b = batches[i % 3]
        for k in keys:
            static[k].copy_(b[k])
        sgeom.load(geoms[i % 3])

    res = {}
    log = lambda *a: print(*a, file=sys.stderr, flush=True)
    for _ in range(3):
        run(batches[0], geoms[0])
    torch.cuda.synchronize()
    eager_loss = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = run(batches[i % 3], geoms[i % 3])
        eager_loss.append(out["loss"].detach())
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    res["eager_ms_per_step"] = (time.perf_counter() - t0) / a.steps * 1e3
    res["eager_host_ms_per_step"] = t_host / a.steps * 1e3
    eager_loss = [float(x) for x in eager_loss]
    del out   # (a live loss keeps the autograd graph and its AccumulateGrad nodes, created on the default stream)
    engine.release_autograd_state(step)

    log("eager done", res)
    # warm-up on a side stream with the static tensors, then capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            run(static, sgeom)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    engine.release_autograd_state(step)
    log("side-stream warm-up done")
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    try:
        with torch.cuda.graph(g, capture_error_mode=os.environ.get("CAPTURE_MODE", "global")):
            stages = os.environ.get("STAGES", "fbo")
            out = step(dict(static, pdf_geometry=sgeom))
            log("forward captured")
            if "b" in stages:
                with torch.autograd.set_multithreading_enabled(os.environ.get("BWD_THREADS", "1") == "1"):
                    out["loss"].backward()
                log("backward captured")
            if "o" in stages:
                opt.step()
                log("optimizer captured")
    except Exception as e:   # noqa: BLE001
        import traceback
        traceback.print_exc()
        res["capture_error"] = repr(e)[:600]
        print(json.dumps(res))
        return
    torch.cuda.synchronize()
    res["capture_s"] = time.perf_counter() - t0
    log("capture done", res)
    graph_loss = []
    for i in range(3):
        load(i)
        g.replay()
        graph_loss.append(float(out["loss"]))
    res["loss_eager_first3"] = eager_loss[:3]
    res["loss_graph_first3"] = graph_loss
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        load(i)
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    res["graph_ms_per_step"] = (time.perf_counter() - t0) / a.steps * 1e3
    res["graph_host_ms_per_step"] = t_host / a.steps * 1e3
    t0 = time.perf_counter()
    for i in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    res["graph_replay_only_ms_per_step"] = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
