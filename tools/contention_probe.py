"""What the pre-pass costs the training stream, by kind of pre-pass work (round 4).

A replayed training step (geometry precomputed, nothing else on the device) is timed alone and next to a side stream that keeps launching
(a) the farthest-point chain of a 12-batch group (24 workgroups of 1,024 threads, ~20 ms per launch: few CUs, long), (b) the kNN tables
of the group (whole-chip kernels of 0.2-0.7 ms).  PDFOPS_PROBE_SIDE_CUS=32 / PDFOPS_PROBE_MAIN_CUS=224: the side work / the training stream
on CU-masked streams (disjoint compute units).

    python tools/contention_probe.py [--steps 30]
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointcloudpdf_amd import _native, engine, synthetic
from pointcloudpdf_amd.geometry import Geometry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--group", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    be = _native.hip_backend()
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    batch = synthetic.make_batch([100000, 100000], first_scene_id=0, device=dev)
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
    # main stream: never the legacy default stream (a CU-masked stream is a blocking stream: it would serialise with it)
    main_cus = int(os.environ.get("PDFOPS_PROBE_MAIN_CUS", "0"))      # e.g. 224: the training stream on CUs [256 - 224, 256)
    side_cus = int(os.environ.get("PDFOPS_PROBE_SIDE_CUS", "0"))      # e.g. 32: the pre-pass work on CUs [0, 32)
    main = _native.cu_masked_stream(256 - main_cus, main_cus) if main_cus else torch.cuda.Stream()
    train = engine.TrainStep(step, opt, graph=True, stream=main)
    data = dict(batch, pdf_geometry=geom)
    for _ in range(3):
        train(data)
    torch.cuda.synchronize()

    # the group's coordinates (12 batches = 24 scenes) for the side work
    group = [synthetic.make_batch([100000, 100000], first_scene_id=10 * (i + 1), device=dev) for i in range(a.group)]
    coord = torch.cat([b["coord"] for b in group])
    o_host, base = [], 0
    for b in group:
        o_host += [base + int(e) for e in b["offset_host"]]
        base = o_host[-1]
    offset = torch.tensor(o_host, dtype=torch.int32, device=dev)
    G = Geometry(coord, offset, o_host)
    lv2, _ = G.down(0, 4)
    L1, L2 = G.levels[0], G.levels[lv2]
    torch.cuda.synchronize()
    ends, tot = [], 0
    for sz in _sizes(o_host):
        tot += sz // 4
        ends.append(tot)
    new_o, new_total = torch.tensor(ends, dtype=torch.int32, device=dev), tot
    ncu = side_cus
    side = _native.cu_masked_stream(0, side_cus) if side_cus else torch.cuda.Stream()

    def fps_once():
        be.farthest_point_sampling(L1.p, L1.o, new_o, max(_sizes(o_host)), new_total)

    def knn_once():
        be.knn_query(8, L1.p, L1.p, L1.o, L1.o)
        be.knn_query(16, L2.p, L2.p, L2.o, L2.o)
        be.knn_query(16, L1.p, L2.p, L1.o, L2.o)
        be.knn_query(3, L2.p, L1.p, L2.o, L1.o)

    def timed(side_fn, label):
        stop = False
        n_side = 0
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # keep the side stream busy: enqueue enough work up front (it runs concurrently with the steps); host pace is not the subject here
        if side_fn is not None:
            with torch.cuda.stream(side):
                for _ in range(side_fn[1]):
                    side_fn[0]()
                    n_side += 1
        t0 = time.perf_counter()
        ev0.record(main)
        for _ in range(a.steps):
            train(data)
        ev1.record(main)
        ev1.synchronize()
        dt = time.perf_counter() - t0
        side_busy = None
        if side_fn is not None:
            se = torch.cuda.Event()
            se.record(side)
            side_busy = not se.query()   # still busy when the steps finished: the side stream covered the whole region
        torch.cuda.synchronize()
        print(json.dumps(dict(case=label, ms_per_step=round(ev0.elapsed_time(ev1) / a.steps, 3), wall_ms_per_step=round(dt / a.steps * 1e3, 3),
                              side_launch_sequences=n_side, side_still_busy_at_end=side_busy, side_cus=side_cus or None, main_cus=main_cus or None)), flush=True)

    est = a.steps * 17.0   # ms of steps to cover
    timed(None, "alone")
    timed((fps_once, int(est / 20) + 2), "beside the farthest-point chain (24 workgroups, ~20 ms per launch)")
    timed((knn_once, int(est / 2.5) + 4), "beside the kNN tables of a 12-batch group (whole-chip kernels)")
    timed(None, "alone again")


def _sizes(o_host):
    prev, out = 0, []
    for e in o_host:
        out.append(e - prev)
        prev = e
    return out


if __name__ == "__main__":
    main()
