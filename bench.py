#!/usr/bin/env python3
"""Headline benchmark: PointTransformer-V1 Seg50 + PDF U-decoder, one training step (fwd + bwd + SGD update) on
S3DIS-shaped synthetic scenes of 100k points, bs = 2 scenes per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch that is already resident in HBM: geometry pre-pass (4 FPS + the
13 distinct kNN tables + interpolation tables), Seg50 forward, U-decoder forward, CE + PDF loss, backward (DDP gradient
all-reduce over RCCL when N > 1), SGD update.  Every step gets a different batch from a small rotating pool and builds
a fresh Geometry -- nothing is cached across steps.  K steps are timed between barrier + synchronize pairs; the MAX over
ranks is reported; value = (points processed by all ranks) / time.

The JSON line also carries
  * "roofline": the dominant kernel's achieved algorithmic HBM GB/s (SURVEY.md 8d byte counts / HIP-event time of that
    kernel measured live on the stream it runs on) against the 8 TB/s HBM3E peak;
  * "cpu_baseline": the CPU oracle (reference algorithms: brute-force kNN, iterative FPS, torch-CPU layers) timed on
    this host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# HBM traffic per launch from rocprofv3 PMC passes (profiles/r01_h_pmc_{fetch,write}_size.txt: one launch = the 24 scenes of a
# 12-batch group; FETCH_SIZE doubled as the guide prescribes for gfx950, KB -> bytes, averaged over the launches of the profiled
# run like `achieved`).  FPS (k_fps_mw<16> / <8> / k_fps<4> by level): 9,498 KB fetch (x2) + 49,096 KB write on average -- the
# tmp-distance stores of the touched buckets.
# Bottleneck backward: sum over the kernels that only run in backward passes (k_b*, k_wg, k_bn_bwd_*, k_colsum: 19.06 GB per step in
# the r01_g PMC passes; a few of their launches belong to the Linear-BN nodes outside the Bottlenecks) / 18 calls per step.
PMC_TRAFFIC_BYTES = {"farthest_point_sampling": (2 * 9498.0 + 49096.0) * 1024, "bottleneck_backward": 19063.8e6 / 18}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--workload", choices=["s3dis", "scannet"], default="s3dis",
                    help="s3dis = BASELINE config 2 / 3 (6 input channels, 13 classes, the headline); scannet = config 4 shape (coord + "
                         "colour + normal = 9 channels, 20 classes, unknown classes {4, 7, 14, 16}, 150k points per scene unless --points)")
    ap.add_argument("--pseudo-label", type=int, default=0,
                    help="1 = run the PDF pseudo-label pass inside the step (config 4; recognizer settings of "
                         "configs/scannet/openseg-pt-v1-0-pointpdf-v1m1-base.py:40-58) instead of the fixed 1-in-7 stand-in mask")
    ap.add_argument("--points", type=int, default=None, help="points per scene (default 100000, scannet: 150000)")
    ap.add_argument("--scenes", type=int, default=2, help="scenes per GPU (batch size per rank)")
    ap.add_argument("--pool", type=int, default=3, help="distinct batches per rank to rotate through")
    ap.add_argument("--jitter", type=float, default=0.0,
                    help="scene sizes drawn in points * (1 +- jitter), seeded per rank and scene (SURVEY 8d config 3, secondary "
                         "number: straggler imbalance across ranks); 0 = every scene exactly --points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops-roofline", action="store_true", help="skip the per-op HBM roofline micro-benchmark (rank 0, N = 1)")
    ap.add_argument("--cpu-points", type=int, default=100000, help="scene size of the bounded CPU-baseline sample")
    ap.add_argument("--amp", action="store_true", help="fp16 autocast around the step (reference enable_amp=True)")
    ap.add_argument("--prefetch", type=int, default=16,
                    help="geometry pre-pass group: the pre-pass of the NEXT `prefetch` batches runs as one launch sequence on a side "
                         "stream while the current group trains (0 = inline, serial)")
    ap.add_argument("--ddp", choices=["flat", "torch"], default="flat",
                    help="gradient exchange for N > 1: one flat-buffer all-reduce after the backward (engine.FlatGradAllReduce) or "
                         "torch DistributedDataParallel (per-parameter bucket copies: +3 ms per step, measured)")
    ap.add_argument("--graph", type=int, default=0, help="replay fwd+bwd+SGD as one captured hipGraph (needs --prefetch > 0)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of individual backend calls on torch's current stream (the stream the kernels are launched on)."""

    def __init__(self, backend, names):
        self.backend, self.names = backend, names
        self.records = {n: [] for n in names}
        self.enabled = False
        self._orig = {}

    def install(self):
        for n in self.names:
            orig = getattr(self.backend, n)
            self._orig[n] = orig

            def wrapped(*a, _orig=orig, _n=n, **k):
                if not self.enabled:
                    return _orig(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _orig(*a, **k)
                e1.record()
                self.records[_n].append((e0, e1, self._bytes(_n, a, out)))
                return out

            setattr(self.backend, n, wrapped)

    @staticmethod
    def _bytes(name, args, out):
        """Algorithmic HBM bytes of one call."""
        if name == "knn_query":  # 12N + 12M + 8B + 8Mk
            k, xyz, new_xyz, offset = args[0], args[1], args[2], args[3]
            return 12 * xyz.shape[0] + 12 * new_xyz.shape[0] + 8 * offset.shape[0] + 8 * new_xyz.shape[0] * k
        if name == "farthest_point_sampling":  # 12N + 4M'
            return 12 * args[0].shape[0] + 4 * out.shape[0]
        if name == "group_forward":  # table once + idx + output (+ xyz / new_xyz / rel-xyz when with_xyz)
            feat, xyz, new_xyz, idx, with_xyz = args
            m, ns = idx.shape
            b = 4 * feat.numel() + 4 * idx.numel() + 4 * m * ns * feat.shape[1]
            if with_xyz:
                b += 12 * xyz.shape[0] + 12 * m + 12 * m * ns
            return b
        if name == "group_backward":
            go, idx, n, c, with_xyz = args
            return 4 * go.numel() + 4 * idx.numel() + 4 * n * c
        if name in ("bottleneck_forward", "bottleneck_backward"):
            # the Bottleneck as a function: x in, y out, coordinates + kNN table (forward); backward: g_y in, g_x out, x, tables.
            # Weights are negligible; the activations saved for the backward are NOT counted (strict I/O figure).
            n, k, c = args[0], args[1], args[2]
            return 8 * n * c + 12 * n + 4 * n * k + (4 * n * c if name == "bottleneck_backward" else 0)
        if name == "pt_layer_forward":  # q,k,v rows once + p + idx + out (SURVEY 8d "fused PT layer fwd")
            xq, idx = args[0], args[4]
            n, c = xq.shape
            return 4 * n * c * 3 + 12 * n + 4 * idx.numel() + 4 * n * c
        if name == "pt_layer_backward":  # the forward's inputs + g_out + the three input gradients
            xq, idx = args[0], args[4]
            n, c = xq.shape
            return 4 * n * c * 3 + 12 * n + 4 * idx.numel() + 4 * n * c + 4 * n * c * 3
        return 0

    def summary(self):
        out = {}
        for n, recs in self.records.items():
            if not recs:
                continue
            ms = [e0.elapsed_time(e1) for e0, e1, _ in recs]
            out[n] = dict(calls=len(recs), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                          avg_bytes=sum(b for _, _, b in recs) / len(recs))
        return out


def cpu_baseline(points):
    """The oracle (a port of the reference algorithms) on this host's cores: one training step on a bounded sample."""
    import oracle
    from pointcloudpdf_amd import _native, engine, synthetic

    be = oracle.backend()
    # threads actually used: capped -- on a 256-core host, 256-way OpenMP/ATen threading of these small per-op loops
    # is slower than 16 threads by two orders of magnitude (measured: 410 s vs seconds for the same sample)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    be.set_num_threads(cores)
    prev = _native._set_backend_for_testing(be)
    try:
        step = engine.OpenSegStep()
        synthetic.fill_parameters_deterministic(step, seed=1)
        step.train()
        batch = synthetic.make_batch([points], first_scene_id=900)
        t0 = time.perf_counter()
        out = step(batch)
        out["loss"].backward()
        dt = time.perf_counter() - t0
    finally:
        _native._set_backend_for_testing(prev)
    # op level (SURVEY 8d "CPU baseline" i / iii): the oracle's kNN (OpenMP over queries) and torch.cdist + topk -- the stand-in for
    # torch-cluster's knn, which is absent on both boxes -- on a bounded sample: the first 20,000 queries of the scene, k = 8
    ops = {}
    try:
        coord, off = batch["coord"], batch["offset"]
        nq = min(20000, points)
        qry, qoff = coord[:nq].contiguous(), torch.tensor([nq], dtype=torch.int32)
        t0 = time.perf_counter(); be.knn_query(8, coord, qry, off, qoff); t_or = time.perf_counter() - t0
        t0 = time.perf_counter()
        for lo in range(0, nq, 2000):
            torch.cdist(qry[lo:lo + 2000], coord).topk(8, dim=1, largest=False)
        t_cd = time.perf_counter() - t0
        ops = {"knn_oracle_queries_per_s": nq / t_or, "knn_cdist_topk_queries_per_s": nq / t_cd,
               "knn_sample": f"{nq} queries over {points} points, k = 8"}
    except Exception as e:   # the baseline is a report, never a reason to lose the line
        ops = {"error": f"{type(e).__name__}: {e}"}
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return dict(value=points / dt, unit="points/s", cores=cores, kind="port", cpu_model=cpu_model, host_cores=os.cpu_count(), ops=ops,
                sample=f"1 step (fwd+bwd) on 1 synthetic scene of {points} points, CPU oracle ops (brute-force kNN, "
                       f"iterative FPS; quadratic in scene size) + torch-CPU layers, {dt:.1f} s wall")


def main():
    args = parse()
    from pointcloudpdf_amd import _native, engine, synthetic

    # Native libraries print through C stdio on stdout (RCCL's version banner at communicator creation); block-buffered, that
    # text would land AFTER the JSON line when the process exits.  File descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank, local_rank, world = engine.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the HIP path has no CPU fallback)"
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.backends.cuda.matmul.allow_tf32 = False
    be = _native.hip_backend()

    scannet = args.workload == "scannet"
    if args.points is None:
        args.points = 150000 if scannet else 100000
    step_kw = dict(in_channels=9, num_classes=20, loss_weight=0.04) if scannet else {}
    if args.pseudo_label:
        from pointcloudpdf_amd import pseudo_label
        step_kw["pseudo_mask_fn"] = pseudo_label.make_pseudo_mask_fn(radius=0.02 * 5, max_neighbor=64, condition_from="msp", beta=1.5,
                                                                    seed_from="ml", seed_range=0.15, num_seed=100, slide_window=True)
    step = engine.OpenSegStep(**step_kw).to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)  # identical "random-init" weights on every rank
    step.train()
    force_dp = bool(os.environ.get("PDFOPS_FORCE_DDP"))   # knob: exercise the N > 1 gradient exchange at world size 1
    use_dp = world > 1 or force_dp
    module = engine.wrap_ddp(step, dev) if (use_dp and args.ddp == "torch") else step
    grad_sync = engine.FlatGradAllReduce(step) if (use_dp and args.ddp == "flat") else None
    opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)

    def scene_sizes(i):
        if args.jitter <= 0:
            return [args.points] * args.scenes
        import numpy as np
        rng = np.random.default_rng(7919 * rank + i)
        return [int(round(args.points * (1.0 + args.jitter * (2.0 * rng.random() - 1.0)))) for _ in range(args.scenes)]

    batch_kw = dict(kind="scannet", unknown=(4, 7, 14, 16)) if scannet else {}
    pool = [synthetic.make_batch(scene_sizes(i), first_scene_id=1000 * rank + 10 * i, device=dev, **batch_kw) for i in range(args.pool)]
    pool_points = [int(b["coord"].shape[0]) for b in pool]
    timer = KernelTimer(be, ["knn_query", "farthest_point_sampling", "group_forward", "group_backward",
                             "pt_layer_forward", "pt_layer_backward", "bottleneck_forward", "bottleneck_backward"])
    timer.install()

    from pointcloudpdf_amd.geometry import GeometryPrefetcher

    prefetcher = GeometryPrefetcher(depth=2) if args.prefetch > 0 else None   # two side streams, alternating groups
    tickets = {}
    submit_host_s = []   # host time of every grouped pre-pass submission

    # Group boundaries sit at warmup + k * D, so the timed region queues exactly steps / D group pre-passes (one pre-pass per
    # trained batch) and, D steps being a whole group's lead time, drains them before the closing fence.  Only ONE group
    # pre-pass is in flight at a time and its latency is the FPS chain (~160 ms however many scenes it carries), so the step
    # time is bounded below by latency / D: D = --steps when that is <= --prefetch, else the largest divisor of --steps in
    # [8, --prefetch], else --prefetch itself (then the window holds ceil(steps / D) groups: more pre-pass work, never less).
    D = args.prefetch
    if D > 0:
        D = max(1, min(D, 64 // max(args.scenes, 1)))   # the grid kNN / radius workspaces hold <= 64 scenes per call (beyond: exact scans)
        if args.steps <= D:
            D = max(args.steps, 1)
        else:
            divs = [d for d in range(8, D + 1) if args.steps % d == 0]
            D = max(divs) if divs else D
    phase = args.warmup % D if D > 0 else 0

    def submit_range(lo, hi):
        """One FPS / kNN launch sequence over the scenes of steps lo .. hi-1 (FPS is a chain of dependent arg-max steps,
        one workgroup per scene: its latency is amortised over the group instead of being paid per step)."""
        group = [pool[j % len(pool)] for j in range(lo, hi)]
        t_sub = time.perf_counter()
        prof = None
        if os.environ.get("PDFOPS_PROFILE_SUBMIT") and len(submit_host_s) == 3:
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        for j, t in enumerate(prefetcher.submit_group(group)):
            tickets[lo + j] = t
        if prof is not None:
            import pstats
            prof.disable(); pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
        submit_host_s.append(time.perf_counter() - t_sub)

    def submit(i):
        """Called with i = step + D right after step's tables were fetched: queues the NEXT group at a boundary."""
        if (i - phase) % D == 0:
            submit_range(i, i + D)

    def one_step(i):
        batch = pool[i % len(pool)]
        data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"],
                    offset_host=batch["offset_host"], segment=batch["segment"])
        if prefetcher is not None:
            data["pdf_geometry"] = prefetcher.get(tickets.pop(i))  # pre-pass of THIS batch, queued with its group
            submit(i + D)                             # (acts once per group) every batch gets exactly one pre-pass
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            out = module(data)
        out["loss"].backward()
        if grad_sync is not None:
            grad_sync.sync(force=force_dp)   # ONE all-reduce (RCCL) over the flat gradient buffer
        opt.step()
        return out

    if prefetcher is not None:
        b0 = phase if phase > 0 else D
        submit_range(0, b0)
        if phase > 0:
            submit_range(b0, b0 + D)

    # ---- optional: the whole step (fwd + bwd + SGD) as ONE captured hipGraph.  The step issues ~3000 kernel launches
    # and is host-bound in eager mode; scene sizes are fixed, so the launch sequence is static.  Inputs and the
    # geometry tables live in static buffers that are refreshed (device-to-device copies) before every replay; the
    # geometry pre-pass itself keeps running eagerly on the side streams.
    use_graph = bool(args.graph) and prefetcher is not None and world == 1 and not use_dp  # (collectives + capture: not attempted)
    graph = None
    if use_graph:
        from pointcloudpdf_amd.geometry import Geometry

        b0 = pool[0]
        static = {k: b0[k].clone() for k in ("coord", "feat", "offset", "segment")}
        static_geom = Geometry(static["coord"], static["offset"], b0["offset_host"]).precompute()
        static_data = dict(coord=static["coord"], feat=static["feat"], offset=static["offset"], offset_host=b0["offset_host"],
                           segment=static["segment"], pdf_geometry=static_geom)
        static_out = {}

        def graph_body():
            opt.zero_grad(set_to_none=False)
            with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
                out = module(dict(static_data))
            out["loss"].backward()
            opt.step()
            static_out["loss"] = out["loss"].detach()

        eager_step = one_step

        def one_step(i):  # noqa: F811
            batch = pool[i % len(pool)]
            geom = prefetcher.get(tickets.pop(i))
            submit(i + D)
            for k in ("coord", "feat", "segment"):
                static[k].copy_(batch[k])
            static_geom.load(geom)
            if graph is None:
                graph_body()
            else:
                graph.replay()
            return static_out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    if use_graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                graph_body()
            graph = g
            fence()
        except Exception as e:  # capture not possible in this configuration: stay eager
            print(f"[bench] hipGraph capture failed, running eager: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].item())

    if rank == 0:
        # points actually processed in the timed steps (sizes differ per batch when --jitter is set; other ranks draw from the
        # same distribution, so the whole-job figure is this rank's count times the world size)
        pts_total = sum(pool_points[(args.warmup + i) % len(pool)] for i in range(args.steps)) * world
        pts_per_step = pts_total / args.steps
        ks = timer.summary()
        dom = max(ks, key=lambda n: ks[n]["total_ms"]) if ks else None
        def roofline_of(name):
            achieved = ks[name]["avg_bytes"] / (ks[name]["avg_ms"] * 1e-3) / 1e9
            return dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=PMC_TRAFFIC_BYTES.get(name),
                        algorithmic_bytes_per_launch=ks[name]["avg_bytes"], avg_launch_ms=ks[name]["avg_ms"],
                        launches_per_step=ks[name]["calls"] / args.steps,
                        gpu_time_share_of_step=ks[name]["total_ms"] / (dt * 1e3))

        roof = roofline_of(dom) if dom else None
        # the gather family is what the HBM roofline is meaningful for (FPS / kNN are latency / VALU bound by design)
        second = [n for n in ("bottleneck_backward", "bottleneck_forward", "pt_layer_backward", "pt_layer_forward", "group_backward", "group_forward") if n in ks]
        roof2 = roofline_of(max(second, key=lambda n: ks[n]["total_ms"])) if second else None
        line = {
            "metric": "points/sec fwd+bwd (PT-v1 Seg50 + PDF U-decoder, 100k-pt scenes)",
            "value": pts_per_step * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16-autocast" if args.amp else "f32",
            "data": "synthetic",
            "config": {"workload": f"{'ScanNet' if scannet else 'S3DIS'}-shaped synthetic voxelised scenes, {args.scenes} x {args.points} points per GPU, "
                                   "PointTransformer-Seg50 + PointPdf-v1m1 U-decoder, fwd+bwd+SGD, geometry recomputed every step"
                                   + (", PDF pseudo-label pass inside the step" if args.pseudo_label else ""),
                       "scenes_per_gpu": args.scenes, "points_per_scene": args.points, "size_jitter": args.jitter, "parallelism": f"dp{world}",
                       "gradient_exchange": (args.ddp if use_dp else "none")},
            "per_gpu_points_per_s": pts_per_step * args.steps / dt / world,
            "loss": loss,
            "geometry_prefetch_group": D,
            "prepass_submit_host_ms": (1e3 * min(submit_host_s)) if submit_host_s else None,   # host time of one group submission (warm)
            "hipgraph": bool(use_graph and graph is not None),
            "kernels": ks,
            "roofline": roof,
            "roofline_gather_family": roof2,
        }
        if world == 1 and not args.no_ops_roofline and not scannet:
            # the pointops drop-in ops on their own (level-1 shapes of this config: 200k points, c = 32, k = 8), HIP-event timed on
            # the launching stream, against the 8 TB/s HBM peak with the SURVEY 8(d) byte counts (tools/ops_roofline.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ops_roofline
            line["roofline_ops"] = [dict(op=r["op"], us=round(r["us"], 1), GBps=round(r["GBps"], 1), frac=round(r["frac"], 4))
                                    for r in ops_roofline.run(iters=10, level2=False, references=False)]
        if world == 1 and not args.no_cpu_baseline and not scannet:   # (the CPU baseline is quoted on the headline workload)
            line["cpu_baseline"] = cpu_baseline(args.cpu_points)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)   # drain C stdio into stderr, then give stdout back for the ONE JSON line
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
