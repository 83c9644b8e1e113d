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

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # (as pointcloudpdf_amd/__init__.py does: before HIP initialises)
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak (same guide: fp32 matrix = fp32 vector rate)
TRAFFIC_GLOB = os.path.join(ROOT, "profiles", "r*_traffic.json")   # tools/traffic_json.py output; the newest file measured on the current kernel sources is used


def kernel_source_hash():
    """Hash of the kernel sources the PMC numbers belong to (csrc/*.hip, csrc/*.h, include/pdfops.h)."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "pointcloudpdf_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(cs, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "pdfops.h"), "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """HBM traffic per launch from the rocprofv3 PMC passes of `bash tools/prof_round.sh <tag>` (separate FETCH_SIZE / WRITE_SIZE runs;
    FETCH_SIZE doubled for gfx950 and KB -> bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes), written by
    tools/traffic_json.py.  The file records the hash of the kernel sources it was measured on: numbers of other sources are not
    reported (traffic = null, note says why) rather than silently going stale."""
    import glob
    want, seen = kernel_source_hash(), []
    for path in sorted(glob.glob(TRAFFIC_GLOB), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("kernel_source_hash") == want:
            t["file"] = os.path.relpath(path, ROOT)
            return t, None
        seen.append(f"{os.path.basename(path)} ({t.get('kernel_source_hash')})")
    return {}, ("no profiles/r*_traffic.json" if not seen else "measured on other kernel sources: " + ", ".join(seen[:3]))


SQ_GLOB = os.path.join(ROOT, "profiles", "r*_sq.json")   # tools/rocpd_sq.py --json output (SQ counter passes of tools/prof_sq.sh)
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 matrix peak (same guide)


def load_sq():
    """Counter-based limiter / matrix-pipe figures per kernel (builder-side rocprofv3 --pmc passes: `bash tools/prof_sq.sh <tag>`), hash-locked
    to the kernel sources like the traffic file."""
    import glob
    want = kernel_source_hash()
    for path in sorted(glob.glob(SQ_GLOB), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("kernel_source_hash") == want:
            t["file"] = os.path.relpath(path, ROOT)
            t["file_date"] = time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))
            return t
    return None


def dense_flops_forward(n1, in_channels=6, num_classes=13, enc_bottlenecks=(1, 2, 3, 5, 2)):
    """Forward FLOPs of every nn.Linear of PointTransformer-Seg50 + the PDF U-decoder for ONE scene of n1 points: 2 * rows * in * out per
    layer (the probe there gives 0.365 + 0.018 MFLOP per point, which this reproduces).  Level sizes n_l = n_{l-1} // 4
    (point_transformer_seg.py:96-99), planes 32..512, nsample 8 / 16; per Bottleneck (point_transformer_seg.py:171-192, 19-78): linear1,
    linear3, q / k / v on n rows (10 n c^2), linear_p (3 -> 3 -> c) and linear_w (c -> c/8 -> c/8) on n k rows; TransitionDown
    (:81-119): Linear(3 + c_in, c) on n k rows (level 1: Linear(in, 32) on n rows); TransitionUp (:122-168) and the recognizer's five
    TransitionUp + confidence head (pt_v1.py:8-44); cls head (:229-234)."""
    planes, ks = [32, 64, 128, 256, 512], [8, 16, 16, 16, 16]
    n = [int(n1)]
    for _ in range(4):
        n.append(n[-1] // 4)
    seg = rec = 0
    for l in range(5):
        N, c, k = n[l], planes[l], ks[l]
        seg += 2 * N * in_channels * c if l == 0 else 2 * N * k * (3 + planes[l - 1]) * c
        seg += (enc_bottlenecks[l] + 1) * (10 * N * c * c + 2 * N * k * (9 + 3 * c + c * c // 8 + (c // 8) ** 2))
        if l == 4:
            seg += 2 * N * (2 * c) * c
            rec += 4 * N * c * c
        else:
            seg += 2 * N * c * c + 2 * n[l + 1] * planes[l + 1] * c
            rec += 2 * N * c * c + 2 * n[l + 1] * planes[l + 1] * c
    seg += 2 * n[0] * (32 * 32 + 32 * num_classes)
    rec += 2 * n[0] * (32 * 32 + 32 * 1)
    return float(seg + rec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--throttle", action="store_true",
                    help="profiling aid: synchronise after every encoder / decoder block (forward and backward), so that at most ~100 "
                         "dispatches are in flight -- rocprofv3 --pmc serialises kernels and its interception dies ('AQL packet is "
                         "malformed' / SIGSEGV) when the host runs a whole step (1,100 dispatches) ahead of the device")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="fused",
                    help="fused = SGD over all parameter tensors as one HIP launch (engine.FusedSGD); torch = torch.optim.SGD(fused=True)")
    ap.add_argument("--workload", choices=["s3dis", "scannet", "stratified"], default="s3dis",
                    help="s3dis = BASELINE config 2 / 3 (6 input channels, 13 classes, the headline); scannet = config 4 shape (coord + "
                         "colour + normal = 9 channels, 20 classes, unknown classes {4, 7, 14, 16}, 150k points per scene unless --points); "
                         "stratified = config 5: StratifiedTransformer ST-v1m1 + ST-v1m1-Recognizer (pointops2 window attention) on S3DIS-shaped "
                         "scenes of 80k points (the reference's SphereCrop point_max) unless --points")
    ap.add_argument("--pseudo-label", type=int, default=0,
                    help="1 = run the PDF pseudo-label pass inside the step (config 4; recognizer settings of "
                         "configs/scannet/openseg-pt-v1-0-pointpdf-v1m1-base.py:40-58) instead of the fixed 1-in-7 stand-in mask")
    ap.add_argument("--points", type=int, default=None, help="points per scene (default 100000, scannet: 150000)")
    ap.add_argument("--scenes", type=int, default=2, help="scenes per GPU (batch size per rank)")
    ap.add_argument("--pool", type=int, default=3, help="distinct batches per rank to rotate through")
    ap.add_argument("--jitter", type=float, default=0.0,
                    help="scene sizes drawn in points * (1 +- jitter), seeded per rank and scene (SURVEY 8d config 3, secondary "
                         "number: straggler imbalance across ranks); 0 = every scene exactly --points")
    ap.add_argument("--size-classes", type=int, default=0,
                    help="with --jitter: draw the scene sizes of a batch from this many fixed signatures (batch i has the sizes of class "
                         "i %% K) and give each class a captured graph of its own (engine.TrainStep(max_captures=K)); 0 = every batch its "
                         "own sizes (eager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops-roofline", action="store_true", help="skip the per-op HBM roofline micro-benchmark (rank 0, N = 1)")
    ap.add_argument("--cpu-points", type=int, default=100000, help="scene size of the bounded CPU-baseline sample")
    ap.add_argument("--st-group", type=int, default=3, help="--workload stratified: batches per grouped coordinate pre-pass (1 = one batch ahead)")
    ap.add_argument("--amp", nargs="?", const="f16", default=None, choices=["bf16", "f16"],
                    help="torch.autocast around the step (reference: enable_amp = True, engines/train.py:340-363): the streaming Linear products "
                         "run with fp16 (default, as the reference's autocast) / bfloat16 operands on the 16x16x16 matrix-core instructions, fp32 storage and accumulation "
                         "(dense.fp32_path).  f16 scales the loss dynamically as the reference's GradScaler does (engine.DeviceGradScaler: scale, "
                         "found-inf flag and growth tracker on the device, so the step still replays as a graph)")
    ap.add_argument("--storage", choices=["f32", "bf16"], default=os.environ.get("PDFOPS_STORAGE", "f32"),
                    help="bf16: the reduced-precision variant -- the fused PointTransformerLayer keeps its saved / scratch row arrays (H, G2, "
                         "softmax weights, g_r rows) as bfloat16 with fp32 accumulation (the reference trains under AMP); the headline stays f32")
    ap.add_argument("--prefetch", type=int, default=32,
                    help="geometry pre-pass group: the pre-pass of the NEXT `prefetch` batches runs as one launch sequence on a side "
                         "stream while the current group trains (0 = inline, serial)")
    ap.add_argument("--graph", choices=["auto", "0", "1"], default="auto",
                    help="1 = forward + backward of the step replayed as ONE captured hipGraph (engine.CapturedStep: batch tensors and the "
                         "batch's geometry tables staged into fixed-address buffers by one copy launch per step); the optimizer, the "
                         "gradient exchange and the geometry pre-pass stay eager.  Needs identical scene sizes in every batch (what "
                         "SphereCrop(point_max) gives the reference's trainer): auto = on unless --jitter (without --size-classes) / "
                         "--throttle / --ddp torch / the stratified workload ask for something the capture does not cover.  With --pseudo-label "
                         "the sync-free pass (csrc/region_grow.hip) is recorded into the same graph (PDFOPS_PL_STATIC=0: the host-driven pass "
                         "of rounds 1-4, eager step, or TWO graphs around it with --graph 1)")
    ap.add_argument("--ddp", choices=["flat", "torch"], default="flat",
                    help="gradient exchange for N > 1: one flat-buffer all-reduce after the backward (engine.FlatGradAllReduce) or "
                         "torch DistributedDataParallel (per-parameter bucket copies: +3 ms per step, measured)")
    ap.add_argument("--no-latency-sweep", action="store_true",
                    help="skip the look-ahead sweep (serial step time with --prefetch 0 and groups of 1 / 2 / 3 batches; rank 0, N = 1)")
    ap.add_argument("--no-affinity", action="store_true", help="N > 1: do not pin each rank to its own contiguous share of the host's cores")
    ap.add_argument("--no-n1-reference", action="store_true",
                    help="N > 1 started without torchrun: do not time the one-rank run that efficiency_vs_n1 is quoted against")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun: this process never touches the GPU; it starts N ranks (one per GPU, the
    environment torchrun would give them, rendezvous on 127.0.0.1), relays rank 0's JSON line and adds the weak-scaling efficiency
    against a one-rank run of the same command (timed first, same steps / warmup).  Mirrors pointcept/engines/launch.py:74-113
    (one process per GPU) without re-executing anything that has initialised HIP."""
    import socket
    import subprocess

    # PDFOPS_BENCH_SHARED_GPU=1 (functional check of the N-rank path on a box with fewer GPUs; tests/test_gpu_model.py): the ranks share
    # the visible devices round-robin and rendezvous over gloo -- RCCL refuses two ranks on one device.  Never a performance number.
    if torch.cuda.device_count() < args.gpus and not os.environ.get("PDFOPS_BENCH_SHARED_GPU"):   # (device_count does not initialise the GPU on this image)
        print(f"[bench] --gpus {args.gpus} but only {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    argv = [a for a in sys.argv[1:]]
    points = args.points if args.points is not None else (150000 if args.workload == "scannet" else (80000 if args.workload == "stratified" else 100000))
    preflight_host(args.gpus, args.scenes, points, max(args.prefetch, 0))   # (exits 3 with the reason when the node cannot hold the ranks)

    def run(world, extra):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PDFOPS_BENCH_CHILD="1")
            cmd = [sys.executable, os.path.abspath(__file__)] + argv + extra
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out, _ = procs[0].communicate()
        codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
        return max(abs(c) for c in codes), out

    n1 = None
    if not args.no_n1_reference:
        rc, out = run(1, ["--gpus", "1", "--no-cpu-baseline", "--no-ops-roofline", "--no-latency-sweep"])
        try:
            n1 = json.loads(out.strip().splitlines()[-1])
        except (ValueError, IndexError):
            print(f"[bench] one-rank reference run failed (exit {rc})", file=sys.stderr)
    rc, out = run(args.gpus, [])
    try:
        line = json.loads(out.strip().splitlines()[-1])
    except (ValueError, IndexError):
        print(f"[bench] the {args.gpus}-rank run printed no JSON line (exit {rc})", file=sys.stderr)
        return rc or 1
    if n1 is not None:
        line["n1_reference"] = {"ms_per_step": n1["ms_per_step"], "value": n1["value"]}
        line["efficiency_vs_n1"] = (line["value"] / line["n_gpus"]) / n1["value"]
    print(json.dumps(line), flush=True)
    return rc


class KernelTimer:
    """HIP-event timing of individual backend calls on torch's current stream (the stream the kernels are launched on).  The
    per-step calls (18 Bottlenecks forward and backward, ...) carry events on every `every`-th step of the timed region; the
    geometry pre-pass calls (one launch sequence per GROUP of batches) on all of them."""

    ALWAYS = ("knn_query", "farthest_point_sampling")

    def __init__(self, backend, names):
        self.backend, self.names = backend, names
        self.records = {n: [] for n in names}
        self.enabled = False
        self.sample = True    # per-step switch: the timed loop samples every `every`-th step (a pair of timing events per call costs
        self.every = int(os.environ.get("PDFOPS_BENCH_TIMER_EVERY", "4"))   # ~0.5 ms per step of marker packets when taken on all of them)
        self._orig = {}

    def install(self):
        for n in self.names:
            orig = getattr(self.backend, n)
            self._orig[n] = orig

            def wrapped(*a, _orig=orig, _n=n, **k):
                if not (self.enabled and (self.sample or _n in self.ALWAYS)):
                    return _orig(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _orig(*a, **k)
                e1.record()
                self.records[_n].append((e0, e1, self._bytes(_n, a, out), self.mfma_flops(_n, a)))
                return out

            setattr(self.backend, n, wrapped)

    # ---- replayed measuring steps (engine.CapturedStep(split_calls=): the step as a sequence of graphs cut around the named calls)
    STEP_CALLS = ("group_forward", "group_backward", "pt_layer_forward", "pt_layer_backward", "bottleneck_forward", "bottleneck_backward")

    def describe(self, name, args, out):
        """What a segmented capture remembers about a named call: its algorithmic bytes / flops (the tensors are gone at replay time)."""
        return (self._bytes(name, args, out), self.mfma_flops(name, args))

    def on_call(self, name, info, replay):
        """The named call of a replayed measuring step: its graph between two HIP events on the replay's stream."""
        if not self.enabled:
            return replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        replay()
        e1.record()
        self.records[name].append((e0, e1) + tuple(info))

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
        if name.startswith(("attention_step", "dot_prod_with_idx", "window_")):
            # libs/pointops2 CSR-by-query ops: every operand once (q / k / v rows (N, C), per-edge scalars (M, h), the edge index and the
            # 3 quantised offsets per edge, the relative-position tables (L, h, d, 3)); the backward passes read the same plus the
            # incoming gradient and write the operands' gradients
            ts = [t for t in list(args) + (list(out) if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(t)]
            return sum(t.numel() * t.element_size() for t in ts)
        if name == "pt_layer_backward":  # the forward's inputs + g_out + the three input gradients
            xq, idx = args[0], args[4]
            n, c = xq.shape
            return 4 * n * c * 3 + 12 * n + 4 * idx.numel() + 4 * n * c + 4 * n * c * 3
        return 0

    @staticmethod
    def mfma_flops(name, args):
        """ALGORITHMIC flops of the libs/pointops2 table ops (config 5), from their definitions -- not the flops of whatever formulation
        the kernel uses (round 3 counted the one-hot matrix products of the round-1 kernels; the factored kernels of csrc/window_attention.hip
        no longer run most of them).  dot_prod_with_idx_v3 (relative_pos_encoding_cuda_kernel_v2.cu:247-330): per edge m and head h,
        out = sum over 3 axes and d channels of q * table_q[r] + k * table_k[r]: 6 d multiply-adds = 12 M C flops forward; every
        multiply-add has two gradient multiply-adds: 24 M C backward.  attention_step2_with_rel_pos_value_v2 (:375-484): out[n] += attn *
        (v + sum of 3 table rows): 5 M C forward; backward (grad_attn dot, grad_v, 3 table rows): 10 M C."""
        if name == "dot_prod_with_idx_v3_backward":
            q = args[1]
            return 24.0 * args[5].shape[0] * q.shape[1] * q.shape[2]
        if name == "window_logits_backward":   # (g, q, k, index1, ...): dot_prod_with_idx_v3's 24 M C + attention_step1's 4 M C
            q = args[1]
            return 28.0 * args[3].shape[0] * q.shape[1] * q.shape[2]
        if name == "window_attention_core_backward":   # (go, qkv, attn, index1, ...): the three ops' 4 + 24 + 10 M C
            return 38.0 * args[3].shape[0] * args[0].shape[1]
        if name == "attention_step2_with_rel_pos_value_v2_backward":
            v = args[2]
            return 10.0 * args[1].shape[0] * v.shape[1] * v.shape[2]
        return 0.0

    def summary(self):
        out = {}
        for n, recs in self.records.items():
            if not recs:
                continue
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            out[n] = dict(calls=len(recs), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                          avg_bytes=sum(r[2] for r in recs) / len(recs))
            fl = [r[3] for r in recs if len(r) > 3]
            if fl and sum(fl) > 0:
                out[n]["avg_mfma_flops"] = sum(fl) / len(fl)
        return out


def cpu_baseline(points):
    """The oracle (a port of the reference algorithms) on this host's cores: one training step on a bounded sample."""
    import oracle
    from pointcloudpdf_amd import _native, engine, synthetic

    be = oracle.backend()
    host = os.cpu_count() or 1

    def run_once(sizes, oracle_threads, torch_threads):
        """One training step (fwd + bwd) of the same host code on the CPU oracle -> seconds."""
        torch.set_num_threads(torch_threads)
        be.set_num_threads(oracle_threads)
        prev = _native._set_backend_for_testing(be)
        try:
            step = engine.OpenSegStep()
            synthetic.fill_parameters_deterministic(step, seed=1)
            step.train()
            batch = synthetic.make_batch(sizes, first_scene_id=900)
            t0 = time.perf_counter()
            out = step(batch)
            out["loss"].backward()
            return time.perf_counter() - t0, batch
        finally:
            _native._set_backend_for_testing(prev)

    # (a) round 1 / 2's setting: 16 threads for both the oracle's OpenMP loops and ATen, one 100k-point scene;
    # (b) the oracle's kNN / FPS / gather loops on ALL host cores (they scale: one query / one scene per iteration), ATen's intra-op
    #     threads kept at 16 (256-way threading of its small per-op loops is slower by two orders of magnitude: 410 s measured), on the
    #     headline batch itself (2 scenes).  `value` is the better of the two: the baseline must not be held back by a thread choice.
    cores_a = min(host, 16)
    dt_a, batch = run_once([points], cores_a, cores_a)
    runs = [dict(points_per_s=points / dt_a, seconds=dt_a, scenes=1, oracle_threads=cores_a, torch_threads=cores_a)]
    # thread sweep of the oracle's kNN (the dominant CPU cost: brute force, OpenMP over queries) on a 5,000-query sample: more threads
    # are used for a second whole-step run only where they pay on THIS host (measured on the 256-thread EPYC 9575F of the GPU box:
    # all 256 threads make the step 36x slower than 16 -- 388 s --, so an unconditional all-core run is neither a fair nor a bounded baseline)
    sweep = {}
    try:
        coord, off = batch["coord"], batch["offset"]
        qs, qoff = coord[:5000].contiguous(), torch.tensor([5000], dtype=torch.int32)
        for t in sorted({cores_a, 32, 64, 128, host}):
            if t > host:
                continue
            be.set_num_threads(t)
            t0 = time.perf_counter(); be.knn_query(8, coord, qs, off, qoff); sweep[t] = 5000 / (time.perf_counter() - t0)
        best_t = max(sweep, key=sweep.get)
        if best_t != cores_a and sweep[best_t] >= 1.3 * sweep[cores_a]:
            dt_b, _ = run_once([points], best_t, cores_a)
            runs.append(dict(points_per_s=points / dt_b, seconds=dt_b, scenes=1, oracle_threads=best_t, torch_threads=cores_a))
    except Exception as e:   # noqa: BLE001
        runs.append(dict(error=f"{type(e).__name__}: {e}"))
    runs.append(dict(knn_queries_per_s_by_oracle_threads={str(k): round(v, 1) for k, v in sweep.items()}))
    best = max((r for r in runs if "points_per_s" in r), key=lambda r: r["points_per_s"])
    dt, cores = best["seconds"], max(best["oracle_threads"], best["torch_threads"])
    be.set_num_threads(best["oracle_threads"])
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
    return dict(value=best["points_per_s"], unit="points/s", cores=cores, kind="port", cpu_model=cpu_model, host_cores=os.cpu_count(), ops=ops,
                runs=runs,
                sample=f"1 step (fwd+bwd) on {best['scenes']} synthetic scene(s) of {points} points, CPU oracle ops (brute-force kNN, "
                       f"iterative FPS; quadratic in scene size; OpenMP over {best['oracle_threads']} threads) + torch-CPU layers "
                       f"({best['torch_threads']} threads), {dt:.1f} s wall; `runs` lists every thread setting tried")


def pin_rank_to_cores(args):
    """One contiguous share of the host's cores per rank (LOCAL_RANK-th of LOCAL_WORLD_SIZE shares of this process's affinity mask): N
    Python ranks that each issue ~1,000 launches per step otherwise migrate across the sockets of the host and share cores with each
    other's pre-pass threads.  Contiguous logical ids keep a rank on one socket / NUMA node of a two-socket EPYC host (the launcher of
    the reference leaves placement to the OS: pointcept/engines/launch.py:74-131).  Returns the description put into the JSON line."""
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if args.no_affinity or world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    cpus = sorted(os.sched_getaffinity(0))
    share = len(cpus) // world
    if share < 2:
        return None
    mine = cpus[lr * share:(lr + 1) * share]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(share, 16)))
    return f"rank {lr}: cpus {mine[0]}-{mine[-1]} ({len(mine)} of {len(cpus)})"


def _mem_available_gib():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 2.0 ** 20
    except OSError:
        pass
    return None


def preflight_host(world, scenes, points, group, out=sys.stderr):
    """Before any rank touches a GPU: will `world` ranks fit this HOST?  One rank holds its pool of synthetic batches on the device, but
    its process (torch + the HIP runtime + RCCL + pinned staging) is ~3.5 GiB of resident host memory, and the look-ahead keeps `group`
    batches of index tables in flight per rank.  A node that cannot hold all ranks kills one of them without a message (the container
    has no swap): fail here, loudly, instead.  Returns a dict for the JSON line; raises SystemExit(3) when the job cannot fit."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    avail = _mem_available_gib()
    per_rank_host = 3.5 + 1e-9 * 64.0 * scenes * points   # process image + the pool's host-side copies (offsets, a few staging buffers)
    need = world * per_rank_host
    info = {"ranks": world, "host_cores": cores, "host_mem_available_gib": avail, "host_mem_needed_gib": round(need, 1),
            "lookahead_batches_per_rank": group}
    problems = []
    if cores < world:
        problems.append(f"{world} ranks on {cores} visible cores: every rank needs a core of its own for its enqueue thread")
    if avail is not None and avail < need:
        problems.append(f"{world} ranks need ~{need:.0f} GiB of host memory, {avail:.0f} GiB available")
    if problems:
        print("[bench] pre-flight FAILED: " + "; ".join(problems), file=out, flush=True)
        raise SystemExit(3)
    if cores < 4 * world:
        print(f"[bench] pre-flight: {cores} cores for {world} ranks (< 4 per rank): the host side of a step may become the bound", file=out, flush=True)
    return info


def preflight_device(dev, world, scenes, points, group, strat=False, out=sys.stderr):
    """In every rank, after torch.cuda.set_device and before the first allocation: does this rank's GPU have room for the step's
    activations (measured: 4.8 GiB at 2 x 100k points -- `hbm_peak_gib` of the N = 1 line -- scaled by the point count) plus `group`
    batches of look-ahead tables (~0.11 GB per 2 x 100k-point batch, twice: the group in flight and the one being consumed)?  On a node
    whose GPUs are shared with other jobs (or a rank mapped onto a busy device) the run would otherwise die minutes in, inside hipMalloc."""
    free, total = torch.cuda.mem_get_info(dev)
    pts = scenes * points
    need = (4.8 * pts / 2e5 * (1.6 if strat else 1.0) + 2 * group * 0.11 * pts / 2e5 + 2.0) * 2.0 ** 30   # + 2 GiB: graph pool slack, RCCL buffers
    info = {"device_free_gib": round(free / 2.0 ** 30, 1), "device_total_gib": round(total / 2.0 ** 30, 1), "device_needed_gib": round(need / 2.0 ** 30, 1)}
    if free < need:
        print(f"[bench] pre-flight FAILED on {dev}: {free / 2.0 ** 30:.1f} GiB free of {total / 2.0 ** 30:.1f}, this rank needs ~{need / 2.0 ** 30:.1f} GiB "
              f"({scenes} x {points} points, look-ahead of {group} batches)", file=out, flush=True)
        raise SystemExit(3)
    return info


class TimedExchange:
    """The gradient exchange with a pair of HIP events around every call (N > 1 only): `exposed_allreduce_ms` of the line is the time the
    training stream spends inside pack + all-reduce + scale + unpack, i.e. what data parallelism adds to a step that overlaps nothing."""

    def __init__(self, inner):
        self.inner, self.events, self.on = inner, [], False

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def sync(self, force=False):
        if not self.on:
            return self.inner.sync(force=force)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.inner.sync(force=force)
        e1.record()
        self.events.append((e0, e1))

    def mean_ms(self):
        return (sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)) if self.events else None


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    from pointcloudpdf_amd import _native, engine, synthetic

    # Native libraries print through C stdio on stdout (RCCL's version banner at communicator creation); block-buffered, that
    # text would land AFTER the JSON line when the process exits.  File descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    affinity = pin_rank_to_cores(args)
    shared_gpu = bool(os.environ.get("PDFOPS_BENCH_SHARED_GPU")) and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if shared_gpu:   # (see launch_ranks: the group exists before engine.init_distributed, which then only reads the environment)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        torch.distributed.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, local_rank, world = engine.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the HIP path has no CPU fallback)"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev_index = local_rank % torch.cuda.device_count() if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    torch.backends.cuda.matmul.allow_tf32 = False
    be = _native.hip_backend()
    be.set_storage(args.storage)

    scannet, strat = args.workload == "scannet", args.workload == "stratified"
    if args.points is None:
        args.points = 150000 if scannet else (80000 if strat else 100000)
    step_kw = dict(in_channels=9, num_classes=20, loss_weight=0.04) if scannet else (dict(backbone="ST-v1m1", loss_weight=0.008) if strat else {})
    pre = {}
    if world > 1 and rank == 0 and not os.environ.get("PDFOPS_BENCH_CHILD"):   # (started by torchrun: nobody ran the host check yet)
        pre.update(preflight_host(world, args.scenes, args.points, max(args.prefetch, 0)))
    if not shared_gpu:
        pre.update(preflight_device(dev, world, args.scenes, args.points, max(args.prefetch, 0), strat=strat))
    if world > 1 and not shared_gpu and torch.distributed.get_backend() != "nccl":
        raise SystemExit(f"bench.py: {world} ranks on their own GPUs must exchange gradients over RCCL, the process group is '{torch.distributed.get_backend()}'")
    st_ahead = 0
    if strat:
        # ST's coordinate-only work (FPS chain + window edge tables) runs one batch ahead on a worker thread + side stream
        # (stratified.StratifiedPrefetcher); --prefetch 0 keeps it inside the forward.  PointTransformer-V1's pre-pass does not apply.
        # With a group of G > 1 batches (--st-group, default 3) the farthest-point chain of the G batches runs as ONE launch sequence
        # (one workgroup per scene: G batches cost the latency of one), as PointTransformer-V1's grouped pre-pass does.
        st_ahead, args.prefetch = (max(1, args.st_group) if args.prefetch > 0 else 0), 0
    if args.pseudo_label:
        from pointcloudpdf_amd import pseudo_label
        step_kw["pseudo_mask_fn"] = pseudo_label.make_pseudo_mask_fn(radius=0.02 * 5, max_neighbor=64, condition_from="msp", beta=1.5,
                                                                    seed_from="ml", seed_range=0.15, num_seed=100, slide_window=True)
    step = engine.OpenSegStep(**step_kw).to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)  # identical "random-init" weights on every rank
    step.train()
    if args.throttle:
        from pointcloudpdf_amd import point_transformer as _pt
        kinds = tuple(getattr(_pt, n) for n in ("Bottleneck", "TransitionDown", "TransitionUp") if hasattr(_pt, n))
        for mod in step.modules():
            if isinstance(mod, kinds):
                mod.register_forward_hook(lambda *a: torch.cuda.synchronize())
                mod.register_full_backward_hook(lambda *a: torch.cuda.synchronize())
    force_dp = bool(os.environ.get("PDFOPS_FORCE_DDP"))   # knob: exercise the N > 1 gradient exchange at world size 1
    use_dp = world > 1 or force_dp
    module = engine.wrap_ddp(step, dev) if (use_dp and args.ddp == "torch") else step
    grad_sync = engine.FlatGradAllReduce(step) if (use_dp and args.ddp == "flat") else None
    if grad_sync is not None and world > 1:
        grad_sync = TimedExchange(grad_sync)
    if args.optimizer == "fused":   # one launch over all 304 tensors (csrc/optim.hip); "torch" = torch.optim.SGD(fused=True), 13 launches
        opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    else:
        opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)

    def scene_sizes(i):
        if args.jitter <= 0:
            return [args.points] * args.scenes
        import numpy as np
        rng = np.random.default_rng(7919 * rank + (i % args.size_classes if args.size_classes > 0 else i))
        return [int(round(args.points * (1.0 + args.jitter * (2.0 * rng.random() - 1.0)))) for _ in range(args.scenes)]

    batch_kw = dict(kind="scannet", unknown=(4, 7, 14, 16)) if scannet else {}
    pool = [synthetic.make_batch(scene_sizes(i), first_scene_id=1000 * rank + 10 * i, device=dev, **batch_kw) for i in range(args.pool)]
    pool_points = [int(b["coord"].shape[0]) for b in pool]
    names = ["knn_query", "farthest_point_sampling", "group_forward", "group_backward",
             "pt_layer_forward", "pt_layer_backward", "bottleneck_forward", "bottleneck_backward"]
    if strat:   # config 5: the libs/pointops2 window-attention ops (csrc/window_attention.hip)
        names += ["attention_step1_v2", "attention_step1_v2_backward", "dot_prod_with_idx_v3", "dot_prod_with_idx_v3_backward",
                  "window_logits", "window_logits_backward",   # (the two ops above as one, csrc/window_attention_bwd.hip)
                  "window_attention_core", "window_attention_core_backward",   # (all of them + the softmax as one node: what the model calls)
                  "attention_step2_with_rel_pos_value_v2", "attention_step2_with_rel_pos_value_v2_backward"]
    timer = KernelTimer(be, names)
    timer.install()

    amp_dtype = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[args.amp]
    # fp16 operands: the reference's AMP loop scales the loss dynamically (torch GradScaler, engines/train.py:343-355); here the same
    # policy with the scale / found-inf flag on the device (engine.DeviceGradScaler), so that the step still replays as a graph
    scaler = engine.DeviceGradScaler(dev) if (args.amp == "f16" and args.optimizer == "fused") else None
    static_scale = 4096.0 if (args.amp == "f16" and scaler is None) else 1.0
    graph_ok = ((args.jitter <= 0 or args.size_classes > 0) and not args.throttle and not strat
                and not (use_dp and args.ddp == "torch"))
    if args.graph == "1" and not graph_ok:
        raise SystemExit("bench.py: --graph 1 needs fixed scene sizes and the plain PointTransformer step (see --help)")
    # The schedule of the run is PRODUCT code (pointcloudpdf_amd/engine.py): GroupedGeometryLoader owns the grouped look-ahead of the
    # coordinate-only pre-pass (what to submit when, on which stream, waiting for what), TrainStep owns the step itself (graph replay
    # when the batch has the captured shape, eager otherwise; gradient exchange; optimizer; loss scaling).  This file only feeds
    # batches, times the region and reports.
    trainer = engine.TrainStep(step, opt, exchange=grad_sync, scaler=scaler, autocast=amp_dtype, module=module, force_exchange=force_dp,
                               graph=(args.graph == "1" or (args.graph == "auto" and graph_ok and (not args.pseudo_label or getattr(step_kw.get("pseudo_mask_fn"), "capturable", False)))), loss_scale=static_scale,
                               max_captures=(args.size_classes if (args.jitter > 0 and args.size_classes > 0) else 1))
    st_prefetcher = None
    if st_ahead:
        from pointcloudpdf_amd.stratified import StratifiedPrefetcher
        st_prefetcher = StratifiedPrefetcher(step.model.backbone, windows=not os.environ.get("PDFOPS_ST_FPS_ONLY"))

    def batch_stream(start=0):
        i = start
        while True:
            b = pool[i % len(pool)]
            yield dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"])
            i += 1

    def make_loader(D, warmup, steps):
        """The look-ahead group for `warmup + steps` steps: D batches per pre-pass (0 = inline, serial step), capped by the scene budget
        of one grouped call, and a divisor of `steps` where one exists -- the timed region then queues exactly steps / D group
        pre-passes (every trained batch gets exactly one full pre-pass, nothing is cached); the first group is sized so that a group
        boundary coincides with the end of the warm-up."""
        if st_prefetcher is not None:
            return engine.GroupedGeometryLoader(batch_stream(), group=st_ahead, prefetcher=st_prefetcher, key="st_geometry", submit_delay=0), st_ahead
        if D > 0:
            D = max(1, min(D, engine.GroupedGeometryLoader.MAX_SCENES // max(args.scenes, 1)))
            if steps <= D:
                D = max(steps, 1)
            else:
                divs = [d for d in range(min(8, D), D + 1) if steps % d == 0]
                D = max(divs) if divs else D
        first = (warmup % D or D) if D > 0 else None
        # (a pseudo-label pass names the coordinate-only table it wants from the pre-pass: its radius table)
        plan = {} if os.environ.get("PDFOPS_PL_INLINE_RADIUS") else dict(getattr(step_kw.get("pseudo_mask_fn"), "prepass_plan", {}))
        return engine.GroupedGeometryLoader(batch_stream(), group=D, first_group=first, threaded=bool(os.environ.get("PDFOPS_PREPASS_THREAD")), **plan), D

    sched_warm = [0]

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    class Region:
        pass

    def timed(D, warmup, steps, with_timer=False):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize pairs -> (seconds, last output, region info)."""
        loader, D = make_loader(D, warmup, steps)
        if D > 0 and st_prefetcher is None:   # allocator pools of the pre-pass streams at their steady state (GroupedGeometryLoader.warm)
            sched_warm[0] = loader.warm([next(batch_stream()) for _ in range(D)])
        it = iter(loader)
        segmented = False
        for w in range(warmup):
            b = next(it)
            # the LAST warm-up step makes (and runs once) the segmented capture the measuring step of the timed region replays: the
            # step as a sequence of graphs cut around the timed calls (engine.CapturedStep(split_calls=)); without it -- eager schedules,
            # a stack that cannot capture, PDFOPS_BENCH_EAGER_SAMPLE=1 -- the measuring step runs eagerly as in rounds 1-5
            if (with_timer and w == warmup - 1 and w >= 1 and trainer.captured is not None and not strat
                    and not os.environ.get("PDFOPS_BENCH_EAGER_SAMPLE") and b.get("pdf_geometry") is not None):
                segmented = trainer.instrument(b, b["pdf_geometry"], KernelTimer.STEP_CALLS, timer.describe)
            out = trainer(b, on_call=(lambda name, info, replay: replay()) if segmented else None)
        fence()
        timer.enabled = with_timer
        if isinstance(trainer.exchange, TimedExchange):
            trainer.exchange.on, trainer.exchange.events = with_timer, []
        # steps that carry the per-kernel HIP events run eagerly (events cannot time kernels inside a replayed graph): every 4th step of
        # an eager run, ONE step of the timed region -- the last -- when the steps are graph replays
        replaying = trainer.captured is not None
        every = timer.every if not replaying else max(steps, 1)
        sample_at = 0 if not replaying else every - 1
        t0 = time.perf_counter()
        for i in range(steps):
            timer.sample = every > 0 and i % every == sample_at
            measuring = with_timer and timer.sample and replaying
            out = trainer(next(it), eager=measuring, on_call=timer.on_call if (measuring and segmented) else None)
        info = Region()
        info.segmented = segmented and trainer.instrumented is not None
        info.instrument_error = trainer.instrument_error
        info.enqueue_s = time.perf_counter() - t0   # host time to enqueue the K steps (the device may still be working)
        fence()
        dt = time.perf_counter() - t0
        timer.enabled = False
        if isinstance(trainer.exchange, TimedExchange):
            trainer.exchange.on = False
        if with_timer:
            timer.sampled_steps = len([i for i in range(steps) if every > 0 and i % every == sample_at])
            timer.every_used = every
        it.close()
        info.D, info.submit_host_s = D, list(loader.submit_host_s)
        return dt, out, info

    for rec in timer.records.values():
        rec.clear()
    dt_local, out, sched = timed(args.prefetch, args.warmup, args.steps, with_timer=True)
    D = sched.D
    if st_prefetcher is not None:
        st_prefetcher.close()
    dt, rank_ms = dt_local, [dt_local / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        dt = max(float(x.item()) for x in allt)   # MAX over ranks
    loss = float(out["loss"].item())

    if rank == 0:
        # points actually processed in the timed steps (sizes differ per batch when --jitter is set; other ranks draw from the
        # same distribution, so the whole-job figure is this rank's count times the world size)
        pts_total = sum(pool_points[(args.warmup + i) % len(pool)] for i in range(args.steps)) * world
        pts_per_step = pts_total / args.steps
        ks = timer.summary()
        traffic, traffic_note = load_traffic()
        if traffic and traffic.get("file"):
            traffic["file_date"] = time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(os.path.join(ROOT, traffic["file"]))))
        steps_of = lambda name: args.steps if name in KernelTimer.ALWAYS else max(timer.sampled_steps, 1)
        cand = [n for n in ks if n.startswith(("attention_step", "dot_prod_with_idx", "window_"))] if strat else list(ks)   # config 5: the window-attention ops
        dom = max(cand or list(ks), key=lambda n: ks[n]["total_ms"] / steps_of(n)) if ks else None

        def roofline_of(name):
            achieved = ks[name]["avg_bytes"] / (ks[name]["avg_ms"] * 1e-3) / 1e9
            r = dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=achieved / HBM_PEAK_GBS, traffic=(traffic.get("host_calls", {}).get(name) if traffic else None),
                     algorithmic_bytes_per_launch=ks[name]["avg_bytes"], avg_launch_ms=ks[name]["avg_ms"],
                     launches_per_step=ks[name]["calls"] / steps_of(name),
                     gpu_time_share_of_step=ks[name]["total_ms"] / steps_of(name) / (dt / args.steps * 1e3),
                     timed_steps=f"{timer.sampled_steps} of {args.steps} (every {getattr(timer, 'every_used', timer.every)}th step of the timed region carries the HIP events"
                                 + (("; those steps replay the SEGMENTED capture -- the same step as a sequence of graphs cut around the timed calls, "
                                     "events between the segments -- the others the one-graph capture)" if getattr(sched, "segmented", False)
                                     else "; those steps run eagerly, the others are graph replays)") if trainer.captured is not None else ")"))
            if ks[name].get("avg_mfma_flops"):   # config 5's table ops: algorithmic flops of the definition next to the bytes
                r["algorithmic_flops_per_launch"] = ks[name]["avg_mfma_flops"]
                r["achieved_TFLOPs_on_algorithmic_flops"] = ks[name]["avg_mfma_flops"] / (ks[name]["avg_ms"] * 1e-3) / 1e12
                r["frac_of_fp32_vector_peak"] = r["achieved_TFLOPs_on_algorithmic_flops"] / FP32_MFMA_PEAK_TFLOPS
            if traffic_note:
                r["traffic_note"] = traffic_note
            elif traffic and r["traffic"] is not None:   # NOT measured in this run: builder-side PMC passes, locked to the kernel sources by hash
                r["traffic_source"] = (f"{traffic.get('file')} ({traffic.get('file_date')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "tools/pmc_only.sh on the builder's gpurun box, 2 x FETCH_SIZE + WRITE_SIZE per launch; kernel-source hash "
                                       f"{traffic.get('kernel_source_hash')} matches this tree")
            return r

        roof = roofline_of(dom) if dom else None
        # the gather family is what the HBM roofline is meaningful for (FPS / kNN are latency / VALU bound by design)
        second = [n for n in ("bottleneck_backward", "bottleneck_forward", "pt_layer_backward", "pt_layer_forward", "group_backward", "group_forward") if n in ks]
        roof2 = roofline_of(max(second, key=lambda n: ks[n]["total_ms"] / steps_of(n))) if second else None
        line = {
            "metric": ("points/sec fwd+bwd (StratifiedTransformer ST-v1m1 + PDF U-decoder)" if strat else
                       "points/sec fwd+bwd (PT-v1 Seg50 + PDF U-decoder, 100k-pt scenes)"),
            "value": pts_per_step * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (f"{args.amp}-autocast (product operands {args.amp}, f32 storage / accumulation)") if args.amp else ("bf16-storage/f32-acc" if args.storage == "bf16" else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{'ScanNet' if scannet else 'S3DIS'}-shaped synthetic voxelised scenes, {args.scenes} x {args.points} points per GPU, "
                                   + ("StratifiedTransformer ST-v1m1 + PointPdf-v1m1 / ST-v1m1-Recognizer, fwd+bwd+SGD, window partition recomputed every step" if strat else
                                      "PointTransformer-Seg50 + PointPdf-v1m1 U-decoder, fwd+bwd+SGD, geometry recomputed every step")
                                   + (", PDF pseudo-label pass inside the step" if args.pseudo_label else ""),
                       "scenes_per_gpu": args.scenes, "points_per_scene": args.points, "size_jitter": args.jitter,
                       **({"size_classes": args.size_classes, "captured_graphs": len(trainer.captures)} if args.size_classes > 0 else {}), "parallelism": f"dp{world}",
                       "gradient_exchange": (args.ddp if use_dp else "none"),
                       **({"shared_gpu": "FUNCTIONAL CHECK ONLY: the ranks share the visible GPU(s) and rendezvous over gloo "
                                         "(PDFOPS_BENCH_SHARED_GPU=1); not a scaling number"} if shared_gpu else {})},
            "per_gpu_points_per_s": pts_per_step * args.steps / dt / world,
            "rccl_ranks": world if (world > 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl") else 0,
            "rank_ms_per_step": rank_ms,
            # what a scaling-efficiency figure computed from `value` is made of: the slowest rank sets the step (barrier + max), the
            # spread says how much of the loss is rank imbalance (host side, clocks), the exchange how much is the all-reduce itself
            "efficiency_breakdown": {"rank_ms_per_step_min": min(rank_ms), "rank_ms_per_step_max": max(rank_ms),
                                     "rank_spread_ms": max(rank_ms) - min(rank_ms),
                                     "exposed_allreduce_ms": (trainer.exchange.mean_ms() if isinstance(trainer.exchange, TimedExchange) else None),
                                     "note": ("exposed_allreduce_ms: HIP events around pack + all-reduce + scale + unpack on the training stream of rank 0, "
                                              "mean over the timed steps; at world size 1 see ddp_overhead_ms") },
            "preflight": pre,
            "loss": loss,
            "execution": ("engine.GroupedGeometryLoader + engine.TrainStep: forward + backward replayed as one captured hipGraph (fixed scene sizes), "
                          "optimizer / gradient exchange / geometry pre-pass eager; the ONE step of the timed region that carries per-kernel HIP events "
                          + ("replays the segmented capture (engine.CapturedStep(split_calls=): graphs cut around the timed calls)" if getattr(sched, "segmented", False)
                             else "runs eagerly" + (f" (segmented capture unavailable: {sched.instrument_error})" if getattr(sched, "instrument_error", None) else ""))
                          if trainer.captured is not None else
                          ("engine.GroupedGeometryLoader + engine.TrainStep, eager (one Python-issued launch sequence per step)"
                           + (f"; graph capture failed: {trainer.capture_error}" if trainer.capture_error else ""))),
            **({"loss_scaling": "dynamic (engine.DeviceGradScaler: torch GradScaler's policy, decisions on the device)"} if scaler is not None else {}),
            "cpu_affinity": affinity,
            "runtime_knobs": __import__("pointcloudpdf_amd").RUNTIME_KNOBS,   # (the ROCm graph-replay switch the package sets at import, and whether it could still take effect)
            "geometry_prefetch_group": (st_ahead if strat else D),
            "host_enqueue_ms_per_step": sched.enqueue_s / args.steps * 1e3,   # < ms_per_step: the host runs ahead, the device is the bound
            "hbm_peak_gib": torch.cuda.max_memory_allocated(dev) / 2.0 ** 30,   # (caching-allocator peak of this rank over the whole run)
            "prepass_submit_host_ms": (1e3 * min(sched.submit_host_s)) if sched.submit_host_s else None,   # host time of one group submission (warm)
            # untimed, before the W warm-up steps: one throw-away group pre-pass per pre-pass stream (allocator pools at their steady state)
            "prepass_allocator_warmup_groups": sched_warm[0],
            "kernels": ks,
            "roofline": roof,
            "roofline_gather_family": roof2,
        }
        if not strat:
            # ---- matrix cores (north_star: "MFMA utilisation against peak"): the Linear layers' FLOPs of the step, 3 x forward (forward,
            # input gradient, weight gradient), against the dense peak of the operand type -- over the whole step (live), over
            # the time of the kernels that run them and as the matrix pipes' counter-measured busy share (builder-side SQ passes)
            sizes = [pool[(args.warmup + i) % len(pool)]["offset_host"] for i in range(args.steps)]
            fl = sum(3.0 * dense_flops_forward(b - a, 9 if scannet else 6, 20 if scannet else 13)
                     for oh in sizes for a, b in zip([0] + list(oh[:-1]), oh)) / args.steps
            peak = BF16_MFMA_PEAK_TFLOPS if args.amp else FP32_MFMA_PEAK_TFLOPS
            ach = fl / (dt / args.steps) / 1e12
            mf = dict(bound="mfma", flops_per_step=fl, flops_definition="3 x sum over every nn.Linear of 2 * rows * in * out (SURVEY.md 8d)",
                      achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                      operands=("fp16/bf16 operands on v_mfma_f32_16x16x16, fp32 accumulate" if args.amp else "fp32 on v_mfma_f32_16x16x4_f32"),
                      note="whole-step figure: the step is HBM / latency bound by design (0.38 MFLOP per point forward), see `roofline`")
            sq = load_sq()
            if sq is not None and not args.amp:
                dk = sq["dense_kernel_ms_per_step"] * 1e-3
                mf.update(dense_kernel_ms_per_step=sq["dense_kernel_ms_per_step"], dense_families_us_per_step=sq["dense_families_us_per_step"],
                          achieved_in_dense_kernels=fl / dk / 1e12, frac_in_dense_kernels=fl / dk / 1e12 / peak,
                          matrix_pipe_busy_share_of_dense_kernel_time=sq["matrix_pipe_busy_share_of_dense_kernel_time"],
                          counters_source=(f"{sq['file']} ({sq['file_date']}): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) per kernel, "
                                           "rocprofv3 --pmc passes of tools/prof_sq.sh on the builder's gpurun box (not measured in this run; "
                                           f"kernel-source hash {sq['kernel_source_hash']} matches this tree)"))
            line["mfma"] = mf
        if traffic and traffic.get("dominant_gpu_kernel"):
            line["dominant_gpu_kernel"] = traffic["dominant_gpu_kernel"]   # the single hottest GPU kernel of the kernel trace (tools/traffic_json.py)
        if traffic and traffic.get("kernels"):
            # the main-stream kernels that cost most per step, with the bandwidth their MEASURED traffic (PMC, profiles/r02_traffic.json)
            # implies: how close each is to the 8 TB/s HBM peak on real bytes (algorithmic bytes: the `roofline` entries above)
            rows = [(v["calls_per_step"] * v["avg_us"], k, v) for k, v in traffic["kernels"].items()
                    if v.get("hbm_bytes_per_launch") and v.get("avg_us") and "k_fps" not in k and "k_grid" not in k and "k_td_tables" not in k]
            line["kernel_traffic_roofline"] = [
                dict(kernel=k.split("(")[0].replace("void ", ""), us_per_step=round(t, 1), launches_per_step=round(v["calls_per_step"], 1),
                     avg_us=round(v["avg_us"], 1), hbm_MB_per_launch=round(v["hbm_bytes_per_launch"] / 1e6, 1),
                     frac_of_hbm_peak=round(v["hbm_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3))
                for t, k, v in sorted(rows, key=lambda r: -r[0])[:12]]

    # ---- latency: how much look-ahead the number above depends on (FPS is a serial chain per scene).  serial = pre-pass inline on
    # the main stream, every step pays the whole FPS chain; then groups of 1 / 2 / 3 batches.
    if world == 1 and not args.no_latency_sweep and not args.pseudo_label and not strat:
        sweep = {}
        for Dl, st, wu in ((0, 4, 1), (1, 6, 2), (2, 6, 2), (3, 6, 3)):
            dtl, _, _ = timed(Dl, wu, st)
            sweep["serial" if Dl == 0 else f"group_{Dl}"] = dtl / st * 1e3
        if trainer.captured is not None:   # the same schedule with the step issued from Python (no graph)
            keep, trainer.captured = trainer.captured, None
            trainer.graph = False
            dtl, _, sch = timed(args.prefetch, 3, args.steps)
            sweep["eager"] = dtl / args.steps * 1e3
            line["eager_ms_per_step"] = sweep["eager"]
            line["eager_host_enqueue_ms_per_step"] = sch.enqueue_s / args.steps * 1e3
            trainer.captured, trainer.graph = keep, True
        # what the data-parallel gradient exchange adds per step, measured at world size 1 over RCCL (pack 609 gradients into the flat
        # buffer, all-reduce 34 MB with itself, scale, unpack): the part of an N-GPU step that is not the ring itself
        try:
            if not torch.distributed.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                torch.distributed.init_process_group("nccl", rank=0, world_size=1)
            gs_keep, fd_keep = trainer.exchange, trainer.force_exchange
            trainer.exchange, trainer.force_exchange = engine.FlatGradAllReduce(step), True
            dta, _, _ = timed(args.prefetch, 3, args.steps)
            trainer.exchange, trainer.force_exchange = gs_keep, fd_keep
            dtb, _, _ = timed(args.prefetch, 3, args.steps)
            line["ddp_overhead_ms"] = (dta - dtb) / args.steps * 1e3
            line["ddp_overhead_note"] = (f"{dta / args.steps * 1e3:.2f} ms per step with the flat exchange forced at world size 1 vs "
                                         f"{dtb / args.steps * 1e3:.2f} ms without, same schedule, back to back")
        except Exception as e:   # noqa: BLE001  (a report, never a reason to lose the line)
            line["ddp_overhead_ms"] = None
            line["ddp_overhead_note"] = f"{type(e).__name__}: {e}"
        line["serial_ms_per_step"] = sweep["serial"]
        line["lookahead_sweep_ms_per_step"] = sweep

    if rank == 0:
        if world == 1 and not args.no_ops_roofline and not scannet and not strat:
            # the pointops drop-in ops on their own (level-1 shapes of this config: 200k points, c = 32, k = 8), HIP-event timed on
            # the launching stream, against the 8 TB/s HBM peak with the SURVEY 8(d) byte counts (tools/ops_roofline.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ops_roofline
            rows = ops_roofline.run(iters=10, level2=False, references=True)
            # "also report against a measured stream-copy ceiling from the same run" -- a 256 MB device-to-device copy
            # (read + write = 512 MB) timed the same way; `frac_of_stream_copy` = the op's algorithmic GB/s / that copy's GB/s
            copy = next((r["GBps"] for r in rows if r["op"].startswith("copy 256 MB")), None)
            line["stream_copy_GBps"] = None if copy is None else round(copy, 1)
            line["roofline_ops"] = [dict(op=r["op"], us=round(r["us"], 1), GBps=round(r["GBps"], 1), frac=round(r["frac"], 4),
                                         **({"frac_of_stream_copy": round(r["GBps"] / copy, 3)} if copy and "bruteforce_pairs" not in r else {}),
                                         **({"evaluated_pairs": r["evaluated_pairs"], "pair_evals_per_s": float(f"{r['pair_evals_per_s']:.4g}"),
                                             "valu_frac_evaluated": round(r["valu_frac_evaluated"], 4),   # evaluated candidate distances x 8 flop / time / 157.3 TFLOP/s
                                             "pruning_factor_vs_bruteforce": round(r["pruning_factor"], 1)}
                                            if "pair_evals_per_s" in r else {}))
                                    for r in rows]
        if world == 1 and not args.no_cpu_baseline and not scannet and not strat:   # (the CPU baseline is quoted on the headline workload)
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
