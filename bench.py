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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=6)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops-roofline", action="store_true", help="skip the per-op HBM roofline micro-benchmark (rank 0, N = 1)")
    ap.add_argument("--cpu-points", type=int, default=100000, help="scene size of the bounded CPU-baseline sample")
    ap.add_argument("--st-group", type=int, default=3, help="--workload stratified: batches per grouped coordinate pre-pass (1 = one batch ahead)")
    ap.add_argument("--amp", nargs="?", const="f16", default=None, choices=["bf16", "f16"],
                    help="torch.autocast around the step (reference: enable_amp = True, engines/train.py:340-363): the streaming Linear products "
                         "run with fp16 (default, as the reference's autocast) / bfloat16 operands on the 16x16x16 matrix-core instructions, fp32 storage and accumulation "
                         "(dense.fp32_path).  f16 adds a static loss scale of 4096 for the backward (the reference uses a GradScaler)")
    ap.add_argument("--storage", choices=["f32", "bf16"], default=os.environ.get("PDFOPS_STORAGE", "f32"),
                    help="bf16: the reduced-precision variant -- the fused PointTransformerLayer keeps its saved / scratch row arrays (H, G2, "
                         "softmax weights, g_r rows) as bfloat16 with fp32 accumulation (the reference trains under AMP); the headline stays f32")
    ap.add_argument("--prefetch", type=int, default=16,
                    help="geometry pre-pass group: the pre-pass of the NEXT `prefetch` batches runs as one launch sequence on a side "
                         "stream while the current group trains (0 = inline, serial)")
    ap.add_argument("--graph", choices=["auto", "0", "1"], default="auto",
                    help="1 = forward + backward of the step replayed as ONE captured hipGraph (engine.CapturedStep: batch tensors and the "
                         "batch's geometry tables staged into fixed-address buffers by one copy launch per step); the optimizer, the "
                         "gradient exchange and the geometry pre-pass stay eager.  Needs identical scene sizes in every batch (what "
                         "SphereCrop(point_max) gives the reference's trainer): auto = on unless --jitter / --amp / --pseudo-label / "
                         "--throttle / --ddp torch / the stratified workload ask for something the capture does not cover")
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
        if name.startswith(("attention_step", "dot_prod_with_idx")):
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
        """fp32 MFMA work of the window-attention table-gradient kernels (one-hot (L x edges) . (edges x d) products per head and axis,
        csrc/window_attention.hip): 2 * 3 axes * L * M * C per table gradient (two tables in dot_prod_with_idx_v3's backward)."""
        if name == "dot_prod_with_idx_v3_backward":
            q, table = args[1], args[6]
            m, c, L = args[5].shape[0], q.shape[1] * q.shape[2], table.shape[0]
            return 2.0 * 2 * 3 * L * m * c
        if name == "attention_step2_with_rel_pos_value_v2_backward":
            v, table = args[2], args[6]
            m, c, L = args[1].shape[0], v.shape[1] * v.shape[2], table.shape[0]
            return 2.0 * 3 * L * m * c
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
    if args.optimizer == "fused":   # one launch over all 304 tensors (csrc/optim.hip); "torch" = torch.optim.SGD(fused=True), 13 launches
        opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    else:
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
    names = ["knn_query", "farthest_point_sampling", "group_forward", "group_backward",
             "pt_layer_forward", "pt_layer_backward", "bottleneck_forward", "bottleneck_backward"]
    if strat:   # config 5: the libs/pointops2 window-attention ops (csrc/window_attention.hip)
        names += ["attention_step1_v2", "attention_step1_v2_backward", "dot_prod_with_idx_v3", "dot_prod_with_idx_v3_backward",
                  "attention_step2_with_rel_pos_value_v2", "attention_step2_with_rel_pos_value_v2_backward"]
    timer = KernelTimer(be, names)
    timer.install()

    from pointcloudpdf_amd.geometry import GeometryPrefetcher

    # two side streams, alternating groups.  PDFOPS_PREPASS_THREAD=1 builds the pre-pass on a worker thread (measured: no gain -- the worker's
    # Python / dispatch work competes with the training thread for the interpreter: 17.7-18.6 vs 18.2-18.4 ms per step)
    prefetcher = GeometryPrefetcher(depth=2, threaded=bool(os.environ.get("PDFOPS_PREPASS_THREAD")))

    amp_dtype = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[args.amp]
    loss_scale = 4096.0 if args.amp == "f16" else 1.0
    graph_ok = (args.jitter <= 0 and not args.pseudo_label and not args.throttle and not strat
                and not (use_dp and args.ddp == "torch"))
    if args.graph == "1" and not graph_ok:
        raise SystemExit("bench.py: --graph 1 needs fixed scene sizes and the plain f32 PointTransformer step (see --help)")
    captured, capture_note, mode_calibration = None, None, None   # (the execution mode is settled below, once `timed` exists)

    class Schedule:
        """Grouped geometry pre-pass for `warmup + steps` steps: the pre-pass of the next D batches runs as ONE launch sequence on a
        side stream (FPS is a chain of dependent arg-max steps, one workgroup per scene: its latency is amortised over the group
        instead of being paid per step).  Group boundaries sit at warmup + k * D, so the timed region queues exactly steps / D group
        pre-passes -- every trained batch gets exactly one full pre-pass, nothing is cached -- and, D steps being a whole group's
        lead time, drains them before the closing fence.  D = 0: the pre-pass runs inline on the main stream (serial step)."""

        def __init__(self, D, warmup, steps):
            if D > 0:
                D = max(1, min(D, 64 // max(args.scenes, 1)))   # the grid kNN / radius workspaces hold <= 64 scenes per call
                if steps <= D:
                    D = max(steps, 1)
                else:
                    divs = [d for d in range(min(8, D), D + 1) if steps % d == 0]
                    D = max(divs) if divs else D
            self.D, self.phase = D, (warmup % D if D > 0 else 0)
            self.tickets, self.submit_host_s, self.pending, self.step_index = {}, [], None, 0

        def _submit_range(self, lo, hi, ready=None):
            group = [pool[j % len(pool)] for j in range(lo, hi)]
            t_sub = time.perf_counter()
            for j, t in enumerate(prefetcher.submit_group(group, ready=ready)):
                self.tickets[lo + j] = t
            self.submit_host_s.append(time.perf_counter() - t_sub)

        def start(self):
            if self.D > 0:
                b0 = self.phase if self.phase > 0 else self.D
                self._submit_range(0, b0)
                if self.phase > 0:
                    self._submit_range(b0, b0 + self.D)

        def geometry(self, i):
            """Tables of step i's batch (queued with its group); queues the NEXT group when i crosses a boundary."""
            if self.D == 0:
                return None
            geom = prefetcher.get(self.tickets.pop(i))
            if (i + self.D - self.phase) % self.D == 0:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream())     # the pre-pass depends on what is queued up to HERE, not on this step
                # submitted by after_step() once a few steps are queued behind the boundary (the group has D steps of lead: its pre-pass
                # is due D steps from here): on a slow host the submission takes longer than the one step queued so far
                # (replayed steps only: issued from Python, two steps of host time are ~30 ms of the group's lead)
                self.pending = (i + self.D, i + 2 * self.D, ready, i + (min(2, self.D - 1) if captured is not None else 0))
            self.step_index = i
            return geom

        def after_step(self):
            """Queue the next group's pre-pass AFTER the current step has been enqueued: its submission is 9-15 ms of host work, and
            issued in front of the step (rounds 1-3) it left the device idle for that long at every group boundary -- with replayed steps
            the device queue is empty right after the opening fence of a timed region."""
            if self.pending is not None and self.step_index >= self.pending[3]:
                lo, hi, ready, _ = self.pending
                self.pending = None
                self._submit_range(lo, hi, ready)

        def drain(self):
            """Pre-passes queued beyond the last step (none when steps is a multiple of D): wait for them, drop them."""
            self.pending = None   # (a group that would only be needed after the last step)
            for t in self.tickets.values():
                prefetcher.get(t)
            self.tickets.clear()

    st_prefetcher, st_tickets, st_state = None, {}, {"next": 0}
    if st_ahead:
        from pointcloudpdf_amd.stratified import StratifiedPrefetcher
        st_prefetcher = StratifiedPrefetcher(step.model.backbone, windows=not os.environ.get("PDFOPS_ST_FPS_ONLY"))

    def one_step(i, sched):
        batch = pool[i % len(pool)]
        data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"],
                    offset_host=batch["offset_host"], segment=batch["segment"])
        geom = sched.geometry(i)
        if geom is not None:
            data["pdf_geometry"] = geom
        if st_prefetcher is not None:
            G = st_ahead

            def st_submit(lo, hi):   # one group: batches lo .. hi - 1
                for j, t in enumerate(st_prefetcher.submit_group([pool[x % len(pool)] for x in range(lo, hi)])):
                    st_tickets[lo + j] = t
            if i not in st_tickets:                      # the first step of a run
                st_submit(i, i + G)
                st_state["next"] = i + G
            if i + G >= st_state["next"]:                # the group after the one in use: built while this one trains
                st_submit(st_state["next"], st_state["next"] + G)
                st_state["next"] += G
            data["st_geometry"] = st_prefetcher.get(st_tickets.pop(i))
        replay = captured is not None and not (timer.enabled and timer.sample) and captured.matches(batch)
        if replay:   # forward + backward as one hipGraph launch; steps that carry the per-kernel HIP events run eagerly
            if geom is None:   # --prefetch 0: the pre-pass (and its packing) inline on this stream
                from pointcloudpdf_amd.geometry import Geometry
                geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
            out = captured(batch, geom)
        else:
            # An eagerly issued step between replays (the one that carries the per-kernel HIP events) runs on the capture's stream: the
            # parameters' AccumulateGrad nodes are bound to it while the captured autograd graph lives, and on any other stream the
            # engine synchronises every one of the 609 of them (20-40 ms for the step instead of 16.5).
            cur = torch.cuda.current_stream()
            run_on = captured.stream if captured is not None else cur
            if run_on is not cur:
                run_on.wait_stream(cur)
            with torch.cuda.stream(run_on):
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None):
                    out = module(data)
                if loss_scale == 1.0:
                    out["loss"].backward()
                else:
                    (out["loss"] * loss_scale).backward()
                    with torch.no_grad():
                        torch._foreach_mul_([p.grad for p in step.parameters() if p.grad is not None], 1.0 / loss_scale)
            if run_on is not cur:
                cur.wait_stream(run_on)
        if grad_sync is not None:
            grad_sync.sync(force=force_dp)   # ONE all-reduce (RCCL) over the flat gradient buffer
        opt.step()
        sched.after_step()
        return out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(D, warmup, steps, with_timer=False):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize pairs -> (seconds, last output, schedule)."""
        sched = Schedule(D, warmup, steps)
        sched.start()
        for i in range(warmup):
            one_step(i, sched)
        fence()
        timer.enabled = with_timer
        # steps that carry the per-kernel HIP events run eagerly (events cannot time kernels inside a replayed graph): every 4th step of
        # an eager run, ONE step of the timed region when the steps are graph replays -- the LAST one: issuing a step from Python takes
        # the host longer than the device needs to run it, which costs nothing behind the backlog of replays queued before it and
        # 4-12 ms of idle device at the front of the region, right after the fence (where rounds 1-3 had it)
        every = timer.every if captured is None else max(steps, 1)
        sample_at = 0 if captured is None else every - 1
        t0 = time.perf_counter()
        for i in range(steps):
            timer.sample = every > 0 and i % every == sample_at
            out = one_step(warmup + i, sched)
        sched.enqueue_s = time.perf_counter() - t0   # host time to enqueue the K steps (the device may still be working)
        fence()
        dt = time.perf_counter() - t0
        timer.enabled = False
        if with_timer:
            timer.sampled_steps = len([i for i in range(steps) if every > 0 and i % every == sample_at])
            timer.every_used = every
        sched.drain()
        for t in st_tickets.values():
            st_prefetcher.get(t)
        st_tickets.clear()
        return dt, out, sched

    # ---- execution mode of the step (set-up, like building the model): forward + backward replayed as one captured hipGraph, or every
    # launch issued from Python.  Replay takes the host off the critical path (1.7 ms of host work per step instead of ~15-20) but costs
    # the device ~0.6 ms per step (staging copy into the fixed-address buffers, graph-launch bookkeeping); where the host keeps ahead of
    # the device anyway the eager step is the faster one (16.2 vs 16.9 ms on the fast hosts of the pool, 19.4 vs 17 on a slow one).
    # --graph auto times a region of the timed region's size of each on THIS host before the warm-up and keeps the faster (eager first: an autograd graph that
    # survives a capture binds the parameters' AccumulateGrad nodes to the capture stream and slows later eager steps).
    if args.graph == "1" or (args.graph == "auto" and graph_ok):
        # (one rank only: with N ranks the replayed step is kept -- N Python processes share the host, and every rank must pass the same
        #  number of barriers)
        cal = max(args.steps, 1)   # (a region of the size of the timed one: the same share of pre-pass submissions and event-carrying steps)
        dt_eager = timed(args.prefetch, 3, cal, with_timer=True)[0] / cal if (args.graph == "auto" and world == 1) else None
        try:
            captured = engine.CapturedStep(step, pool[0], autocast=amp_dtype, loss_scale=loss_scale)
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001  (auto: a stack that cannot capture the step still gets its line, on the eager path)
            if args.graph == "1":
                raise
            captured, capture_note = None, f"graph capture failed ({type(e).__name__}: {e}); eager path"
            print(f"bench.py: {capture_note}", file=sys.stderr)
        if captured is not None and dt_eager is not None:
            dt_graph = timed(args.prefetch, 3, cal, with_timer=True)[0] / cal
            mode_calibration = {"eager_ms_per_step": dt_eager * 1e3, "graph_ms_per_step": dt_graph * 1e3, "steps_each": cal}
            if dt_eager < 0.97 * dt_graph:
                captured = None
        if captured is None:   # (dropped or failed: no autograd state of the capture may outlive it)
            import gc
            torch.cuda.synchronize()
            engine.release_autograd_state(step)
            for p_ in step.parameters():
                p_.grad = None
            gc.collect()
            torch.cuda.empty_cache()
    for rec in timer.records.values():   # (the calibration regions carried kernel events like the timed one will: only the timed region's count)
        rec.clear()
    dt_local, out, sched = timed(args.prefetch, args.warmup, args.steps, with_timer=True)
    D = sched.D
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
        steps_of = lambda name: args.steps if name in KernelTimer.ALWAYS else max(timer.sampled_steps, 1)
        cand = [n for n in ks if n.startswith(("attention_step", "dot_prod_with_idx"))] if strat else list(ks)   # config 5: the window-attention ops
        dom = max(cand or list(ks), key=lambda n: ks[n]["total_ms"] / steps_of(n)) if ks else None

        def roofline_of(name):
            if ks[name].get("avg_mfma_flops"):   # MFMA-bound kernel: fp32 one-hot products against the dense fp32 matrix peak
                ach = ks[name]["avg_mfma_flops"] / (ks[name]["avg_ms"] * 1e-3) / 1e12
                return dict(bound="mfma", kernel=name, achieved=ach, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP32_MFMA_PEAK_TFLOPS,
                            traffic=None, flops_per_launch=ks[name]["avg_mfma_flops"], algorithmic_bytes_per_launch=ks[name]["avg_bytes"],
                            hbm_GBps_on_algorithmic_bytes=ks[name]["avg_bytes"] / (ks[name]["avg_ms"] * 1e-3) / 1e9,
                            avg_launch_ms=ks[name]["avg_ms"], launches_per_step=ks[name]["calls"] / steps_of(name),
                            gpu_time_share_of_step=ks[name]["total_ms"] / steps_of(name) / (dt / args.steps * 1e3))
            achieved = ks[name]["avg_bytes"] / (ks[name]["avg_ms"] * 1e-3) / 1e9
            r = dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=achieved / HBM_PEAK_GBS, traffic=(traffic.get("host_calls", {}).get(name) if traffic else None),
                     algorithmic_bytes_per_launch=ks[name]["avg_bytes"], avg_launch_ms=ks[name]["avg_ms"],
                     launches_per_step=ks[name]["calls"] / steps_of(name),
                     gpu_time_share_of_step=ks[name]["total_ms"] / steps_of(name) / (dt / args.steps * 1e3),
                     timed_steps=f"{timer.sampled_steps} of {args.steps} (every {getattr(timer, 'every_used', timer.every)}th step of the timed region carries the HIP events"
                                 + ("; those steps run eagerly, the others are graph replays)" if captured is not None else ")"))
            if traffic_note:
                r["traffic_note"] = traffic_note
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
                       "scenes_per_gpu": args.scenes, "points_per_scene": args.points, "size_jitter": args.jitter, "parallelism": f"dp{world}",
                       "gradient_exchange": (args.ddp if use_dp else "none"),
                       **({"shared_gpu": "FUNCTIONAL CHECK ONLY: the ranks share the visible GPU(s) and rendezvous over gloo "
                                         "(PDFOPS_BENCH_SHARED_GPU=1); not a scaling number"} if shared_gpu else {})},
            "per_gpu_points_per_s": pts_per_step * args.steps / dt / world,
            "rccl_ranks": world if (world > 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl") else 0,
            "rank_ms_per_step": rank_ms,
            "loss": loss,
            "execution": ("forward + backward replayed as one captured hipGraph (engine.CapturedStep; fixed scene sizes), optimizer / gradient "
                          "exchange / geometry pre-pass eager; the steps that carry per-kernel HIP events run eagerly" if captured is not None
                          else (capture_note or "eager (one Python-issued launch sequence per step)")),
            **({"execution_calibration": mode_calibration} if mode_calibration else {}),
            "cpu_affinity": affinity,
            "geometry_prefetch_group": (st_ahead if strat else D),
            "host_enqueue_ms_per_step": sched.enqueue_s / args.steps * 1e3,   # < ms_per_step: the host runs ahead, the device is the bound
            "hbm_peak_gib": torch.cuda.max_memory_allocated(dev) / 2.0 ** 30,   # (caching-allocator peak of this rank over the whole run)
            "prepass_submit_host_ms": (1e3 * min(sched.submit_host_s)) if sched.submit_host_s else None,   # host time of one group submission (warm)
            "kernels": ks,
            "roofline": roof,
            "roofline_gather_family": roof2,
        }
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
        if captured is not None:   # the same schedule with the step issued from Python (no graph)
            keep, captured = captured, None
            dtl, _, sch = timed(args.prefetch, 3, args.steps)
            sweep["eager"] = dtl / args.steps * 1e3
            line["eager_ms_per_step"] = sweep["eager"]
            line["eager_host_enqueue_ms_per_step"] = sch.enqueue_s / args.steps * 1e3
            captured = keep
        # what the data-parallel gradient exchange adds per step, measured at world size 1 over RCCL (pack 609 gradients into the flat
        # buffer, all-reduce 34 MB with itself, scale, unpack): the part of an N-GPU step that is not the ring itself
        try:
            if not torch.distributed.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                torch.distributed.init_process_group("nccl", rank=0, world_size=1)
            gs_keep, fd_keep = grad_sync, force_dp
            grad_sync, force_dp = engine.FlatGradAllReduce(step), True
            dta, _, _ = timed(args.prefetch, 3, args.steps)
            grad_sync, force_dp = gs_keep, fd_keep
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
                                         **({"frac_of_stream_copy": round(r["GBps"] / copy, 3)} if copy and "pair_evals_per_s" not in r else {}),
                                         **({"pair_evals_per_s": float(f"{r['pair_evals_per_s']:.4g}"),
                                             "valu_frac_bruteforce_equivalent": round(r["valu_frac_bruteforce_equivalent"], 4)}
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
