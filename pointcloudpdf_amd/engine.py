"""One open-world training step of the hot path, as a single module (so ONE DistributedDataParallel reducer covers
segmentor + U-decoder; the reference wraps them separately: pointcept/engines/train.py:218-222 and :419-423).

``OpenSegStep.forward(batch)`` == ``OpenSegTrainer.model_forward`` (engines/train.py:373-380): label_rename
(segment := segment_known, :387-391), segmentor forward + CE, recognizer forward (+ PDF loss), summed loss.
Whole scenes are the sharding unit: each rank runs its own scenes, gradients are the only thing exchanged
(DDP all-reduce over RCCL/xGMI), BatchNorm statistics stay per rank (sync_bn=False, broadcast_buffers=False upstream).
"""
import os

import torch

from . import dense
import torch.nn as nn

from . import point_transformer, recognizer, segmentor  # noqa: F401  (registers the classes)
from .model_hook import BaseModelHook
from .registry import MODELS, RECOGNIZER

PT_V1_HOOKS = {  # configs/s3dis/openseg-pt-v1-0-pointpdf-v1m1-base.py:11-27
    **{f"backbone.enc{i}": ["forward_output"] for i in range(1, 6)},
    **{f"backbone.dec{i}.1": ["forward_output"] for i in range(1, 6)},
    "backbone": ["forward_output"],
}


def default_pseudo_mask(coord, seg_logits, offset):
    """Stand-in for the PDF pseudo-label pass (scope row f-2): a fixed 1-in-7 pattern."""
    return (torch.arange(coord.shape[0], device=coord.device) % 7) == 3


ST_V1M1_HOOKS = {  # configs/s3dis/openseg-st-v1m1-0-origin-pointpdf-v1m1-base.py:41-51 (the non-existent "backbone.upsamples.3" left out)
    **{f"backbone.upsamples.{i}": ["forward_input", "forward_output"] for i in range(3)},
    "backbone": ["forward_output"],
}
ST_V1M1_BACKBONE = dict(  # configs/s3dis/openseg-st-v1m1-0-origin-pointpdf-v1m1-base.py:13-38
    type="ST-v1m1", downsample_scale=8, depths=[2, 2, 6, 2], channels=[48, 96, 192, 384], num_heads=[3, 6, 12, 24],
    window_size=[0.16, 0.32, 0.64, 1.28], up_k=3, grid_sizes=[0.04, 0.08, 0.16, 0.32], quant_sizes=[0.01, 0.02, 0.04, 0.08],
    rel_query=True, rel_key=True, rel_value=True, drop_path_rate=0.3, num_layers=4, concat_xyz=True, ratio=0.25, k=16,
    prev_grid_size=0.04, sigma=1.0, stem_transformer=True, kp_ball_radius=0.04 * 2.5, kp_max_neighbor=34)


class OpenSegStep(nn.Module):
    def __init__(self, backbone="PointTransformer-Seg50", in_channels=6, num_classes=13, loss_weight=0.1,
                 start_epoch=0, pseudo_mask_fn=default_pseudo_mask):
        """``backbone``: a registered PointTransformer-Seg* name (PT-v1 + PDF U-decoder, BASELINE configs 2-4) or "ST-v1m1"
        (StratifiedTransformer + ST-v1m1-Recognizer with the reference's S3DIS settings, BASELINE config 5)."""
        super().__init__()
        from . import stratified  # noqa: F401  (registers ST-v1m1 / ST-v1m1-Recognizer)

        ce = [dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)]
        if backbone == "ST-v1m1":
            bb = dict(ST_V1M1_BACKBONE, num_classes=num_classes)
            rec, hooks = dict(type="ST-v1m1-Recognizer", up_k=3, channels=bb["channels"], num_layers=4), ST_V1M1_HOOKS
        else:
            bb = dict(type=backbone, in_channels=in_channels, num_classes=num_classes)
            rec, hooks = dict(type="PointTransformer-Recognizer"), PT_V1_HOOKS
        self.model = MODELS.build(dict(type="DefaultSegmentor", backbone=bb, criteria=ce))
        self.recognizer = RECOGNIZER.build(dict(type="PointPdf-v1m1", recognizer=rec,
                                                criteria=ce, loss_weight=loss_weight, step_loss_weight=False,
                                                num_classes=num_classes, start_epoch=start_epoch,
                                                pseudo_mask_fn=pseudo_mask_fn))
        self.hooks = BaseModelHook(hooks, clone_tensor=True, exclude_clone={"backbone": ["forward_output"]})
        self.hooks.set_model(self.model)
        self.recognizer.model_hooks = self.hooks
        self.recognizer.set_epoch(start_epoch)
        self.recognizer.trigger_operation()  # release the U-decoder parameters before DDP sees them

    def forward(self, batch):
        input_dict = dict(batch)
        if "segment_known" in input_dict:
            input_dict["segment"] = input_dict["segment_known"]
        with self.hooks, dense.deferred_counters():   # BatchNorm step counters: one multi-tensor add per step
            out = self.model(input_dict)
            rec = self.recognizer(input_dict)
        loss = out["loss"]
        if "loss" in rec:
            loss = loss + rec["loss"]
        return dict(loss=loss, model_loss=out["loss"].detach(), recognizer_loss=rec.get("loss", loss.new_zeros(())).detach(),
                    score=rec["score"].detach())


def release_autograd_state(step):
    """Drop every reference an ``OpenSegStep`` keeps to the last step's autograd graph (the hook tap's captured tensors).  A live graph
    keeps its AccumulateGrad nodes -- and the stream they were created on -- alive; needed before a step is captured into a hipGraph on
    another stream (``CapturedStep``), harmless otherwise."""
    for per_module in step.hooks.output.values():
        for key in per_module:
            per_module[key] = None


class CapturedStep:
    """Forward + backward of an ``OpenSegStep`` captured ONCE into a hipGraph and replayed per batch (the step is ~1,200 dependent
    launches: issued from Python it is host-bound, replayed it costs the host one call).  Everything the captured kernels read lives
    at fixed addresses: the batch tensors (``coord``, ``feat``, ``offset``, ``segment``: static copies) and the batch's coordinate-only
    tables (``geometry.StaticGeometry``: one flat buffer), both filled by ONE staging launch per step that reads the pre-pass's tensors
    where they lie (csrc/stage_copy.hip).
    The graph is specific to the scene sizes it was captured with -- what ``SphereCrop(point_max)`` hands the reference's trainer for
    every scene above the limit; batches of another shape run the eager path (``matches``).

    The optimizer and the data-parallel gradient exchange stay outside the graph (one launch / one all-reduce): parameters' ``.grad``
    are the graph's own static tensors (re-bound after every replay, so an eager step in between does no harm).  Python-side scalars
    of the step (``PointPdfV1.alpha``, ``epoch`` gates, BatchNorm momenta) are baked in at capture: re-capture when they change."""

    KEYS = ("coord", "feat", "offset", "segment")

    def __init__(self, step, batch, geom=None, warmup=2, autocast=None, loss_scale=1.0):
        """``autocast``: torch.float16 / torch.bfloat16 -> the forward is captured under torch.autocast (the path then runs its
        reduced-precision products, dense.fp32_path); ``loss_scale``: static scale of the loss for the backward (fp16 operands: gradients of
        ~1e-6 are below fp16's normal range), divided out of the gradients again inside the graph."""
        from .geometry import Geometry, StaticGeometry

        self.autocast, self.loss_scale = autocast, float(loss_scale)

        assert step.training, "CapturedStep captures a TRAINING step (forward + backward)"
        self.step = step
        dev = batch["coord"].device
        self.sizes = [int(v) for v in batch["offset_host"]]
        self.static = {k: batch[k].clone() for k in self.KEYS}
        self.static["offset_host"] = list(self.sizes)
        if geom is None:
            geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
        self.geometry = StaticGeometry(geom)
        self.layout = self.geometry.layout
        self.params = [p for p in step.parameters() if p.requires_grad]
        # BatchNorm buffers / step counters move with every forward: the warm-up passes below must not count as training steps
        buffers = [b for b in step.buffers()]
        saved = [b.detach().clone() for b in buffers]
        release_autograd_state(step)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):   # lazily created handles / caches must exist before the capture
                for p in self.params:
                    p.grad = None
                self._backward(self._eager())
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for p in self.params:
            p.grad = None
        release_autograd_state(step)
        with torch.no_grad():
            for b, v in zip(buffers, saved):
                b.copy_(v)
        # Capture on the SAME side stream the warm-up ran on: a parameter's AccumulateGrad node remembers the stream it was created on, and a
        # node that survived the warm-up on another stream makes the engine fork the capture onto that stream -- work and allocations of
        # the fork are then outside the graph's private pool (observed: replays that read recycled memory once eager work ran in between).
        self.stream = side   # an EAGER step of the same module between replays is fastest on this stream (see the comment above: the
        #                      parameters' AccumulateGrad nodes stay bound to it for as long as the captured autograd graph lives)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.out = self._eager()
            self._backward(self.out)
        self.grads = [p.grad for p in self.params]
        with torch.no_grad():   # (the capture itself does not execute anything, but keep the contract obvious)
            for b, v in zip(buffers, saved):
                b.copy_(v)
        release_autograd_state(step)

    def _eager(self):
        with torch.autocast("cuda", dtype=self.autocast or torch.float16, enabled=self.autocast is not None):
            return self.step(dict(self.static, pdf_geometry=self.geometry))

    def _backward(self, out):
        if self.loss_scale == 1.0:
            out["loss"].backward()
        else:
            (out["loss"] * self.loss_scale).backward()
            with torch.no_grad():
                torch._foreach_mul_([p.grad for p in self.params if p.grad is not None], 1.0 / self.loss_scale)

    def matches(self, batch):
        return [int(v) for v in batch["offset_host"]] == self.sizes and all(batch[k].shape == self.static[k].shape for k in self.KEYS)

    @torch.no_grad()
    def __call__(self, batch, geom):
        """One training step's forward + backward on ``batch`` with the coordinate-only tables ``geom`` (any Geometry of the batch: its own
        pre-pass or its share of a grouped one).  Two launches: the staging copy (batch tensors + ~70 tables into the fixed-address
        buffers, csrc/stage_copy.hip) and the graph.  Returns the static output dict (``loss``, ``model_loss``, ``recognizer_loss``,
        ``score``: overwritten by the next call); gradients are in ``p.grad``."""
        self.geometry.stage(geom, extra=[(batch[k], self.static[k]) for k in self.KEYS])
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.out


def wrap_ddp(module, device):
    """DDP over RCCL: one flat bucket (34 MB of fp32 gradients), bucket views, per-rank BN buffers."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    if device.type == "cuda":
        return DDP(module, device_ids=[device.index], broadcast_buffers=False, bucket_cap_mb=64,
                   gradient_as_bucket_view=True)
    return DDP(module, broadcast_buffers=False, bucket_cap_mb=64, gradient_as_bucket_view=True)


class FlatGradAllReduce:
    """The data-parallel gradient exchange as ONE all-reduce over ONE flat fp32 buffer (7,767,729 + 792,513 parameters =
    34.3 MB), fired after the backward pass: gradients are packed with a multi-tensor copy, summed over the ranks (RCCL ring
    over xGMI: ~0.5 ms at 8 GPUs), scaled by 1/world and unpacked with a second multi-tensor copy -- about ten launches per
    step.  torch's DistributedDataParallel gives the same result (``wrap_ddp``; upstream wraps its models in it,
    engines/defaults.py:22-43) but its reducer copies every one of the 609 per-parameter gradients into the bucket with its
    own kernel: +3 ms on a 22 ms step, measured at world size 1.  Parameters are broadcast from rank 0 at construction, buffers
    (BatchNorm statistics) stay per rank as upstream (``broadcast_buffers=False``, engines/train.py:220)."""

    def __init__(self, module, broadcast_parameters=True):
        import torch.distributed as dist

        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if broadcast_parameters and self.world > 1:
            with torch.no_grad():
                torch._foreach_copy_(self.views, [p.data for p in self.params])
                dist.broadcast(self.flat, 0)
                torch._foreach_copy_([p.data for p in self.params], self.views)

    @torch.no_grad()
    def sync(self, force=False):
        """Call between ``loss.backward()`` and ``optimizer.step()``: every ``p.grad`` becomes the mean over the ranks.
        ``force`` runs the pack / all-reduce / unpack sequence at world size 1 too (overhead measurements)."""
        if self.world == 1 and not (force and self.dist.is_initialized()):
            return
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        if len(have) != len(self.params):
            self.flat.zero_()   # a parameter without a local gradient contributes zeros
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self.dist.all_reduce(self.flat)
        self.flat.mul_(1.0 / self.world)
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                p.grad = v.clone()
        torch._foreach_copy_([g for _, g in have], [v for v, _ in have])


class FusedSGD(torch.optim.Optimizer):
    """``torch.optim.SGD(params, lr, momentum, weight_decay)`` (dampening 0, no Nesterov: what the reference's configs build,
    pointcept/utils/optimizer.py) as ONE HIP launch per parameter group and step over all of the group's tensors (csrc/optim.hip).
    torch's fused multi-tensor SGD needs 13 launches / 275 us for this model's 304 tensors; this is one launch / ~30 us.  Same
    arithmetic per element (``g + wd p``, ``momentum buf + g'``, ``p - lr buf``; the first step's ``buf = g'`` is the zero-initialised
    buffer's update).  Parameters without a gradient are skipped, as torch does.

    A ``torch.optim.Optimizer``: ``param_groups`` (``lr`` / ``momentum`` / ``weight_decay`` are read from the group at every step, so
    torch's LR schedulers attach to it -- the reference steps its scheduler every iteration, engines/train.py:366), ``state_dict`` /
    ``load_state_dict`` with torch.optim.SGD's own layout (``state[p]["momentum_buffer"]``: checkpoints move both ways).  Parameter and
    momentum pointers are read at every step (``model.to()``, ``load_state_dict`` may move them)."""

    RING = 8   # pinned pointer tables in flight (the host may run several steps ahead of the device)

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0):
        import ctypes
        from . import _native

        self.be = _native.hip_backend()
        self.ctypes = ctypes
        super().__init__(params, dict(lr=float(lr), momentum=float(momentum), weight_decay=float(weight_decay)))
        for group in self.param_groups:
            group["params"] = [p for p in group["params"] if p.requires_grad]
        every = [p for group in self.param_groups for p in group["params"]]
        assert every and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in every)
        self.device = every[0].device
        for p in every:
            self.state[p]["momentum_buffer"] = torch.zeros_like(p)
        self.chunk = int(self.be.lib.pdf_sgd_chunk())
        self._plans = {}      # (group, tuple of parameter indices with a gradient) -> (chunk list on the device, number of chunks, lengths)
        rows = max(len(g["params"]) for g in self.param_groups)
        self._ring = [(torch.empty((rows, 4), dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(self.RING)]
        self._tabs = [torch.empty((rows, 4), dtype=torch.int64, device=self.device) for _ in range(self.RING)]
        self._n = 0
        self._captured = []   # pointer tables of captured steps (hipGraph replays read them)
        self._spare = []
        self.reserve_capture_tables(2 * len(self.param_groups))

    def reserve_capture_tables(self, n):
        """Pinned pointer tables for ``n`` more (group, captured step) pairs; must be called outside stream capture."""
        rows = max(len(g["params"]) for g in self.param_groups)
        self._spare += [torch.empty((rows, 4), dtype=torch.int64).pin_memory() for _ in range(n)]

    @property
    def params(self):
        return [p for group in self.param_groups for p in group["params"]]

    def _plan(self, gi, have, params):
        key = (gi, have)
        if key not in self._plans:
            import numpy as np

            pairs = [(row, c) for row, i in enumerate(have) for c in range((params[i].numel() + self.chunk - 1) // self.chunk)]
            lengths = np.array([params[i].numel() for i in have], dtype=np.int64)
            self._plans[key] = (torch.tensor(pairs, dtype=torch.int32, device=self.device).contiguous(), len(pairs), lengths)
        return self._plans[key]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import _native

        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            all_grads = [p.grad for p in params]   # (one attribute read per parameter and step: 304 of them)
            have = tuple(i for i, g in enumerate(all_grads) if g is not None)
            if not have:
                continue
            _native.require_current_device(self._tabs[0])   # (launches go onto the current device's current stream)
            chunks, nchunks, lengths = self._plan(gi, have, params)
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing:   # a captured step replays this copy + launch: the tables must outlive the graph and never be rewritten
                if not self._spare:
                    raise RuntimeError("FusedSGD: more captured steps than pinned pointer tables (pinning host memory is not allowed "
                                       "during stream capture); call reserve_capture_tables(n) before capturing")
                host = self._spare.pop()
                tab, ev = torch.empty((len(have), 4), dtype=torch.int64, device=self.device), None
                self._captured.append((host, tab))
            else:
                slot = self._n % self.RING
                host, ev = self._ring[slot]
                tab = self._tabs[slot]
                self._n += 1
                ev.synchronize()   # (the copy AND the launch that last used this slot have run)
            full = len(have) == len(all_grads)
            grads = all_grads if full else [all_grads[i] for i in have]
            grads = [g if (g.dtype is f32 and g.is_contiguous()) else g.float().contiguous() for g in grads]   # (alive until queued)
            ps = params if full else [params[i] for i in have]
            state = self.state
            bufs = []
            for q in ps:
                st = state[q]
                buf = st.get("momentum_buffer")
                if buf is None or buf.shape != q.shape or buf.device != q.device or buf.dtype is not f32 or not buf.is_contiguous():
                    buf = st["momentum_buffer"] = torch.zeros_like(q) if buf is None else buf.to(q.device, f32).reshape(q.shape).contiguous()
                bufs.append(buf)
            rows = host.numpy()[:len(have)]
            rows[:, 0] = [q.data_ptr() for q in ps]
            rows[:, 1] = [g.data_ptr() for g in grads]
            rows[:, 2] = [b.data_ptr() for b in bufs]
            rows[:, 3] = lengths
            tab[:len(have)].copy_(host[:len(have)], non_blocking=True)
            rc = self.be.lib.pdf_sgd_step(nchunks, tab.data_ptr(), chunks.data_ptr(), float(group["lr"]), float(group["momentum"]),
                                          float(group["weight_decay"]), self.ctypes.c_void_p(_native.raw_stream()))
            if ev is not None:
                ev.record()
            if rc != 0:
                raise RuntimeError(f"pdf_sgd_step failed with status {rc}")
        return loss


def shard_scene_ids(num_scenes, rank, world_size):
    """Whole scenes are the sharding unit (engines/defaults.py:139 + DistributedSampler, engines/train.py:437-438)."""
    return list(range(rank, num_scenes, world_size))


def init_distributed():
    """Read the torchrun environment; returns (rank, local_rank, world_size). Backend 'nccl' is RCCL on ROCm."""
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if (world_size > 1 or os.environ.get("PDFOPS_FORCE_DDP")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return rank, local_rank, world_size
