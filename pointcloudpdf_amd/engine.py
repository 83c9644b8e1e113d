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


default_pseudo_mask.capturable = True   # (three elementwise kernels, no host read: recorded into the step's ONE graph like the real pass)


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


def graph_node_census(raw_graph):
    """Kinds of the nodes of a captured hipGraph (``torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()``), read through the
    runtime's own hipGraphGetNodes / hipGraphNodeGetType: {"nodes", "kernel", "memcpy", "memset", "other"}."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    g = ctypes.c_void_p(int(raw_graph))
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(g, None, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed: {rc}")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    rc = hip.hipGraphGetNodes(g, nodes, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed: {rc}")
    kinds = {"nodes": int(n.value), "kernel": 0, "memcpy": 0, "memset": 0, "other": 0}
    names = {0: "kernel", 1: "memcpy", 2: "memset"}   # hipGraphNodeTypeKernel / Memcpy / Memset (hip_runtime_api.h)
    for i in range(n.value):
        t = ctypes.c_int(-1)
        rc = hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        if rc != 0:
            raise RuntimeError(f"hipGraphNodeGetType failed: {rc}")
        kinds[names.get(t.value, "other")] += 1
    return kinds


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

    def __init__(self, step, batch, geom=None, warmup=2, autocast=None, loss_scale=1.0, stream=None, debug_graph=False, split_calls=None,
                 describe=None):
        """``split_calls``: names of backend entry points (``_native.HipBackend`` methods, e.g. ``("bottleneck_backward",)``): the step is
        captured as a SEQUENCE of graphs sharing one memory pool, cut before and after every call of one of them, so that a replayed step
        can carry HIP events around exactly those calls (``__call__(..., on_call=)``; events cannot be recorded inside a graph on this
        stack: tools/probes/event_in_graph_probe.py).  ``describe(name, args, out)``: what to remember about such a call (handed back to
        ``on_call``).  A training loop replays the ONE-graph capture; the segmented one is for the steps that measure.
        ``debug_graph``: keep the captured hipGraph inspectable (``node_census``: the rule "no memset node inside a captured step" is
        checked by walking the graph, tests/test_gpu_model.py).
        ``stream``: the stream to capture on (default: a new one) -- e.g. a CU-masked stream (``_native.cu_masked_stream``), see TrainStep.
        ``autocast``: torch.float16 / torch.bfloat16 -> the forward is captured under torch.autocast (the path then runs its
        reduced-precision products, dense.fp32_path).  ``loss_scale``: a ``DeviceGradScaler`` (the reference's AMP loop,
        engines/train.py:343-355: the captured backward starts from ``loss * scale`` with the scale read from DEVICE memory at replay time,
        and the gradients stay scaled until ``scaler.step(optimizer)`` / ``scaler.unscale_(optimizer)`` outside the graph -- after the
        data-parallel exchange, as torch orders it), or a float: static scale, divided out of the gradients again inside the graph
        (no overflow check: bench / A-B use only)."""
        from .geometry import Geometry, StaticGeometry

        self.autocast = autocast
        self.scaler = loss_scale if isinstance(loss_scale, DeviceGradScaler) else None
        self.loss_scale = 1.0 if self.scaler is not None else float(loss_scale)

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
        side = stream if stream is not None else torch.cuda.Stream(device=dev)
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
        self._keep_graph = bool(debug_graph)
        self.graph, self.graph2, self.mask_fn = torch.cuda.CUDAGraph(keep_graph=self._keep_graph), None, None
        self._census = None
        rec = getattr(step, "recognizer", None)
        fn = getattr(rec, "pseudo_mask_fn", None)
        self.segments = None
        if split_calls:
            if not (fn is None or getattr(fn, "capturable", False)):
                raise RuntimeError("CapturedStep(split_calls=): a step with a host-driven pseudo-label pass cannot be segmented")
            self._capture_segmented(side, dev, tuple(split_calls), describe)
        elif fn is None or getattr(fn, "capturable", False):   # (the sync-free pseudo-label pass is recorded like any other stage)
            with torch.cuda.graph(self.graph, stream=side):
                self.out = self._eager()
                self._backward(self.out)
        else:
            self._capture_around_the_pseudo_label_pass(rec, side, dev)
        self.grads = [p.grad for p in self.params]
        self.frozen = self._python_state()
        with torch.no_grad():   # (the capture itself does not execute anything, but keep the contract obvious)
            for b, v in zip(buffers, saved):
                b.copy_(v)
        release_autograd_state(step)

    def node_census(self):
        """Kinds of the nodes of the captured graph, read from the runtime (hipGraphGetNodes; needs ``debug_graph=True``, which keeps
        the hipGraph next to its executable form): {"nodes", "kernel", "memcpy", "memset", "other"}.
        A memset node inside a captured step is what replayed with stale arguments on ROCm 7.2 (docs/NOTEBOOK.md, round 5): the step must
        hold none, whatever the runtime's packet-capture switch says."""
        if self._census is None and self.segments is not None:
            self._census = {}
            for g, _, _ in self.segments:
                for k, v in graph_node_census(g.raw_cuda_graph()).items():
                    self._census[k] = self._census.get(k, 0) + v
            self._census["segments"] = len(self.segments)
        if self._census is None:
            self._census = graph_node_census(self.graph.raw_cuda_graph())
            if self.graph2 is not None:   # (a step split around a host-driven pseudo-label pass: both halves)
                second = graph_node_census(self.graph2.raw_cuda_graph())
                self._census = {k: v + second[k] for k, v in self._census.items()}
        return self._census

    def _capture_segmented(self, side, dev, split_calls, describe):
        """The step as a sequence of graphs (one pool): [... up to a named call][the call][... up to the next][the call] ... -- the
        capture switches graphs from INSIDE the forward / backward, in a wrapper around the named backend methods.  The backward runs on
        the capturing thread (autograd's per-device worker thread could not end a capture this thread began)."""
        import gc
        from . import _native

        be = _native.hip_backend()
        segs, cur, pool, graveyard = [], {"g": None}, torch.cuda.graph_pool_handle(), []

        def begin():
            g = torch.cuda.CUDAGraph(keep_graph=self._keep_graph)
            g.capture_begin(pool=pool)
            cur["g"] = g

        def end(label, info=None):
            import warnings
            with warnings.catch_warnings(record=True) as caught:   # (two named calls back to back leave an EMPTY graph between them)
                warnings.simplefilter("always")
                cur["g"].capture_end()
            empty = any("Graph is empty" in str(w.message) for w in caught)
            if not empty or label is not None:
                segs.append((cur["g"], label, info))
            else:   # dropped from the replay list, but kept alive until the whole capture is over: a graph destroyed while the NEXT
                graveyard.append(cur["g"])   # one captures makes a runtime call the capture forbids (the process aborts)

        originals = {}
        for name in split_calls:
            orig = getattr(be, name)
            originals[name] = (name in be.__dict__, orig)

            def wrapped(*a, _orig=orig, _n=name, **k):
                end(None)
                begin()
                out = _orig(*a, **k)
                end(_n, describe(_n, a, out) if describe is not None else None)
                begin()
                return out

            setattr(be, name, wrapped)
        torch.cuda.synchronize(dev)
        gc.collect()
        torch.cuda.empty_cache()
        try:
            with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
                try:
                    begin()
                    self.out = self._eager()
                    self._backward(self.out)
                    end(None)
                except BaseException:
                    try:   # leave no capture open behind the error (a graph destroyed while its stream captures aborts the process)
                        cur["g"].capture_end()
                    except Exception:   # noqa: BLE001
                        pass
                    raise
        finally:
            for name, (own, orig) in originals.items():
                if own:
                    setattr(be, name, orig)
                else:
                    delattr(be, name)
        self.segments, self.graph = segs, None
        cur["g"] = None
        del graveyard[:]

    def _capture_around_the_pseudo_label_pass(self, rec, side, dev):
        """A step whose recognizer runs the PDF pseudo-label pass (``PointPdfV1.pseudo_mask_fn``: region growing with data-dependent
        shapes and host reads, pointpdf_v1m1_base.py:118-382) is captured as TWO graphs sharing one memory pool: everything up to the
        pass (segmentor forward, U-decoder forward), and everything after it (the recognizer's loss, the whole backward).  A replay runs
        graph 1, the pass eagerly on the static logits (its host reads are legal there: nothing is being captured), copies the mask into
        the static mask tensor graph 2 was recorded with, and runs graph 2.  The capture switches graphs from INSIDE the forward: the
        recognizer's ``pseudo_mask_fn`` is swapped for a function that ends the first capture, begins the second and hands back the
        static mask."""
        import gc

        n = int(self.static["coord"].shape[0])
        self.static_mask = torch.zeros(n, dtype=torch.bool, device=dev)
        self.mask_fn, self.static_logits, g1, g2 = rec.pseudo_mask_fn, None, self.graph, torch.cuda.CUDAGraph(keep_graph=self._keep_graph)
        state = {"split": False}

        def switch_graphs(coord, seg_logits, offset):
            g1.capture_end()
            self.static_logits = seg_logits.detach()
            g2.capture_begin(pool=g1.pool())
            state["split"] = True
            return self.static_mask

        torch.cuda.synchronize(dev)
        gc.collect()
        torch.cuda.empty_cache()
        rec.pseudo_mask_fn = switch_graphs
        try:
            with torch.cuda.stream(side):
                g1.capture_begin()
                try:
                    self.out = self._eager()
                    self._backward(self.out)
                finally:
                    (g2 if state["split"] else g1).capture_end()
        finally:
            rec.pseudo_mask_fn = self.mask_fn
        if state["split"]:
            self.graph2 = g2
        else:   # (the pass is not part of this step: before start_epoch the recognizer returns ahead of it)
            self.mask_fn = None

    def _eager(self):
        with torch.autocast("cuda", dtype=self.autocast or torch.float16, enabled=self.autocast is not None):
            return self.step(dict(self.static, pdf_geometry=self.geometry))

    def _backward(self, out):
        if self.scaler is not None:
            self.scaler.scale(out["loss"]).backward()
        elif self.loss_scale == 1.0:
            out["loss"].backward()
        else:
            (out["loss"] * self.loss_scale).backward()
            with torch.no_grad():
                torch._foreach_mul_([p.grad for p in self.params if p.grad is not None], 1.0 / self.loss_scale)

    def _python_state(self):
        """Python-side scalars the capture bakes into the graph: the recognizer's loss weight ``alpha`` (decayed once at ``start_epoch``),
        whether the PDF loss is active (epoch gate), ``step_loss_weight``, train / eval mode of every module, the autocast mode."""
        rec = getattr(self.step, "recognizer", None)   # (a step module without a recognizer captures like any other)
        mods = self.__dict__.get("_modules_at_capture")
        if mods is None:   # the module list is walked once per capture, not on every replay
            mods = self.__dict__["_modules_at_capture"] = list(self.step.modules())
        return (float(getattr(rec, "alpha", 0.0)), int(getattr(rec, "epoch", 0)) >= int(getattr(rec, "start_epoch", 0)),
                bool(getattr(rec, "step_loss_weight", False)), tuple(m.training for m in mods), self.autocast)

    def matches(self, batch):
        """True when the captured graph IS this step: same scene sizes / shapes and the same Python-side schedule state as at capture.
        (A graph replays what was recorded: after ``PointPdfV1.trigger_operation`` moved ``alpha`` or an epoch gate opened, or after
        ``step.eval()``, the caller must run the eager step or capture again -- ``__call__`` refuses a stale graph.)"""
        return ([int(v) for v in batch["offset_host"]] == self.sizes and all(batch[k].shape == self.static[k].shape for k in self.KEYS)
                and self._python_state() == self.frozen)

    @torch.no_grad()
    def __call__(self, batch, geom, on_call=None):
        """``on_call(name, info, replay)`` (segmented captures only): called for every named backend call of the step instead of replaying
        its graph directly -- it must call ``replay()`` once (e.g. between two HIP events).
        One training step's forward + backward on ``batch`` with the coordinate-only tables ``geom`` (any Geometry of the batch: its own
        pre-pass or its share of a grouped one).  Two launches: the staging copy (batch tensors + ~70 tables into the fixed-address
        buffers, csrc/stage_copy.hip) and the graph.  Returns the static output dict (``loss``, ``model_loss``, ``recognizer_loss``,
        ``score``: overwritten by the next call); gradients are in ``p.grad``."""
        if self._python_state() != self.frozen:
            raise RuntimeError("CapturedStep: the step's Python-side state (recognizer alpha / epoch gate / step_loss_weight / train mode / "
                               f"autocast) changed since the capture: {self.frozen} -> {self._python_state()}; capture again "
                               "(CapturedStep(step, batch, ...)) or run the eager step")
        self.geometry.stage(geom, extra=[(batch[k], self.static[k]) for k in self.KEYS])
        if self.segments is not None:
            for g, label, info in self.segments:
                if label is not None and on_call is not None:
                    on_call(label, info, g.replay)
                else:
                    g.replay()
            for p, g in zip(self.params, self.grads):
                p.grad = g
            return self.out
        self.graph.replay()
        if self.graph2 is not None:   # the pseudo-label pass between the two halves (eager: it reads sizes back to the host)
            self.static_mask.copy_(self.mask_fn(self.static["coord"], self.static_logits, self.static["offset"]).bool())
            self.graph2.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return self.out


class GroupedGeometryLoader:
    """Wraps the training loader (any iterable of batch dicts: the reference's ``build_train_loader`` result,
    pointcept/engines/train.py:426-465) and yields the same batches with the coordinate-only tables of the step attached
    (``batch["pdf_geometry"]``: 4 FPS levels, 13 kNN tables, interpolation / inverse tables -- ``geometry.Geometry``), computed AHEAD of
    the step on a side HIP stream:

        loader = engine.GroupedGeometryLoader(build_train_loader(...), group=12)
        for batch in loader:              # batch tensors on the device, batch["pdf_geometry"] ready (stream-ordered, no host sync)
            out = train_step(batch)

    Why a group: farthest-point sampling is a chain of dependent arg-max steps on ONE workgroup per scene (100k -> 25k points: ~33 ms)
    that more CUs cannot shorten, so a per-batch pre-pass inline costs ~50 ms of latency per step.  The pre-pass of the NEXT ``group``
    batches runs as one launch sequence (one FPS launch with 2 * group workgroups, kNN launches ``group`` times larger) while the
    current group trains; ``Geometry.split`` hands out per-batch views that are bit-identical to a pre-pass of the batch alone
    (tests: test_geometry_split_matches_per_batch_prepass, test_grouped_loader_steps_are_bit_identical_to_serial_steps).  Every batch
    gets exactly one full pre-pass; nothing is cached across batches.  Two batches of look-ahead already clear 5 M points/s on
    2 x 100k-point batches; the default (12) also amortises the kNN launches.

    Timing owned here (none of it is the caller's business): the next group's batches are pulled from the wrapped loader -- and their
    host-to-device transfers started on a copy stream -- when the current group is entered; its pre-pass is submitted once the consumer
    came back for batch ``submit_delay`` of the current group, i.e. AFTER that many training steps were enqueued (the ~10 ms of host
    work of a submission never sits in front of a step), and waits only for the batch tensors (an event recorded when they were
    pulled), not for the steps queued since.
    ``first_group``: size of the first group only (default ``group``): a short first group gets the first batch out sooner.
    The wrapped loader should have a group's batches ready when it is pulled, one group ahead of use
    (``DataLoader(num_workers=w, prefetch_factor >= 2 * group / w)``).
    """

    MAX_SCENES = 64   # scenes per grouped kNN / FPS call (workspace layout of the grid kNN)

    def __init__(self, loader, group=12, device=None, first_group=None, submit_delay=2, prefetcher=None, threaded=False, key="pdf_geometry",
                 **plan):
        """``prefetcher`` / ``key``: another coordinate-only pre-pass with the same ``submit_group(batches, ready=)`` / ``get(ticket)``
        interface and the batch key its result goes under (``stratified.StratifiedPrefetcher`` -> "st_geometry": BASELINE config 5)."""
        from .geometry import GeometryPrefetcher

        self.key = key

        self.loader, self.group = loader, max(int(group), 0)
        self.first_group = self.group if first_group is None else max(int(first_group), 1)
        self.device = torch.device(device) if device is not None else None
        self.submit_delay = max(int(os.environ.get("PDFOPS_SUBMIT_DELAY", submit_delay)), 0)   # (env: A/B knob)
        self.prefetcher = prefetcher if prefetcher is not None else (GeometryPrefetcher(depth=2, threaded=threaded, **plan) if self.group > 0 else None)
        self.inline = None
        if self.group == 0 and prefetcher is None and key == "pdf_geometry" and torch.cuda.is_available():
            from .geometry import Geometry
            self.inline = lambda b: Geometry(b["coord"], b["offset"], b["offset_host"]).precompute(**plan)
        self.submit_host_s = []   # host time of every group submission (diagnostics)
        self._held, self._copy_stream = None, None

    def __len__(self):
        return len(self.loader)

    def warm(self, batches):
        """One throw-away pre-pass of ``batches`` (a group of the size that will be trained: any batches of those scene sizes) on EVERY
        side stream of the pre-pass.  The caching allocator keeps a pool per stream, and the first full-size group a stream sees takes its
        ~1 GB of tables and workspaces from hipMalloc: 60-70 ms of host time inside ``submit_group`` instead of 8, during which the
        training stream runs dry (seen as +3 ms per step on a 12-step measurement whose only in-region submission was such a first).
        Call it before the first step; a run of many groups reaches the same state by itself after ``depth`` + 1 groups."""
        if self.prefetcher is None or not hasattr(self.prefetcher, "streams") or not torch.cuda.is_available():
            return 0
        batches = [self._to_device(self._with_host_offset(b))[0] for b in batches]
        if self._copy_stream is not None:
            torch.cuda.current_stream().wait_stream(self._copy_stream)
            for b in batches:
                self._used_on(b, torch.cuda.current_stream())
        n = len(self.prefetcher.streams)
        for _ in range(n):
            tickets = self.prefetcher.submit_group(batches)
            for t in tickets:
                self.prefetcher.get(t)
            del tickets
        torch.cuda.synchronize()
        return n

    def _to_device(self, batch):
        """engines/train.py:373-376 (every tensor of the input dict moves to the device), on a copy stream of its own: the transfers of
        an upcoming group must not queue behind -- or in front of -- the training steps on the consumer's stream."""
        dev = self.device
        if dev is None or not any(torch.is_tensor(v) and v.device != dev for v in batch.values()):
            return batch, False
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        out = dict(batch)
        with torch.cuda.stream(self._copy_stream):
            for k, v in batch.items():
                if torch.is_tensor(v) and v.device != dev:
                    out[k] = v.to(dev, non_blocking=True)
        return out, True

    @staticmethod
    def _used_on(batch, stream):
        """The batch's tensors were allocated on the copy stream's pool: tell the caching allocator that ``stream`` reads them too, or
        their blocks go back to the copy stream's pool when the group is dropped and the NEXT group's transfers may overwrite them
        while steps that read them are still queued (the host runs several steps ahead of the device)."""
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(stream)

    @staticmethod
    def _with_host_offset(batch):
        if "offset_host" not in batch:   # scene ends on the host: level sizes of the pre-pass without a device sync per level
            batch = dict(batch)
            batch["offset_host"] = [int(v) for v in batch["offset"].tolist()]
        return batch

    def _pull(self, it, n):
        """Up to ``n`` batches (fewer at the end of the epoch, or when the scene budget of one grouped call is reached), moved to the
        device -> (batches, ready event | None) or None.  ``ready`` is recorded NOW: on the copy stream when tensors were moved, else on
        the consumer's stream as it stands -- a pre-pass submitted later waits for this point, not for the steps queued in between."""
        got, scenes, moved = [], 0, False
        while len(got) < n:
            held, self._held = self._held, None
            if held is None:
                try:
                    held = self._with_host_offset(next(it))
                except StopIteration:
                    break
            k = len(held["offset_host"])
            if got and scenes + k > self.MAX_SCENES:
                self._held = held
                break
            b, m = self._to_device(held)
            got.append(b)
            moved, scenes = moved or m, scenes + k
        if not got:
            return None
        ready = None
        if torch.cuda.is_available():
            ready = torch.cuda.Event()
            ready.record(self._copy_stream if moved else torch.cuda.current_stream())
        return got, ready, moved

    def _submit(self, pulled):
        import time

        batches, ready, moved = pulled
        t0 = time.perf_counter()
        tickets = self.prefetcher.submit_group(batches, ready=ready)
        self.submit_host_s.append(time.perf_counter() - t0)
        return [(b, t, ready if moved else None) for b, t in zip(batches, tickets)]

    def __iter__(self):
        it = iter(self.loader)
        self._held, self._copy_stream = None, None
        if self.group == 0:   # no look-ahead: the same pre-pass, per batch, inline on the consumer's stream (the serial step)
            for batch in it:
                b, moved = self._to_device(self._with_host_offset(batch))
                if moved:
                    torch.cuda.current_stream().wait_stream(self._copy_stream)
                    self._used_on(b, torch.cuda.current_stream())
                if self.inline is not None:
                    b = dict(b)
                    b[self.key] = self.inline(b)
                yield b
            return
        first = self._pull(it, self.first_group)
        if first is None:
            return
        current, pulled, upcoming, exhausted = self._submit(first), None, None, False
        while current is not None:
            delay = min(self.submit_delay, len(current) - 1)
            for j, (batch, ticket, copied) in enumerate(current):
                # The NEXT group: its batches are pulled (and their transfers started) when this group is entered -- BEFORE any of its steps
                # is enqueued, so that the pre-pass depends on nothing that trains -- and its pre-pass is submitted once we are back for
                # batch `delay` of this group, i.e. after that many steps were enqueued (the submission's host work never sits in front of them).
                if j == 0 and not exhausted:
                    pulled = self._pull(it, self.group)
                    exhausted = pulled is None
                if j == delay and pulled is not None:
                    upcoming, pulled = self._submit(pulled), None
                if copied is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(copied)   # the batch's own tensors (moved on the copy stream)
                    self._used_on(batch, cur)
                batch = dict(batch)
                batch[self.key] = self.prefetcher.get(ticket)   # consumer stream waits for the pre-pass (stream-ordered)
                yield batch
            if pulled is not None:   # (a group shorter than the delay)
                upcoming, pulled = self._submit(pulled), None
            current, upcoming = upcoming, None


class TrainStep:
    """One optimisation step on one batch -- ``Trainer.run_step`` of the reference (pointcept/engines/train.py:334-371): forward under
    autocast when AMP is on, (scaled) backward, gradient exchange, optimizer step, scaler update -- with forward + backward REPLAYED from
    a ``CapturedStep`` whenever the batch has the captured scene sizes (``graph=True``; the capture happens on the first such batch), and
    issued eagerly otherwise (other sizes, ``eager=True``, or a changed Python-side schedule state: see ``CapturedStep.matches``).

        train_step = engine.TrainStep(step, optimizer, exchange=engine.FlatGradAllReduce(step), autocast=torch.float16,
                                      scaler=engine.DeviceGradScaler(device))
        for batch in engine.GroupedGeometryLoader(loader, group=12):
            out = train_step(batch)       # dict(loss, model_loss, recognizer_loss, score)
    """

    def __init__(self, step, optimizer, exchange=None, scaler=None, autocast=None, graph=True, force_exchange=False, loss_scale=1.0, module=None,
                 stream=None, max_captures=1):
        """``module``: what the eager forward calls (a torch DistributedDataParallel wrapper of ``step``; default ``step`` itself);
        ``loss_scale``: static loss scale when no ``scaler`` is given (A/B runs; no overflow check).
        ``stream``: run EVERYTHING of the step (replay / eager issue, gradient exchange, optimizer) on this stream instead of the caller's
        current one -- with ``_native.cu_masked_stream`` the training stream and the coordinate pre-pass get disjoint compute units
        (``partition_compute_units``): the farthest-point chain (24 workgroups, ~20 ms per launch) next to the training kernels
        cost every one of them a straggler tail (+2.3 ms per step beside it at full duty, tools/contention_probe.py).
        ``max_captures``: how many scene-size classes get a graph of their own (default 1: the first batch's sizes -- what
        ``SphereCrop(point_max)`` hands the trainer for every scene above the limit).  With more, the first ``max_captures`` distinct
        size signatures are captured as they arrive (each capture costs ~3 eager steps once, and its private memory pool: the step's
        activations, ~3 GB at 2 x 100k points); further sizes run eagerly.  A capture whose Python-side schedule state went stale
        (``CapturedStep.matches``: the recognizer's alpha at ``start_epoch``) is released and its size class captured again."""
        self.stream = stream
        self.step, self.optimizer, self.exchange, self.scaler, self.autocast = step, optimizer, exchange, scaler, autocast
        self.module = module if module is not None else step
        self.graph, self.captured, self.capture_error = bool(graph), None, None
        self.captures, self.max_captures, self._capture_stream = [], max(int(max_captures), 1), stream
        self.instrumented, self.instrument_error = None, None   # the segmented capture of measuring steps (``__call__(..., on_call=)``)
        self.force_exchange, self.loss_scale = force_exchange, float(loss_scale)
        self.params = [p for p in step.parameters() if p.requires_grad]

    def capture(self, batch, geom=None):
        """Capture forward + backward for ``batch``'s scene sizes.  Every capture of one TrainStep records on the SAME stream (the first
        capture's): the parameters' AccumulateGrad nodes are bound to it for as long as any captured autograd graph lives."""
        cap = CapturedStep(self.step, batch, geom=geom, autocast=self.autocast,
                           loss_scale=self.scaler if self.scaler is not None else self.loss_scale, stream=self._capture_stream)
        self._capture_stream = cap.stream
        self.captures.append(cap)
        self.captured = self.captures[0]   # (the capture whose stream eager steps run on; ``is not None`` == "this trainer replays")
        return cap

    def drop_capture(self):
        """Release the captured graphs and every autograd / gradient reference they pinned (eager steps run at full speed again)."""
        import gc

        self.captured, self.graph, self.captures = None, False, []
        self._capture_stream = self.stream
        torch.cuda.synchronize()
        release_autograd_state(self.step)
        for p in self.step.parameters():
            p.grad = None
        gc.collect()
        torch.cuda.empty_cache()

    def _capture_for(self, batch, geom):
        """The capture that IS this batch's step (same scene sizes, same schedule state), made now when a slot is free; else None."""
        import gc

        for cap in self.captures:
            if cap.matches(batch):
                return cap
        if self.capture_error is not None:
            return None
        stale = [cap for cap in self.captures if cap._python_state() != cap.frozen]
        if stale:   # the schedule moved on (alpha / epoch gate): those graphs can never match again
            torch.cuda.synchronize()
            self.captures = [cap for cap in self.captures if cap not in stale]
            self.captured = self.captures[0] if self.captures else None
            for p in self.step.parameters():   # (p.grad may be a released capture's static tensor)
                p.grad = None
            del stale, cap
            gc.collect()
            torch.cuda.empty_cache()
        if len(self.captures) >= self.max_captures:
            return None
        try:
            return self.capture(batch, geom)
        except Exception as e:   # noqa: BLE001  (a stack that cannot capture the step trains on the eager path)
            self.capture_error = f"{type(e).__name__}: {e}"
            self.drop_capture()
            return None

    def _eager(self, batch):
        # An eager step between replays runs on the capture's stream: the parameters' AccumulateGrad nodes are bound to it while the
        # captured autograd graph lives, and on any other stream the engine synchronises every one of them (20-40 ms instead of 16).
        cur = torch.cuda.current_stream()
        run_on = self.captured.stream if self.captured is not None else cur
        if run_on is not cur:
            run_on.wait_stream(cur)
        with torch.cuda.stream(run_on):
            # every parameter of the step, not the requires_grad list of construction time: PointPdfV1.trigger_operation releases the
            # U-decoder's parameters INSIDE the forward at start_epoch, and their gradients must not accumulate across eager steps
            for p in self.step.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=self.autocast or torch.float16, enabled=self.autocast is not None):
                out = self.module(batch)
            if self.scaler is not None:
                self.scaler.scale(out["loss"]).backward()
            elif self.loss_scale == 1.0:
                out["loss"].backward()
            else:
                (out["loss"] * self.loss_scale).backward()
                with torch.no_grad():
                    torch._foreach_mul_([p.grad for p in self.params if p.grad is not None], 1.0 / self.loss_scale)
        if run_on is not cur:
            cur.wait_stream(run_on)
        return out

    def instrument(self, batch, geom, split_calls, describe=None):
        """Capture the step a second time as a sequence of graphs cut around ``split_calls`` (``CapturedStep(split_calls=)``), for steps
        that carry HIP events around those calls while everything else replays (``__call__(..., on_call=)``).  Costs one more pool of
        activations; made by whoever measures (bench.py, during its warm-up), never by a training loop.  Returns True when available."""
        if self.instrumented is not None and self.instrumented.matches(batch):
            return True
        if self.captured is None or self.instrument_error is not None:
            return False
        try:
            self.instrumented = CapturedStep(self.step, batch, geom=geom, autocast=self.autocast,
                                             loss_scale=self.scaler if self.scaler is not None else self.loss_scale, stream=self._capture_stream,
                                             split_calls=split_calls, describe=describe)
            return True
        except Exception as e:   # noqa: BLE001  (the measuring step then runs eagerly, as before round 6)
            if os.environ.get("PDFOPS_DEBUG_INSTRUMENT"):
                import traceback
                traceback.print_exc()
            self.instrument_error = f"{type(e).__name__}: {e}"
            self.instrumented = None
            release_autograd_state(self.step)
            return False

    def __call__(self, batch, eager=False, on_call=None):
        """``on_call``: run this step on the segmented capture (``instrument``) with ``on_call(name, info, replay)`` around the named calls;
        without a matching segmented capture the step runs eagerly when ``eager`` says so, else as usual."""
        self._on_call = on_call
        try:
            return self._call(batch, eager)
        finally:
            self._on_call = None

    def _call(self, batch, eager=False):
        if self.stream is None:
            return self._run(batch, eager)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)             # (the batch's tensors / tables were handed over on the caller's stream)
        # ... and were allocated on other streams' pools (the caller's, the copy stream's, the pre-pass's): the allocator must know that
        # self.stream reads them, or a block freed by the caller can be handed out again before the staging copy / the step ran here
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(self.stream)
        geom = batch.get("pdf_geometry")
        if geom is not None:
            if hasattr(geom, "tensors"):
                for t in geom.tensors():
                    t.record_stream(self.stream)
            packed = getattr(geom, "packed", None)
            if packed is not None:
                packed.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            out = self._run(batch, eager)
        return out                               # (the caller's stream does NOT wait: results are consumed on self.stream or after a sync)

    def _run(self, batch, eager=False):
        geom = batch.get("pdf_geometry")
        on_call = getattr(self, "_on_call", None)
        measuring = on_call is not None and self.instrumented is not None and self.instrumented.matches(batch) and geom is not None
        cap = self.instrumented if measuring else (self._capture_for(batch, geom) if self.graph and not eager else None)
        if measuring:
            out = cap(batch, geom, on_call=on_call)
        elif cap is not None:
            if geom is None:   # no look-ahead: the pre-pass inline on this stream
                from .geometry import Geometry
                geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
            out = cap(batch, geom)
        else:
            out = self._eager(batch)
        if self.exchange is not None:
            self.exchange.sync(force=self.force_exchange)   # ONE all-reduce (RCCL) over the flat gradient buffer
        if self.scaler is not None:
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            self.optimizer.step()
        return out


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
    ``load_state_dict`` with torch.optim.SGD's own layout: ``state[p]["momentum_buffer"]`` and param groups that carry every key
    torch.optim.SGD's do (``dampening`` 0, ``nesterov`` False, ``maximize`` False, ``foreach`` / ``fused`` None, ``differentiable``
    False), so checkpoints move both ways (tests/test_ddp_gloo.py: test_fused_sgd_state_dict_round_trip).  ``dampening != 0``,
    ``nesterov`` and ``maximize`` are refused at ``step()``.  Parameter and momentum pointers are read at every step (``model.to()``,
    ``load_state_dict`` may move them)."""

    RING = 8   # pinned pointer tables in flight (the host may run several steps ahead of the device)
    SGD_DEFAULTS = dict(dampening=0, nesterov=False, maximize=False, foreach=None, differentiable=False, fused=None)   # torch.optim.SGD's other keys

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0, backend=None):
        import ctypes
        from . import _native

        self.be = backend if backend is not None else _native.hip_backend()   # (backend: tests of the host logic without a GPU)
        self.ctypes = ctypes
        self._rows = 0
        self._ring, self._tabs, self._spare, self._captured, self._plans, self._n = [], [], [], [], {}, 0
        super().__init__(params, dict(lr=float(lr), momentum=float(momentum), weight_decay=float(weight_decay), **self.SGD_DEFAULTS))
        every = [p for group in self.param_groups for p in group["params"]]
        assert every and all(p.dtype == torch.float32 and p.is_contiguous() for p in every)
        self.device = every[0].device
        for p in every:
            self.state[p]["momentum_buffer"] = torch.zeros_like(p)
        self._table_cache, self._table_cache_on = {}, os.environ.get("PDFOPS_SGD_TABLE_CACHE") != "0"
        self.chunk = int(self.be.lib.pdf_sgd_chunk()) if self.be is not None else 4096
        self._size_tables()
        self.reserve_capture_tables(2 * len(self.param_groups))

    def add_param_group(self, param_group):
        """torch.optim.Optimizer.add_param_group + the pointer tables re-sized for the largest group (they are pinned once, not per step)."""
        super().add_param_group(param_group)
        group = self.param_groups[-1]
        group["params"] = [p for p in group["params"] if p.requires_grad]
        if self._rows:   # (during __init__ the tables are sized once, after every group is in)
            self._size_tables()

    def _size_tables(self):
        rows = max(len(g["params"]) for g in self.param_groups)
        if rows <= self._rows:
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedSGD.add_param_group: pinning host memory is not allowed during stream capture")
        for _, ev in self._ring:   # tables of steps still in flight stay alive until their launch has run
            if ev is not None:
                ev.synchronize()
        self._rows = rows
        pin = (lambda t: t.pin_memory()) if self.device.type == "cuda" else (lambda t: t)
        self._ring = [(pin(torch.empty((rows, 4), dtype=torch.int64)), torch.cuda.Event() if self.device.type == "cuda" else None)
                      for _ in range(self.RING)]
        self._tabs = [torch.empty((rows, 4), dtype=torch.int64, device=self.device) for _ in range(self.RING)]
        self._spare = [pin(torch.empty((rows, 4), dtype=torch.int64)) for _ in self._spare]
        self._plans = {}

    def reserve_capture_tables(self, n):
        """Pinned pointer tables for ``n`` more (group, captured step) pairs; must be called outside stream capture."""
        pin = (lambda t: t.pin_memory()) if self.device.type == "cuda" else (lambda t: t)
        self._spare += [pin(torch.empty((self._rows, 4), dtype=torch.int64)) for _ in range(n)]

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

    def _tables(self, gi, group):
        """Device table {param, grad, momentum, length} + chunk list of group ``gi`` for the parameters that have a gradient now.
        -> (nchunks, tab, chunks, event | None) or None when no parameter of the group has a gradient."""
        from . import _native

        f32 = torch.float32
        params = group["params"]
        all_grads = [p.grad for p in params]   # (one attribute read per parameter and step: 304 of them)
        have = tuple(i for i, g in enumerate(all_grads) if g is not None)
        if not have:
            return None
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
        fixed = [g if (g.dtype is f32 and g.is_contiguous()) else g.float().contiguous() for g in grads]   # (alive until queued)
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
        rows[:, 1] = [g.data_ptr() for g in fixed]
        rows[:, 2] = [b.data_ptr() for b in bufs]
        rows[:, 3] = lengths
        if not capturing and self._table_cache_on:
            # Replayed steps hand the SAME gradient tensors back every time: the table of the last step is then still right, and the host
            # -> device copy (19 KB through the copy engine, a cross-queue dependency in front of the optimizer launch: ~0.1 ms of idle
            # time on a quiet device, ~0.45 ms beside the pre-pass queues, profiles/r06_z_timeline.txt) is skipped.  The cached table is
            # its own device tensor, written only here.
            last = self._table_cache.get(gi)
            if last is not None and last[0].shape == rows.shape and (last[0] == rows).all():
                return nchunks, last[1], chunks, ev, fixed
            keep = torch.empty((len(have), 4), dtype=torch.int64, device=self.device)
            keep.copy_(host[:len(have)], non_blocking=True)
            self._table_cache[gi] = (rows.copy(), keep)
            return nchunks, keep, chunks, ev, fixed
        tab[:len(have)].copy_(host[:len(have)], non_blocking=True)
        return nchunks, tab, chunks, ev, fixed

    def _grad_key(self):
        return tuple(p.grad.data_ptr() if p.grad is not None else 0 for group in self.param_groups for p in group["params"])

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        """``found_inf``: a device float (``DeviceGradScaler``): non-zero -> the kernel leaves parameters and momenta untouched."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import _native

        prepared, self._prepared = getattr(self, "_prepared", None), None
        if prepared is not None:   # tables of an unscale_: valid only for the gradients they were built from (an iteration that aborted
            key, prepared = prepared   # between unscale_ and step, or re-assigned gradients, must not reuse stale pointer tables)
            if key != self._grad_key():
                prepared = None
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise RuntimeError("FusedSGD: dampening / nesterov / maximize are not implemented (the reference's configs use none of them)")
            t = prepared[gi] if prepared is not None else self._tables(gi, group)   # (unscale_ of the same iteration built them already)
            if t is None:
                continue
            nchunks, tab, chunks, ev, _alive = t
            rc = self.be.lib.pdf_sgd_step(nchunks, tab.data_ptr(), chunks.data_ptr(), float(group["lr"]), float(group["momentum"]),
                                          float(group["weight_decay"]), None if found_inf is None else found_inf.data_ptr(),
                                          self.ctypes.c_void_p(_native.raw_stream()))
            if ev is not None:
                ev.record()
            if rc != 0:
                raise RuntimeError(f"pdf_sgd_step failed with status {rc}")
        return loss


class DeviceGradScaler:
    """``torch.cuda.amp.GradScaler`` -- the dynamic loss scaling of the reference's AMP loop (pointcept/engines/train.py:343-355:
    ``scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()``) -- with the scale, the growth tracker and the found-inf
    flag as DEVICE scalars and every decision taken on the device: nothing is read back, so the step stays free of host syncs and a
    ``CapturedStep`` replays with whatever the scale is at replay time.  Same policy and defaults as torch's (init 65536, growth 2,
    backoff 0.5, interval 2000): a step that produced an inf / nan gradient leaves parameters AND momenta untouched and halves the scale.

        scaler = DeviceGradScaler(device)
        scaler.scale(loss).backward()          # or CapturedStep(..., loss_scale=scaler)
        exchange.sync()                        # data-parallel mean of the (still scaled) gradients: an inf reaches every rank
        scaler.step(optimizer)                 # FusedSGD: unscale_ (one launch: g *= 1 / scale, found_inf) + the guarded update
        scaler.update()                        # one 1-thread launch

    ``state_dict`` / ``load_state_dict`` use torch.amp.GradScaler's keys."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.device = torch.device(device)
        self.enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        # [scale, inv_scale, found_inf] in one allocation (three fixed addresses for captured steps) + the int32 tracker
        self._f = torch.tensor([float(init_scale), 1.0 / float(init_scale), 0.0], dtype=torch.float32, device=self.device)
        self._tracker = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._unscaled = False

    @property
    def scale_tensor(self):
        return self._f[0]

    @property
    def found_inf(self):
        return self._f[2:3]

    def scale(self, loss):
        return loss * self._f[0] if self.enabled else loss

    def get_scale(self):
        return float(self._f[0]) if self.enabled else 1.0   # (a host sync: logging only)

    @torch.no_grad()
    def unscale_(self, optimizer):
        """g *= 1 / scale for every gradient of ``optimizer`` (a FusedSGD) in one launch per group; sets found_inf."""
        if not self.enabled or self._unscaled:
            return
        if not isinstance(optimizer, FusedSGD):
            raise TypeError("DeviceGradScaler drives engine.FusedSGD (the guarded update is part of its kernel); use torch.amp.GradScaler "
                            "with other optimizers")
        from . import _native
        import ctypes

        prepared = []
        for gi, group in enumerate(optimizer.param_groups):
            t = optimizer._tables(gi, group)
            prepared.append(t)
            if t is None:
                continue
            nchunks, tab, chunks, _ev, _alive = t
            rc = optimizer.be.lib.pdf_grad_unscale(nchunks, tab.data_ptr(), chunks.data_ptr(), self._f[1:2].data_ptr(), self._f[2:3].data_ptr(),
                                                   ctypes.c_void_p(_native.raw_stream()))
            if rc != 0:
                raise RuntimeError(f"pdf_grad_unscale failed with status {rc}")
        optimizer._prepared = (optimizer._grad_key(), prepared)   # the step of this iteration reuses the tables (same gradients, same pointers)
        self._unscaled, self._prepared_for = True, optimizer

    def step(self, optimizer):
        if not self.enabled:
            return optimizer.step()
        self.unscale_(optimizer)
        return optimizer.step(found_inf=self._f[2:3])

    @torch.no_grad()
    def update(self):
        if not self.enabled:
            return
        from . import _native
        import ctypes

        lib = _native.hip_backend().lib
        rc = lib.pdf_scaler_update(self._f[0:1].data_ptr(), self._f[1:2].data_ptr(), self._tracker.data_ptr(), self._f[2:3].data_ptr(),
                                   ctypes.c_float(self.growth_factor), ctypes.c_float(self.backoff_factor), int(self.growth_interval),
                                   ctypes.c_void_p(_native.raw_stream()))
        if rc != 0:
            raise RuntimeError(f"pdf_scaler_update failed with status {rc}")
        self._unscaled = False
        opt, self._prepared_for = getattr(self, "_prepared_for", None), None
        if opt is not None:   # (an iteration that skipped optimizer.step(): its tables die with it)
            opt._prepared = None

    def state_dict(self):
        return dict(scale=self.get_scale(), growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                    growth_interval=self.growth_interval, _growth_tracker=int(self._tracker[0])) if self.enabled else {}

    def load_state_dict(self, state):
        if not self.enabled or not state:
            return
        self.growth_factor, self.backoff_factor = float(state["growth_factor"]), float(state["backoff_factor"])
        self.growth_interval = int(state["growth_interval"])
        with torch.no_grad():   # (in place: captured steps hold these addresses)
            self._f.copy_(torch.tensor([float(state["scale"]), 1.0 / float(state["scale"]), 0.0]))
            self._tracker.fill_(int(state["_growth_tracker"]))


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
