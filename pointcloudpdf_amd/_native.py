"""ctypes binding of the C ABI declared in include/pdfops.h.

``CBackend`` turns torch tensors into raw pointers + sizes and calls the C entry points.  The
product instance (``hip_backend()``) binds ``libpdfops.so`` (prefix ``pdf_``, device pointers, a
trailing hipStream_t).  There is NO CPU fallback in this package: if the library is missing or a
tensor is not on a ROCm device the call raises.  (The test-only oracle binds its own library
through the same class from ``oracle/``; nothing here imports it.)
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDFOPS_DIST_FMA=1|2: the library whose geometry kernels (kNN, ball query, FPS) evaluate the squared distance as an explicit FMA
# chain -- the arithmetic an `nvcc -O2` build of libs/pointops most likely runs (csrc/pdfops_common.h: pdf_sqdist3).
# Default: the as-written IEEE fp32 expression.
DIST_FMA = int(os.environ.get("PDFOPS_DIST_FMA", "0") or 0)
if DIST_FMA not in (0, 1, 2):
    raise RuntimeError(f"PDFOPS_DIST_FMA={DIST_FMA}: expected 0 (as written), 1 or 2 (csrc/pdfops_common.h)")


def library_path(dist_fma=0):
    return os.path.join(_HERE, "lib", "libpdfops.so" if not dist_fma else f"libpdfops_fma{int(dist_fma)}.so")


LIB_PATH = library_path(DIST_FMA)
ABI_VERSION = 6   # include/pdfops.h: PDF_ABI_VERSION (the argtypes below are THIS version's parameter lists)

c_int = ctypes.c_int
c_long = ctypes.c_long
c_void_p = ctypes.c_void_p
c_float = ctypes.c_float
c_double = ctypes.c_double

# name -> argument kinds ("i" int, "l" long, "f" float, "p" pointer); the stream pointer is appended for HIP.
_PROTOS = {
    "knn_query": "iippppipp",
    "ball_query": "iiffppppipp",
    "random_ball_query": "iiffpppppipp",
    "farthest_point_sampling": "iippppp",
    "grouping_forward": "iiippp",
    "grouping_backward": "iiippp",
    "interpolation_forward": "iiipppp",
    "interpolation_backward": "iiipppp",
    "subtraction_forward": "iiipppp",
    "subtraction_backward": "iiipppp",
    "aggregation_forward": "iiiippppp",
    "aggregation_backward": "iiiipppppppp",
    "attention_relation_step_forward": "iiipppppp",
    "attention_relation_step_backward": "iiippppppppp",
    "attention_fusion_step_forward": "iiippppp",
    "attention_fusion_step_backward": "iiippppppp",
    # libs/pointops2 window attention (v2 / v3 launchers)
    "attention_step1_forward_v2": "iiiiippppp",
    "attention_step1_backward_v2": "iiiiippppppp",
    "dot_prod_with_idx_forward_v3": "iiiiipppppppp",
    "dot_prod_with_idx_backward_v3": "iiiiipppppppppppp",
    "attention_step2_with_rel_pos_value_forward_v2": "iiiiippppppp",
    "attention_step2_with_rel_pos_value_backward_v2": "iiiiipppppppppp",
    "segment_softmax_forward": "iiippp",
    "segment_softmax_backward": "iiipppp",
}
_HIP_ONLY_PROTOS = {
    "grid_hash": "lippdddippp",
    "radius_neighbors_self": "iifppipppl",
    "vote_accumulate": "lipppppp",
    "graph_forest": "lipppppipppl",
    "gmm2_1d": "ipppidd",
    "dot_prod_with_idx_forward_v3_l": "iiiiipppppppp",
    "dot_prod_with_idx_backward_v3_l": "iiiiipppppppppppp",
    "attention_step2_with_rel_pos_value_backward_v2_l": "iiiiipppppppppp",
    "scene_sum_rows": "ipiplip",
    "scene_repeat_rows": "iplipip",
    "region_stats": "ippppifppp",
    "region_seeds": "ippppip",
    "region_grow": "ippipppipiipppp",
    "region_edges": "ipppppippppppppppppl",
    "sort_floats_dev": "ipppppp",
    "gmm2_weak_dev": "ippppppppidd",
    "region_mask": "ippppppp",
    "region_tree": "ippipppppppp",
    "graph_forest_dev": "lipppppippppl",
    "gmm2_1d_dev": "ippppidd",
    "graph_forest_batch_dev": "ippippppppipppl",
    "wa_segment_rows": "iiiipppppplfpplf",
    "wa_segment_rows_ordered": "iiiipppppplfpplfp",
    "wa_table_grad": "iiiippppplfpp",
    "wa_grad_attn": "iiiiiplppplppp",
    "wa_logits_forward": "iiiiipplfpppppp",
    "wa_permute_edges": "iippp",
    "wa_logits_forward_ordered": "iiiiipplfppppppp",
    "wa_grad_attn_ordered": "iiiiiplppplpppp",
    "window_keys": "ippippfippp",
    "window_edges_count": "ippipppppp",
    "window_edges_fill": "ipppppppffipppp",
    "group_forward": "iiiippppp",
    "group_backward": "iiiippp",
    "interpolation_weights": "iipp",
    "farthest_point_sampling_bucketed": "iiipppplp",
    "grouping_forward_ordered": "iiipppp",
    "group_forward_ordered": "iiiipppppp",
    "interpolation_forward_ordered": "iiippppp",
    "subtraction_forward_ordered": "iiippppp",
    "aggregation_forward_ordered": "iiiipppppp",
    "seg_sum_rows": "lipppifp",
    "seg_sum_rows_strided": "liplppifp",
    "seg_sum_rows_own": "liipppifppp",
    "seg_sum_weighted": "liiippppip",
    "seg_sum_weighted_ordered": "liiippppipp",
}
_KIND = {"i": c_int, "l": c_long, "f": c_float, "d": c_double, "p": c_void_p}


class PdfOpsError(RuntimeError):
    pass


# default since round 3: the complete GPU suite (315 tests, incl. the 10-evaluation bit-reproducibility test) passes with it on;
# PDFOPS_RAW_STREAM=0 goes back through torch.cuda.current_stream()
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("PDFOPS_RAW_STREAM", "1") != "0" else None


def raw_stream():
    """hipStream_t (as an integer) of torch's current stream on the current device.  ``torch.cuda.current_stream().cuda_stream`` builds a
    Stream object per call (~12 us; ~60 calls per training step); the raw accessor torch's own code generators use takes < 1 us."""
    dev = torch.cuda.current_device()
    return _RAW_STREAM(dev) if _RAW_STREAM is not None else torch.cuda.current_stream().cuda_stream


def cu_masked_stream(first_cu, n_cus, device=None):
    """A torch stream whose kernels run only on ``n_cus`` compute units starting at logical CU ``first_cu`` (hipExtStreamCreateWithCUMask).
    MI355X numbering, measured with tools/probes/cu_mask_probe.hip: mask bit i = CU (i // 8) of XCD (i % 8) -- consecutive bits go round
    the eight XCDs, so [first_cu, first_cu + n_cus) with both multiples of 8 takes the same CUs out of every XCD (a mask that leaves an XCD
    without any CU is not honoured: that XCD then runs on all of its CUs).  The stream is a BLOCKING stream in HIP's sense (the call has no
    flags): it synchronises implicitly with the legacy default stream, so nothing of a loop that uses it may run on that stream
    (engine.TrainStep(stream=...) keeps the whole step on its own stream).  The handle lives as long as the process."""
    if first_cu % 8 or n_cus % 8 or n_cus < 8:
        raise PdfOpsError("cu_masked_stream: first_cu and n_cus must be multiples of 8 (one CU of every XCD per step of 8)")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    if first_cu + n_cus > total:
        raise PdfOpsError(f"cu_masked_stream: CUs [{first_cu}, {first_cu + n_cus}) of {total}")
    hip = ctypes.CDLL("libamdhip64.so")
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(first_cu, first_cu + n_cus):
        mask[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise PdfOpsError(f"hipExtStreamCreateWithCUMask failed with status {rc}")
    return torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", dev))


def require_current_device(*tensors):
    """Every launch goes onto torch's current stream of the CURRENT device (the reference relies on
    ``torch.cuda.set_device(local_rank)`` the same way, engines/launch.py:131).  A tensor that lives on another GPU would be
    handed to a stream of the wrong device -- a fault, or a kernel unordered against the tensor's own stream -- so that is an
    error here, raised before anything is enqueued.  The public ``pointops`` ops switch devices themselves (CBackend._call)."""
    cur = None
    for t in tensors:
        if t is None or not t.is_cuda:
            continue
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise PdfOpsError(f"pointcloudpdf_amd: tensor on {t.device} but the current device is cuda:{cur}; call "
                              "torch.cuda.set_device(...) or wrap the call in torch.cuda.device(...)")


# ------------------------------------------------------------------------------------------------------------------
# Inverse neighbour tables (csrc/seg_gather.hip): entry ids of a table idx (m, nsample) grouped by destination row.
# Plumbing only (one stable device sort + one binary search per table); built once per table -- by the geometry pre-pass on
# its side stream for the tables the model uses, lazily here for any other idx tensor -- and cached on the idx tensor object.
# ------------------------------------------------------------------------------------------------------------------
_INV = "_pdf_inverse"


def attach_inverse(idx, n, table):
    setattr(idx, _INV, (idx.data_ptr(), idx._version, int(n), table))
    return table


def inverse_table(idx, n):
    """-> (inv_off (n + 1) int32, inv_entry int32, entry_base int) for idx (m, nsample) int32 with values in [-1, n)."""
    cached = getattr(idx, _INV, None)
    if cached is not None and cached[0] == idx.data_ptr() and cached[1] == idx._version and cached[2] == int(n):
        return cached[3]
    flat = idx.reshape(-1)
    vals, perm = torch.sort(flat, stable=True)        # ascending destination, ascending entry id inside a destination; -1 first
    off = torch.searchsorted(vals, torch.arange(int(n) + 1, device=idx.device, dtype=vals.dtype), out_int32=True)
    return attach_inverse(idx, n, (off, perm.to(torch.int32), 0))


_CSC = "_pdf_window_csc"
_WORDER = "_pdf_window_order"   # on the CSR offsets of a window edge table: the queries sorted by fine-window key (int32 permutation)


def window_order_of(offsets):
    """The visiting order ``HipBackend.window_edges`` left on a table's CSR offsets (queries window by window), or None."""
    tag = getattr(offsets, _WORDER, None)
    if tag is not None and tag[0] == offsets.data_ptr() and tag[1] == offsets._version and tag[2].shape[0] == offsets.shape[0] - 1:
        return tag[2]
    return None


def window_csc(index1, offsets, rel_idx=None, n_keys=None):
    """The window-attention edge list grouped by KEY (csrc/window_attention_bwd.hip): for index1 (M) int32 = key of every edge and the
    CSR offsets (N + 1) of the queries -> (key_off (N + 1) int32, key_edge (M) int32: edge ids grouped by key in ascending order,
    key_q (M) int32: the query of those edges[, key_rel (M, 3) int32: their rel_idx rows]).  Coordinate-only, so the pre-pass of a
    window partition builds it once (stratified.BasicLayer.window_tables) and every block / step reuses it: cached on ``index1``."""
    cache = getattr(index1, _CSC, None)
    if cache is None or cache["key"] != (index1.data_ptr(), index1._version, offsets.data_ptr(), offsets._version):
        n, m = offsets.shape[0] - 1, index1.shape[0]
        nk = n if n_keys is None else int(n_keys)
        vals, perm = torch.sort(index1, stable=True)
        key_off = torch.searchsorted(vals, torch.arange(nk + 1, device=index1.device, dtype=vals.dtype), out_int32=True)
        counts = (offsets[1:] - offsets[:-1]).long()
        index0 = torch.repeat_interleave(torch.arange(n, device=index1.device, dtype=torch.int32), counts, output_size=m)
        cache = {"key": (index1.data_ptr(), index1._version, offsets.data_ptr(), offsets._version),
                 "base": (key_off, perm.to(torch.int32), index0[perm].contiguous()), "perm": perm, "rel": {}}
        # Visiting order of the KEY-side row passes: the rows by key are long (a down-sampled point is a key of every query of its coarse
        # window: ~10x the entries) or short; owners that share a workgroup walk their rows in lockstep, so equal lengths belong together:
        # longest first, window by window inside a length (the window order of the table when the edge builder left one).  Level 0 of
        # config 5: grad_k 305 -> 152 us, grad_v 106 -> 59 us per call (tools/probes/wa_bwd_probe.py).
        if nk == n:
            wo = window_order_of(offsets)
            base = wo.long() if wo is not None else torch.arange(nk, device=index1.device)
            klen = (key_off[1:] - key_off[:-1]).index_select(0, base)
            cache["order"] = base.index_select(0, torch.sort(klen, descending=True, stable=True)[1]).to(torch.int32)
        setattr(index1, _CSC, cache)
    if rel_idx is None:
        return cache["base"]
    rk = (rel_idx.data_ptr(), rel_idx._version)
    if rk not in cache["rel"]:
        cache["rel"] = {rk: rel_idx[cache["perm"]].contiguous()}   # (one table per edge list in practice; a new one replaces the old)
    return cache["base"] + (cache["rel"][rk],)


def window_key_order(index1):
    """The key-side visiting order ``window_csc`` cached on ``index1`` (longest rows first, window by window), or None."""
    cache = getattr(index1, _CSC, None)
    return cache.get("order") if cache is not None and cache["key"][:2] == (index1.data_ptr(), index1._version) else None


_ORD = "_pdf_order"


def attach_order(idx, order, src_order=None):
    """Remember a visiting order of the QUERIES of a neighbour table (a permutation of its rows: the Morton order of the query points,
    Geometry.order) on the idx tensor; the forward gathers then walk the queries in that order (csrc/gather_ops.hip, *_ord kernels).
    ``src_order``: the same for the SOURCE rows (the destinations of the backward's segmented gathers)."""
    setattr(idx, _ORD, (idx.data_ptr(), idx._version, order, src_order))
    return idx


def src_order_of(idx, n):
    tag = getattr(idx, _ORD, None)
    if tag is not None and tag[0] == idx.data_ptr() and tag[1] == idx._version and tag[3] is not None and tag[3].shape[0] == n:
        return tag[3]
    return None


def order_of(idx):
    tag = getattr(idx, _ORD, None)
    if tag is not None and tag[0] == idx.data_ptr() and tag[1] == idx._version and tag[2].shape[0] == idx.shape[0]:
        return tag[2]
    return None


_MOM = "_pdf_moments"


def attach_moments(idx, moments):
    """Remember the batch's relative-coordinate sums of a SELF neighbour table (9 doubles on the device: csrc/geom_moments.hip) on
    its idx tensor; the fused layer's forward then takes the geometry branch's BatchNorm statistics from them."""
    setattr(idx, _MOM, (idx.data_ptr(), idx._version, moments))
    return idx


def moments_of(idx):
    tag = getattr(idx, _MOM, None)
    if tag is not None and tag[0] == idx.data_ptr() and tag[1] == idx._version:
        return tag[2]
    return None


def _check(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


class CBackend:
    """Calls a C library that implements the pdfops ABI on tensors of one device kind."""

    # the reference kernels ACCUMULATE into interpolation / aggregation / subtraction outputs, so its wrappers pre-zero them
    # (interpolation.py:39, aggregation.py:21); a backend whose kernels overwrite those outputs skips the fill
    overwrites_outputs = False

    def __init__(self, lib, prefix, device_type, use_stream, extra_protos=(), proto_overrides=None):
        self.lib = lib
        self.prefix = prefix
        self.device_type = device_type
        self.use_stream = use_stream
        self._fn = {}
        protos = dict(_PROTOS)
        for k in extra_protos:
            protos[k] = _HIP_ONLY_PROTOS[k]
        protos.update(proto_overrides or {})
        for name, kinds in protos.items():
            f = getattr(lib, prefix + name)
            f.restype = c_int
            f.argtypes = [_KIND[k] for k in kinds] + ([c_void_p] if use_stream else [])
            self._fn[name] = f

    # -- plumbing -------------------------------------------------------------------------------
    def _ptr(self, t):
        if t.device.type != self.device_type:
            raise PdfOpsError(
                f"pointcloudpdf_amd: tensor on '{t.device}' but this backend runs on '{self.device_type}' "
                "(the HIP path has no CPU fallback)"
            )
        return c_void_p(t.data_ptr())

    def _call(self, name, *args):
        conv, dev = [], None
        for a in args:
            if isinstance(a, torch.Tensor):
                conv.append(self._ptr(a))
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise PdfOpsError(f"{self.prefix}{name}: tensors on different devices ({dev}, {a.device})")
            else:
                conv.append(a)
        if self.use_stream:
            if dev is not None and dev.index != torch.cuda.current_device():
                # the stream must belong to the tensors' device (and per-device kernel attributes are set for the current one)
                with torch.cuda.device(dev):
                    rc = self._fn[name](*conv, c_void_p(raw_stream()))
            else:
                rc = self._fn[name](*conv, c_void_p(raw_stream()))
        else:
            rc = self._fn[name](*conv)
        if rc != 0:
            raise PdfOpsError(f"{self.prefix}{name} failed with status {rc}")

    def _new(self, like, shape, dtype, zero=False):
        return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=like.device)

    # -- reference ops ----------------------------------------------------------------------------
    def knn_query(self, nsample, xyz, new_xyz, offset, new_offset):
        """-> idx (m, nsample) int32, dist2 (m, nsample) f32 (squared).  knn_query_cuda, query.py:19-23."""
        _check(xyz, torch.float32, "xyz"); _check(new_xyz, torch.float32, "new_xyz")
        _check(offset, torch.int32, "offset"); _check(new_offset, torch.int32, "new_offset")
        if not 1 <= nsample <= 128:
            raise ValueError("nsample must be in 1..128")
        m = new_xyz.shape[0]
        idx = self._new(xyz, (m, nsample), torch.int32)
        dist2 = self._new(xyz, (m, nsample), torch.float32)
        if self.use_stream:
            self._call("knn_query", m, nsample, xyz, new_xyz, offset, new_offset, offset.shape[0], idx, dist2)
        else:
            self._call("knn_query", m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2)
        return idx, dist2

    def ball_query(self, nsample, max_radius, min_radius, xyz, new_xyz, offset, new_offset, order=None):
        """-> idx (m, nsample) int32, dist2 (m, nsample) f32.  ball_query_cuda / random_ball_query_cuda (``order`` given),
        query.py:48-64, 96-111."""
        _check(xyz, torch.float32, "xyz"); _check(new_xyz, torch.float32, "new_xyz")
        _check(offset, torch.int32, "offset"); _check(new_offset, torch.int32, "new_offset")
        if not min_radius < max_radius:
            raise ValueError("min_radius must be below max_radius")
        if not 1 <= nsample <= 2048:
            raise ValueError("nsample must be in 1..2048")
        m = new_xyz.shape[0]
        idx = self._new(xyz, (m, nsample), torch.int32)
        dist2 = self._new(xyz, (m, nsample), torch.float32)
        head = [m, nsample, float(min_radius), float(max_radius)]
        name = "ball_query"
        if order is not None:
            _check(order, torch.int32, "order")
            head.append(order)
            name = "random_ball_query"
        tail = [offset.shape[0], idx, dist2] if self.use_stream else [idx, dist2]
        self._call(name, *head, xyz, new_xyz, offset, new_offset, *tail)
        return idx, dist2

    def farthest_point_sampling(self, xyz, offset, new_offset, n_max, m_total):
        """-> idx (m_total,) int32.  farthest_point_sampling_cuda, sampling.py:18-22."""
        _check(xyz, torch.float32, "xyz")
        _check(offset, torch.int32, "offset"); _check(new_offset, torch.int32, "new_offset")
        idx = self._new(xyz, (m_total,), torch.int32, zero=True)
        tmp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
        self._call("farthest_point_sampling", offset.shape[0], int(n_max), xyz, offset, new_offset, tmp, idx)
        return idx

    def grouping_forward(self, input, idx):
        _check(input, torch.float32, "input"); _check(idx, torch.int32, "idx")
        m, ns = idx.shape
        c = input.shape[1]
        out = self._new(input, (m, ns, c), torch.float32)
        self._call("grouping_forward", m, ns, c, input, idx, out)
        return out

    def grouping_backward(self, grad_output, idx, n):
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx")
        m, ns, c = grad_output.shape
        gi = self._new(grad_output, (n, c), torch.float32, zero=True)
        self._call("grouping_backward", m, ns, c, grad_output, idx, gi)
        return gi

    def interpolation_forward(self, input, idx, weight):
        _check(input, torch.float32, "input"); _check(idx, torch.int32, "idx"); _check(weight, torch.float32, "weight")
        n, k = idx.shape
        c = input.shape[1]
        out = self._new(input, (n, c), torch.float32, zero=not self.overwrites_outputs)
        self._call("interpolation_forward", n, c, k, input, idx, weight, out)
        return out

    def interpolation_backward(self, grad_output, idx, weight, m):
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx"); _check(weight, torch.float32, "weight")
        n, c = grad_output.shape
        k = idx.shape[1]
        gi = self._new(grad_output, (m, c), torch.float32, zero=True)
        self._call("interpolation_backward", n, c, k, grad_output, idx, weight, gi)
        return gi

    def subtraction_forward(self, input1, input2, idx):
        _check(input1, torch.float32, "input1"); _check(input2, torch.float32, "input2"); _check(idx, torch.int32, "idx")
        n, c = input1.shape
        ns = idx.shape[-1]
        out = self._new(input1, (n, ns, c), torch.float32, zero=not self.overwrites_outputs)
        self._call("subtraction_forward", n, ns, c, input1, input2, idx, out)
        return out

    def subtraction_backward(self, idx, grad_output, n2=None):
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx")
        n, ns, c = grad_output.shape
        g1 = self._new(grad_output, (n, c), torch.float32, zero=True)
        g2 = self._new(grad_output, (n if n2 is None else n2, c), torch.float32, zero=True)
        self._call("subtraction_backward", n, ns, c, idx, grad_output, g1, g2)
        return g1, g2

    def aggregation_forward(self, input, position, weight, idx):
        for t, nm in ((input, "input"), (position, "position"), (weight, "weight")):
            _check(t, torch.float32, nm)
        _check(idx, torch.int32, "idx")
        n, ns, c = position.shape
        w_c = weight.shape[-1]
        out = self._new(input, (n, c), torch.float32, zero=not self.overwrites_outputs)
        self._call("aggregation_forward", n, ns, c, w_c, input, position, weight, idx, out)
        return out

    def aggregation_backward(self, input, position, weight, idx, grad_output):
        _check(grad_output, torch.float32, "grad_output")
        n, ns, c = position.shape
        w_c = weight.shape[-1]
        gi = self._new(input, tuple(input.shape), torch.float32, zero=True)
        gp = self._new(input, (n, ns, c), torch.float32, zero=not self.overwrites_outputs)
        gw = self._new(input, (n, ns, w_c), torch.float32, zero=True)
        self._call("aggregation_backward", n, ns, c, w_c, input, position, weight, idx, grad_output, gi, gp, gw)
        return gi, gp, gw

    def attention_relation_step_forward(self, query, key, weight, index_target, index_refer):
        for t, nm in ((query, "query"), (key, "key"), (weight, "weight")):
            _check(t, torch.float32, nm)
        _check(index_target, torch.int32, "index_target"); _check(index_refer, torch.int32, "index_refer")
        _, g, c = query.shape
        m = index_target.shape[0]
        out = self._new(query, (m, g), torch.float32, zero=True)
        self._call("attention_relation_step_forward", m, g, c, query, key, weight, index_target, index_refer, out)
        return out

    def attention_relation_step_backward(self, query, key, weight, index_target, index_refer, grad_output):
        _check(grad_output, torch.float32, "grad_output")
        n, g, c = query.shape
        m = index_target.shape[0]
        gq = self._new(query, (n, g, c), torch.float32, zero=True)
        gk = self._new(query, tuple(key.shape), torch.float32, zero=True)
        gw = self._new(query, (c,), torch.float32, zero=True)
        self._call("attention_relation_step_backward", m, g, c, query, gq, key, gk, weight, gw,
                   index_target, index_refer, grad_output)
        return gq, gk, gw

    def attention_fusion_step_forward(self, weight, value, index_target, index_refer):
        _check(weight, torch.float32, "weight"); _check(value, torch.float32, "value")
        _check(index_target, torch.int32, "index_target"); _check(index_refer, torch.int32, "index_refer")
        n, g, c = value.shape
        m = index_refer.shape[0]
        out = self._new(value, (n, g, c), torch.float32, zero=True)
        self._call("attention_fusion_step_forward", m, g, c, weight, value, index_target, index_refer, out)
        return out

    def attention_fusion_step_backward(self, weight, value, index_target, index_refer, grad_output):
        _check(grad_output, torch.float32, "grad_output")
        n, g, c = value.shape
        m = index_target.shape[0]
        gw = self._new(value, (m, g), torch.float32, zero=True)
        gv = self._new(value, (n, g, c), torch.float32, zero=True)
        self._call("attention_fusion_step_backward", m, g, c, weight, gw, value, gv, index_target, index_refer, grad_output)
        return gw, gv


    # -- libs/pointops2 window attention ---------------------------------------------------------------
    def attention_step1_v2(self, q, k, index1, offsets, n_max):
        """-> attn (M, h).  AttentionStep1_v2.forward, libs/pointops2/functions/pointops.py:170-199."""
        _check(q, torch.float32, "q"); _check(k, torch.float32, "k")
        _check(index1, torch.int32, "index1"); _check(offsets, torch.int32, "index0_offsets")
        n, h, d = q.shape
        m = index1.shape[0]
        if offsets.shape[0] != n + 1:
            raise ValueError("index0_offsets must have N + 1 entries")
        attn = self._new(q, (m, h), torch.float32)
        self._call("attention_step1_forward_v2", n, m, h, h * d, int(n_max), q, k, offsets, index1, attn)
        return attn

    def attention_step1_v2_backward(self, grad_out, q, k, index1, offsets, n_max):
        _check(grad_out, torch.float32, "grad_output")
        n, h, d = q.shape
        m = index1.shape[0]
        gq = self._new(q, tuple(q.shape), torch.float32)
        gk = self._new(q, tuple(k.shape), torch.float32, zero=True)
        self._call("attention_step1_backward_v2", n, m, h, h * d, int(n_max), grad_out, offsets, index1, q, k, gq, gk)
        return gq, gk

    def dot_prod_with_idx_v3(self, q, offsets, n_max, k, index_k, table_q, table_k, rel_idx):
        """-> out (M, h).  DotProdWithIdx_v3.forward, pointops.py:632-676."""
        for t, nm in ((q, "q"), (k, "k"), (table_q, "table_q"), (table_k, "table_k")):
            _check(t, torch.float32, nm)
        for t, nm in ((offsets, "index_q_offsets"), (index_k, "index_k"), (rel_idx, "rel_idx")):
            _check(t, torch.int32, nm)
        n, h, d = q.shape
        m = index_k.shape[0]
        if offsets.shape[0] != n + 1 or table_q.shape != table_k.shape or tuple(table_q.shape[1:]) != (h, d, 3):
            raise ValueError("dot_prod_with_idx_v3: inconsistent shapes")
        out = self._new(q, (m, h), torch.float32)
        if self.use_stream:   # HIP: per-head kernels with the table slabs in LDS (table length instead of n_max)
            self._call("dot_prod_with_idx_forward_v3_l", n, m, h, d, int(table_q.shape[0]), q, offsets, k, index_k, table_q, table_k, rel_idx, out)
        else:
            self._call("dot_prod_with_idx_forward_v3", n, m, h, d, int(n_max), q, offsets, k, index_k, table_q, table_k, rel_idx, out)
        return out

    def dot_prod_with_idx_v3_backward(self, grad_out, q, offsets, n_max, k, index_k, table_q, table_k, rel_idx):
        _check(grad_out, torch.float32, "grad_output")
        n, h, d = q.shape
        m = index_k.shape[0]
        gq = self._new(q, tuple(q.shape), torch.float32)
        gk = self._new(q, tuple(k.shape), torch.float32, zero=True)
        gtq = self._new(q, tuple(table_q.shape), torch.float32, zero=True)
        gtk = self._new(q, tuple(table_k.shape), torch.float32, zero=True)
        name, x = ("dot_prod_with_idx_backward_v3_l", int(table_q.shape[0])) if self.use_stream else ("dot_prod_with_idx_backward_v3", int(n_max))
        self._call(name, n, m, h, d, x, grad_out, q, offsets, k, index_k, table_q, table_k, rel_idx, gq, gk, gtq, gtk)
        return gq, gk, gtq, gtk

    def attention_step2_with_rel_pos_value_v2(self, attn, v, offsets, n_max, index1, table, rel_idx):
        """-> out (N, h, d).  AttentionStep2WithRelPosValue_v2.forward, pointops.py:854-895."""
        for t, nm in ((attn, "attn"), (v, "v"), (table, "table")):
            _check(t, torch.float32, nm)
        for t, nm in ((offsets, "index0_offsets"), (index1, "index1"), (rel_idx, "rel_idx")):
            _check(t, torch.int32, nm)
        m, h = attn.shape
        n, _, d = v.shape
        if offsets.shape[0] != n + 1 or tuple(table.shape[1:]) != (h, d, 3):
            raise ValueError("attention_step2_with_rel_pos_value_v2: inconsistent shapes")
        out = self._new(v, (n, h, d), torch.float32)
        self._call("attention_step2_with_rel_pos_value_forward_v2", n, m, h, d, int(n_max), attn, v, offsets, index1, table, rel_idx, out)
        return out

    def attention_step2_with_rel_pos_value_v2_backward(self, grad_out, attn, v, offsets, n_max, index1, table, rel_idx):
        _check(grad_out, torch.float32, "grad_output")
        m, h = attn.shape
        n, _, d = v.shape
        ga = self._new(v, (m, h), torch.float32, zero=True)
        gv = self._new(v, (n, h, d), torch.float32, zero=True)
        gt = self._new(v, tuple(table.shape), torch.float32, zero=True)
        name, x = (("attention_step2_with_rel_pos_value_backward_v2_l", int(table.shape[0])) if self.use_stream
                   else ("attention_step2_with_rel_pos_value_backward_v2", int(n_max)))
        self._call(name, n, m, h, d, x, grad_out, offsets, index1, attn, v, table, rel_idx, ga, gv, gt)
        return ga, gv, gt


    def segment_softmax(self, x, offsets):
        """-> y (M, h): softmax over the edges of every query, per head (scatter_softmax for a CSR-ordered index)."""
        _check(x, torch.float32, "src"); _check(offsets, torch.int32, "index0_offsets")
        m, h = x.shape
        if h > 64:
            raise ValueError("segment_softmax: at most 64 heads")
        y = self._new(x, (m, h), torch.float32)
        self._call("segment_softmax_forward", offsets.shape[0] - 1, m, h, offsets, x, y)
        return y

    def segment_softmax_backward(self, y, grad_y, offsets):
        _check(grad_y, torch.float32, "grad_output")
        m, h = y.shape
        gx = self._new(y, (m, h), torch.float32)
        self._call("segment_softmax_backward", offsets.shape[0] - 1, m, h, offsets, y, grad_y, gx)
        return gx


class HipBackend(CBackend):
    """libpdfops.so on the current ROCm device; adds the fused entry points."""

    overwrites_outputs = True

    def __init__(self, lib):
        super().__init__(lib, "pdf_", "cuda", True, extra_protos=tuple(_HIP_ONLY_PROTOS))
        lib.pdf_fps_workspace_bytes.restype = c_long
        lib.pdf_fps_workspace_bytes.argtypes = [c_int, c_int]
        lib.pdf_wa_table_grad_ws_floats.restype = c_long
        lib.pdf_wa_table_grad_ws_floats.argtypes = [c_int, c_int, c_int]
        lib.pdf_layernorm_supported.restype = c_int
        lib.pdf_layernorm_supported.argtypes = [c_int]
        lib.pdf_layernorm_partial_floats.restype = c_long
        lib.pdf_layernorm_partial_floats.argtypes = [c_long, c_int]
        lib.pdf_layernorm_forward.restype = c_int
        lib.pdf_layernorm_forward.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_layernorm_backward.restype = c_int
        lib.pdf_layernorm_backward.argtypes = [c_long, c_int] + [c_void_p] * 10
        lib.pdf_fps_stats_offset.restype = c_long
        lib.pdf_fps_stats_offset.argtypes = [c_int, c_int]
        self.collect_fps_stats = False  # debug: keep the work counters of the last bucketed FPS call (forces a sync)
        self.last_fps_stats = None
        lib.pdf_abi_version.restype = c_int
        lib.pdf_build_info.restype = ctypes.c_char_p
        got = int(lib.pdf_abi_version())
        if got != ABI_VERSION:   # parameter lists differ between versions: calling through would hand shifted arguments to the kernels
            raise PdfOpsError(f"libpdfops ABI {got} != {ABI_VERSION} expected by pointcloudpdf_amd/_native.py (include/pdfops.h: PDF_ABI_VERSION): "
                              "stale library -- rebuild with `python -m pointcloudpdf_amd.build --force`")
        for nm in ("pre_forward", "post_forward"):
            f = getattr(lib, "pdf_block_" + nm)
            f.restype = c_int
            f.argtypes = [c_long, c_int, c_void_p, c_int, ctypes.c_float, ctypes.c_float, c_int, c_void_p]
        lib.pdf_td_tables.restype = c_int
        lib.pdf_td_tables.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p,
                                      c_int, c_void_p]
        lib.pdf_td_supported.restype = c_int
        lib.pdf_td_supported.argtypes = [c_int, c_int, c_int]
        lib.pdf_td_gram_floats.restype = c_long
        lib.pdf_td_gram_floats.argtypes = [c_int]
        lib.pdf_td_fwd_scratch_floats.restype = c_long
        lib.pdf_td_fwd_scratch_floats.argtypes = [c_long, c_int]
        lib.pdf_td_bwd_scratch_floats.restype = c_long
        lib.pdf_td_bwd_scratch_floats.argtypes = [c_long, c_int, c_int]
        lib.pdf_td_forward.restype = c_int
        lib.pdf_td_forward.argtypes = [c_long, c_long, c_int, c_int, c_void_p, c_int, ctypes.c_float, ctypes.c_float, c_int, c_void_p]
        lib.pdf_td_backward.restype = c_int
        lib.pdf_td_backward.argtypes = [c_long, c_long, c_int, c_int, c_void_p, c_int, c_int, c_void_p]
        lib.pdf_ce_workspace_floats.restype = c_long
        lib.pdf_ce_workspace_floats.argtypes = []
        lib.pdf_ce_forward.restype = c_int
        lib.pdf_ce_forward.argtypes = [c_long, c_int, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_ce_backward.restype = c_int
        lib.pdf_ce_backward.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_knn_rel_moments.restype = c_int
        lib.pdf_knn_rel_moments.argtypes = [c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_knn_rel_moments_ws_doubles.restype = c_long
        lib.pdf_knn_rel_moments_ws_doubles.argtypes = [c_int, c_long]
        lib.pdf_knn_rel_moments_q.restype = c_int
        lib.pdf_knn_rel_moments_q.argtypes = [c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        class CopySeg(ctypes.Structure):   # include/pdfops.h: PdfCopySeg
            _fields_ = [("src", c_void_p), ("dst", c_void_p), ("nbytes", c_long), ("src_offset", c_void_p), ("sub", c_void_p),
                        ("sub_const", c_int), ("src_elems", c_int)]

        self.CopySeg = CopySeg
        lib.pdf_stage_copy.restype = c_int
        lib.pdf_stage_copy.argtypes = [c_int, c_void_p, c_void_p]
        lib.pdf_scene_morton_keys.restype = c_int
        lib.pdf_scene_morton_keys.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_sgd_chunk.restype = c_int
        lib.pdf_sgd_chunk.argtypes = []
        lib.pdf_sgd_step.restype = c_int
        lib.pdf_sgd_step.argtypes = [c_int, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]
        lib.pdf_grad_unscale.restype = c_int
        lib.pdf_grad_unscale.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.pdf_scaler_update.restype = c_int
        lib.pdf_scaler_update.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_int, c_void_p]
        lib.pdf_linbn_forward.restype = c_int
        lib.pdf_linbn_forward.argtypes = [c_long, c_int, c_int, c_void_p, c_int, c_int, ctypes.c_float, ctypes.c_float, c_int, c_void_p]
        lib.pdf_linbn_backward.restype = c_int
        lib.pdf_linbn_backward.argtypes = [c_long, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]
        lib.pdf_bottleneck_forward.restype = c_int
        lib.pdf_bottleneck_forward.argtypes = [c_long, c_int, c_int, c_void_p, c_int, ctypes.c_float, ctypes.c_float, c_int, c_int, c_void_p]
        lib.pdf_bottleneck_backward.restype = c_int
        lib.pdf_bottleneck_backward.argtypes = [c_long, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
        for nm in ("pre_backward", "post_backward"):
            f = getattr(lib, "pdf_block_" + nm)
            f.restype = c_int
            f.argtypes = [c_long, c_int, c_void_p, c_int, c_int, c_void_p]
        lib.pdf_rowlin_multi.restype = c_int
        lib.pdf_rowlin_multi.argtypes = [c_long, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_long, c_int, c_int, c_void_p]
        lib.pdf_kpconv_supported.restype = c_int
        lib.pdf_kpconv_supported.argtypes = [c_int, c_int]
        for nm in ("pdf_kpconv_gather", "pdf_kpconv_scatter"):
            f = getattr(lib, nm)
            f.restype = c_int
            f.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p]
        lib.pdf_rowlin_wgrad_group.restype = c_int
        lib.pdf_rowlin_wgrad_group.argtypes = [c_long, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        lib.pdf_rowlin_wgrad_multi.restype = c_int
        lib.pdf_rowlin_wgrad_multi.argtypes = [c_long, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        lib.pdf_rowlin_wgrad_ws_floats.restype = c_long
        lib.pdf_rowlin_wgrad_ws_floats.argtypes = [c_long, c_int, c_int, c_int]
        lib.pdf_rowlin_dgrad_bstats.restype = c_int
        lib.pdf_rowlin_dgrad_bstats.argtypes = [c_long, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long,
                                                c_void_p, c_int, c_void_p, ctypes.POINTER(c_int), c_int, c_void_p]
        lib.pdf_bn_act_backward_presummed.restype = c_int
        lib.pdf_bn_act_backward_presummed.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                                      c_void_p, c_void_p]
        lib.pdf_rowlin_partial_floats.restype = c_long
        lib.pdf_rowlin_partial_floats.argtypes = [c_long, c_int]
        lib.pdf_rowlin_partial_rows.restype = c_int
        lib.pdf_rowlin_partial_rows.argtypes = [c_long, c_int, c_int]
        lib.pdf_rowlin_forward.restype = c_int
        lib.pdf_rowlin_forward.argtypes = [c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_int, c_void_p, c_long, c_int, c_void_p, c_int, c_void_p]
        lib.pdf_rowlin_wgrad.restype = c_int
        lib.pdf_rowlin_wgrad.argtypes = [c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        lib.pdf_bn_coef_from_partial.restype = c_int
        lib.pdf_bn_coef_from_partial.argtypes = [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]
        lib.pdf_bn_coef.restype = c_int
        lib.pdf_bn_coef.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_float,
                                    ctypes.c_float, c_void_p, c_void_p, c_void_p]
        lib.pdf_bn_apply.restype = c_int
        lib.pdf_bn_apply.argtypes = [c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        lib.pdf_bn_supported.restype = c_int
        lib.pdf_bn_supported.argtypes = [c_int]
        lib.pdf_bn_partial_floats.restype = c_long
        lib.pdf_bn_partial_floats.argtypes = [c_long, c_int]
        lib.pdf_bn_act_forward.restype = c_int
        lib.pdf_bn_act_forward.argtypes = [c_long, c_int] + [c_void_p] * 6 + [c_int, ctypes.c_float, ctypes.c_float, c_int] + [c_void_p] * 4
        lib.pdf_bn_act_backward.restype = c_int
        lib.pdf_bn_act_backward.argtypes = [c_long, c_int] + [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 5
        lib.pdf_pt_layer_supported.restype = c_int
        lib.pdf_pt_layer_supported.argtypes = [c_int, c_int]
        for fn in (lib.pdf_pt_layer_partial_floats, lib.pdf_pt_layer_bwd_partial_floats):
            fn.restype = c_long
            fn.argtypes = [c_int, c_int, c_int]
        lib.pdf_pt_layer_bwd_sums_floats.restype = c_long
        lib.pdf_pt_layer_bwd_sums_floats.argtypes = [c_int]
        lib.pdf_pt_layer_forward.restype = c_int
        lib.pdf_pt_layer_forward.argtypes = [c_int, c_int, c_int] + [c_void_p] * 8 + [c_int, ctypes.c_float, ctypes.c_float] + [c_void_p] * 5 + [c_int, c_void_p, c_void_p]
        lib.pdf_knn_grid_build.restype = c_int
        lib.pdf_knn_grid_build.argtypes = [c_int, c_void_p, c_void_p, c_int, c_void_p, c_long, c_void_p]
        lib.pdf_knn_query_grid.restype = c_int
        lib.pdf_knn_query_grid.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p]
        lib.pdf_pt_layer_backward.restype = c_int
        lib.pdf_pt_layer_backward.argtypes = [c_int, c_int, c_int] + [c_void_p] * 19 + [c_int] + [c_void_p] * 2 + [c_int, c_void_p, c_void_p, c_void_p]
        self.fps_mode = os.environ.get("PDFOPS_FPS", "bucketed")  # "bucketed" | "plain"
        self.knn_mode = os.environ.get("PDFOPS_KNN", "grid")      # "grid" | "scan"
        lib.pdf_knn_workspace_bytes.restype = c_long
        lib.pdf_knn_workspace_bytes.argtypes = [c_int, c_int, c_int]
        lib.pdf_graph_forest_workspace_bytes.restype = c_long
        lib.pdf_region_grow_list_points.restype = c_long
        lib.pdf_region_grow_list_points.argtypes = []
        lib.pdf_graph_forest_workspace_bytes.argtypes = [c_long, c_long, c_long]
        lib.pdf_knn_grid_supported.restype = c_int
        lib.pdf_knn_grid_supported.argtypes = [c_int]
        lib.pdf_knn_query_ws.restype = c_int
        lib.pdf_knn_query_ws.argtypes = [c_int, c_int, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_long, c_void_p]
        lib.pdf_knn_query_ws_counted.restype = c_int
        lib.pdf_knn_query_ws_counted.argtypes = [c_int, c_int, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_long, c_void_p, c_void_p]

    KNN_GRID_MAX_SCENES = 64
    # scatter-adds of the gather family as segmented gathers over inverse tables (PDFOPS_INVERSE=0: the atomic kernels, for A/B runs)
    use_inverse = os.environ.get("PDFOPS_INVERSE", "1") != "0"
    # reduced-precision variant (the reference trains under AMP, engines/train.py:343-356): the fused PointTransformerLayer keeps the
    # row arrays only it reads -- H (saved), G2 / Wsm / GR (backward scratch) -- as bfloat16; sums, products, statistics, parameters and
    # every tensor that crosses the layer boundary stay fp32.  Set per process (PDFOPS_STORAGE=bf16) or with set_storage().
    storage_bf16 = os.environ.get("PDFOPS_STORAGE", "f32") == "bf16"

    def set_storage(self, kind):
        if kind not in ("f32", "bf16"):
            raise ValueError("storage: 'f32' or 'bf16'")
        self.storage_bf16 = kind == "bf16"

    # forward gathers: queries visited in the order attached to the idx tensor (Geometry: Morton order of the query level), if any
    def grouping_forward(self, input, idx):
        _check(input, torch.float32, "input"); _check(idx, torch.int32, "idx")
        m, ns = idx.shape
        c = input.shape[1]
        out = self._new(input, (m, ns, c), torch.float32)
        self._call("grouping_forward_ordered", m, ns, c, input, idx, order_of(idx), out)
        return out

    def interpolation_forward(self, input, idx, weight):
        _check(input, torch.float32, "input"); _check(idx, torch.int32, "idx"); _check(weight, torch.float32, "weight")
        n, k = idx.shape
        c = input.shape[1]
        out = self._new(input, (n, c), torch.float32)
        self._call("interpolation_forward_ordered", n, c, k, input, idx, weight, order_of(idx), out)
        return out

    def subtraction_forward(self, input1, input2, idx):
        _check(input1, torch.float32, "input1"); _check(input2, torch.float32, "input2"); _check(idx, torch.int32, "idx")
        n, c = input1.shape
        ns = idx.shape[-1]
        out = self._new(input1, (n, ns, c), torch.float32)
        self._call("subtraction_forward_ordered", n, ns, c, input1, input2, idx, order_of(idx), out)
        return out

    def aggregation_forward(self, input, position, weight, idx):
        for t, nm in ((input, "input"), (position, "position"), (weight, "weight")):
            _check(t, torch.float32, nm)
        _check(idx, torch.int32, "idx")
        n, ns, c = position.shape
        out = self._new(input, (n, c), torch.float32)
        self._call("aggregation_forward_ordered", n, ns, c, weight.shape[-1], input, position, weight, idx, order_of(idx), out)
        return out

    def grouping_backward(self, grad_output, idx, n):
        if not self.use_inverse:
            return super().grouping_backward(grad_output, idx, n)
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx")
        m, ns, c = grad_output.shape
        if grad_output.numel() == 0 or n == 0:
            return self._new(grad_output, (n, c), torch.float32, zero=True)
        off, ent, base = inverse_table(idx, n)
        gi = self._new(grad_output, (n, c), torch.float32)
        self._call("seg_sum_rows", n, c, grad_output, off, ent, base, 1.0, gi)
        return gi

    def interpolation_backward(self, grad_output, idx, weight, m):
        if not self.use_inverse:
            return super().interpolation_backward(grad_output, idx, weight, m)
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx"); _check(weight, torch.float32, "weight")
        n, c = grad_output.shape
        if grad_output.numel() == 0 or m == 0:
            return self._new(grad_output, (m, c), torch.float32, zero=True)
        off, ent, base = inverse_table(idx, m)
        gi = self._new(grad_output, (m, c), torch.float32)
        self._call("seg_sum_weighted_ordered", m, c, idx.shape[1], 1, grad_output, weight, off, ent, base, src_order_of(idx, m), gi)
        return gi

    def subtraction_backward(self, idx, grad_output, n2=None):
        if not self.use_inverse:
            return super().subtraction_backward(idx, grad_output, n2)
        _check(grad_output, torch.float32, "grad_output"); _check(idx, torch.int32, "idx")
        n, ns, c = grad_output.shape
        n2 = n if n2 is None else n2
        if grad_output.numel() == 0 or n2 == 0:
            return super().subtraction_backward(idx, grad_output, n2)
        off, ent, base = inverse_table(idx, n2)
        g2 = self._new(grad_output, (n2, c), torch.float32)
        if n == n2 and c % 4 == 0 and self.fuse_own_rows:   # self table: both sums in one walk over the gradient (csrc/seg_gather.hip)
            g1 = self._new(grad_output, (n, c), torch.float32)
            self._call("seg_sum_rows_own", n2, c, ns, grad_output, off, ent, base, -1.0, src_order_of(idx, n2), g2, g1)
            return g1, g2
        g1 = self._new(grad_output, (n, c), torch.float32, zero=True)
        self._call("subtraction_backward", n, ns, c, idx, grad_output, g1, None)       # row sums only
        self._call("seg_sum_rows", n2, c, grad_output, off, ent, base, -1.0, g2)
        return g1, g2

    fuse_own_rows = os.environ.get("PDFOPS_FUSE_OWN_ROWS", "1") != "0"   # 0: two passes over the gradient (rounds 2-3; A/B runs)

    def aggregation_backward(self, input, position, weight, idx, grad_output):
        if not self.use_inverse:
            return super().aggregation_backward(input, position, weight, idx, grad_output)
        _check(grad_output, torch.float32, "grad_output")
        n, ns, c = position.shape
        w_c = weight.shape[-1]
        if position.numel() == 0 or input.shape[0] == 0:
            return super().aggregation_backward(input, position, weight, idx, grad_output)
        off, ent, base = inverse_table(idx, input.shape[0])
        gi = self._new(input, tuple(input.shape), torch.float32)
        gp = self._new(input, (n, ns, c), torch.float32)
        gw = self._new(input, (n, ns, w_c), torch.float32, zero=True)
        self._call("aggregation_backward", n, ns, c, w_c, input, position, weight, idx, grad_output, None, gp, gw)
        self._call("seg_sum_weighted_ordered", input.shape[0], c, ns, w_c, grad_output, weight, off, ent, base, src_order_of(idx, input.shape[0]), gi)
        return gi, gp, gw

    def knn_grid(self, xyz, offset, m_max):
        """The uniform grid over the source points ``xyz`` (csrc/knn_grid.hip) as a workspace tensor that ``knn_query(..., grid=)`` takes for
        any query set of up to ``m_max`` points and nsample 3 / 8 / 16; None where the grid path does not apply."""
        n, b = xyz.shape[0], offset.shape[0]
        if self.knn_mode == "scan" or b > self.KNN_GRID_MAX_SCENES or n < 1:
            return None
        _check(xyz, torch.float32, "xyz"); _check(offset, torch.int32, "offset")
        require_current_device(xyz, offset)
        nbytes = int(self.lib.pdf_knn_workspace_bytes(b, n, int(max(m_max, n))))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
        rc = self.lib.pdf_knn_grid_build(n, self._ptr(xyz), self._ptr(offset), b, self._ptr(ws), nbytes, c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_knn_grid_build failed with status {rc}")
        return ws

    def knn_query(self, nsample, xyz, new_xyz, offset, new_offset, grid=None):
        if grid is not None and self.lib.pdf_knn_grid_supported(int(nsample)) and new_xyz.shape[0] > 0:
            _check(new_xyz, torch.float32, "new_xyz"); _check(new_offset, torch.int32, "new_offset")
            require_current_device(xyz, new_xyz, offset, new_offset)
            n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
            if int(self.lib.pdf_knn_workspace_bytes(b, n, m)) > grid.numel():
                raise PdfOpsError("knn_query: the grid workspace was built for fewer queries")
            idx = self._new(xyz, (m, nsample), torch.int32)
            dist2 = self._new(xyz, (m, nsample), torch.float32)
            rc = self.lib.pdf_knn_query_grid(m, int(nsample), n, self._ptr(xyz), self._ptr(new_xyz), self._ptr(offset), self._ptr(new_offset), b,
                                             self._ptr(idx), self._ptr(dist2), self._ptr(grid), grid.numel(), c_void_p(raw_stream()))
            if rc != 0:
                raise PdfOpsError(f"pdf_knn_query_grid failed with status {rc}")
            return idx, dist2
        if self.knn_mode == "scan" or not self.lib.pdf_knn_grid_supported(int(nsample)):
            return super().knn_query(nsample, xyz, new_xyz, offset, new_offset)
        _check(xyz, torch.float32, "xyz"); _check(new_xyz, torch.float32, "new_xyz")
        _check(offset, torch.int32, "offset"); _check(new_offset, torch.int32, "new_offset")
        n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
        if b > self.KNN_GRID_MAX_SCENES:   # the grid workspace holds <= 64 scenes; beyond, the C side scans anyway (no workspace)
            return super().knn_query(nsample, xyz, new_xyz, offset, new_offset)
        require_current_device(xyz, new_xyz, offset, new_offset)
        idx = self._new(xyz, (m, nsample), torch.int32)
        dist2 = self._new(xyz, (m, nsample), torch.float32)
        nbytes = int(self.lib.pdf_knn_workspace_bytes(b, n, m))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
        rc = self.lib.pdf_knn_query_ws(m, int(nsample), n, self._ptr(xyz), self._ptr(new_xyz), self._ptr(offset),
                                       self._ptr(new_offset), b, self._ptr(idx), self._ptr(dist2), self._ptr(ws), nbytes,
                                       c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_knn_query_ws failed with status {rc}")
        return idx, dist2

    def scene_morton_keys(self, xyz, offset):
        """-> keys (n,) int64: scene << 30 | Morton code on the scene's own bounding box (include/pdfops.h: pdf_scene_morton_keys)."""
        _check(xyz, torch.float32, "xyz"); _check(offset, torch.int32, "offset")
        require_current_device(xyz, offset)
        n, b = xyz.shape[0], offset.shape[0]
        keys = torch.empty((n,), dtype=torch.int64, device=xyz.device)
        bounds = torch.empty((4 * b,), dtype=torch.float32, device=xyz.device)
        rc = self.lib.pdf_scene_morton_keys(n, b, self._ptr(xyz), self._ptr(offset), self._ptr(bounds), self._ptr(keys), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_scene_morton_keys failed with status {rc}")
        return keys

    def knn_query_counted(self, nsample, xyz, new_xyz, offset, new_offset):
        """Measurement aid: the grid kNN with its kernel counting the candidate distances it evaluates -> (idx, dist2, evaluated pairs)
        (one host read-back).  tools/ops_roofline.py prices the kernel against the fp32 vector peak with this count."""
        n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
        require_current_device(xyz, new_xyz, offset, new_offset)
        idx = self._new(xyz, (m, nsample), torch.int32)
        dist2 = self._new(xyz, (m, nsample), torch.float32)
        nbytes = int(self.lib.pdf_knn_workspace_bytes(b, n, m))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
        pairs = torch.zeros((1,), dtype=torch.int64, device=xyz.device)
        rc = self.lib.pdf_knn_query_ws_counted(m, int(nsample), n, self._ptr(xyz), self._ptr(new_xyz), self._ptr(offset),
                                               self._ptr(new_offset), b, self._ptr(idx), self._ptr(dist2), self._ptr(ws), nbytes,
                                               self._ptr(pairs), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_knn_query_ws_counted failed with status {rc}")
        return idx, dist2, int(pairs.item())

    def farthest_point_sampling(self, xyz, offset, new_offset, n_max, m_total):
        if self.fps_mode == "plain":
            return super().farthest_point_sampling(xyz, offset, new_offset, n_max, m_total)
        _check(xyz, torch.float32, "xyz")
        _check(offset, torch.int32, "offset"); _check(new_offset, torch.int32, "new_offset")
        b, n_total = offset.shape[0], xyz.shape[0]
        idx = self._new(xyz, (m_total,), torch.int32, zero=True)
        nbytes = int(self.lib.pdf_fps_workspace_bytes(b, n_total))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
        self._call("farthest_point_sampling_bucketed", b, int(n_max), n_total, xyz, offset, new_offset, ws, nbytes, idx)
        if self.collect_fps_stats:
            o = int(self.lib.pdf_fps_stats_offset(b, n_total))
            self.last_fps_stats = ws[o:o + 16 * b].view(torch.int32).view(b, 4).cpu()
        return idx

    # -- fused PointTransformerLayer -----------------------------------------------------------------
    def pt_layer_supported(self, nsample, c):
        return bool(self.lib.pdf_pt_layer_supported(int(nsample), int(c)))

    @staticmethod
    def _ptr_array(tensors):
        return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])

    def pt_layer_forward(self, xq, xk, xv, p, idx, weights, bn_params, bn_buffers, training, eps, momentum):
        """-> out (N,C), ctx tensors (bn scale/shift, saved mean/rstd, H).  See include/pdfops.h."""
        n, c = xq.shape
        k = idx.shape[1]
        cs = c // 8
        for t in (xq, xk, xv, p, *weights, *bn_params):
            _check(t, torch.float32, "pt_layer tensor")
        _check(idx, torch.int32, "idx")
        require_current_device(xq, xk, xv, p, idx)
        bn = self._new(xq, (2 * (3 + c + cs),), torch.float32)
        saved = self._new(xq, (2 * (3 + c + cs),), torch.float32)
        H = self._new(xq, (n, k, cs), torch.float32)
        partial = self._new(xq, (int(self.lib.pdf_pt_layer_partial_floats(n, k, c)),), torch.float32)
        out = self._new(xq, (n, c), torch.float32)
        rc = self.lib.pdf_pt_layer_forward(
            n, k, c, self._ptr(xq), self._ptr(xk), self._ptr(xv), self._ptr(p), self._ptr(idx),
            self._ptr_array(weights), self._ptr_array(bn_params), self._ptr_array(bn_buffers), int(bool(training)),
            ctypes.c_float(eps), ctypes.c_float(momentum), self._ptr(bn), self._ptr(saved), self._ptr(H),
            self._ptr(partial), self._ptr(out), self.layer_flags(self.storage_bf16), self._order_ptr(idx), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_pt_layer_forward failed with status {rc}")
        return out, bn, saved, H

    # visiting order of the points in the fused layer passes (PDFOPS_LAYER_ORDER=0: storage order, for A/B runs)
    layer_order = os.environ.get("PDFOPS_LAYER_ORDER", "0") == "1"     # measured: no gain (22.3 vs 22.1 ms per step), the passes are not
    layer_chunked = os.environ.get("PDFOPS_LAYER_CHUNKED", "0") == "1"  # bound by where their rows come from -- both off by default

    def layer_flags(self, bf16):
        """the `storage_bf16` argument of the layer entry points: bit 0 = bfloat16 row arrays, bit 1 = chunked point walk"""
        return int(bool(bf16)) | (2 if self.layer_chunked else 0) | (4 if self.layer_order else 0)

    use_moments = os.environ.get("PDFOPS_BNP_MOMENTS", "1") != "0"   # 0: the layer's own first statistics pass (P1) instead

    def _moments_ptr(self, idx):
        m = moments_of(idx) if self.use_moments else None
        return None if m is None else m.data_ptr()

    def knn_rel_moments(self, nsample, xyz, offset, idx, new_xyz=None):
        """Per-scene sums (b, 9) float64 of rel = xyz[idx] - new_xyz over the rows of a kNN table (csrc/geom_moments.hip); ``offset`` =
        scene ends of the QUERIES; ``new_xyz`` None = a self table."""
        q = xyz if new_xyz is None else new_xyz
        _check(xyz, torch.float32, "xyz"); _check(q, torch.float32, "new_xyz"); _check(offset, torch.int32, "offset"); _check(idx, torch.int32, "idx")
        out = torch.empty((offset.shape[0], 9), dtype=torch.float64, device=xyz.device)
        ws = torch.empty((max(int(self.lib.pdf_knn_rel_moments_ws_doubles(int(offset.shape[0]), int(q.shape[0]))), 1),), dtype=torch.float64, device=xyz.device)
        require_current_device(xyz, q, offset, idx)
        rc = self.lib.pdf_knn_rel_moments_q(int(offset.shape[0]), int(q.shape[0]), int(nsample), xyz.data_ptr(), q.data_ptr(), offset.data_ptr(),
                                            idx.data_ptr(), out.data_ptr(), ws.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_knn_rel_moments failed with status {rc}")
        return out

    def _order_ptr(self, idx):
        o = order_of(idx)   # (always handed over: the g_xv gather visits its destinations in this order; the passes only with bit 2)
        return None if o is None else o.data_ptr()

    def pt_layer_backward(self, xq, xk, xv, p, idx, weights, bn, saved, H, gout, storage_bf16=None):
        n, c = xq.shape
        k = idx.shape[1]
        cs = c // 8
        _check(gout, torch.float32, "gout")
        require_current_device(xq, gout)
        gxq = self._new(xq, (n, c), torch.float32)
        gxk = self._new(xq, (n, c), torch.float32)
        gxv = self._new(xq, (n, c), torch.float32)
        G2 = self._new(xq, (n * k * cs,), torch.float32)
        G3 = self._new(xq, (n * k * 3,), torch.float32)
        Wsm = self._new(xq, (n * k * cs,), torch.float32)
        GR = self._new(xq, (n * k * c,), torch.float32)
        inv_off, inv_entry, entry_base = inverse_table(idx, n)
        partial = self._new(xq, (int(self.lib.pdf_pt_layer_bwd_partial_floats(n, k, c)),), torch.float32)
        nsum = int(self.lib.pdf_pt_layer_bwd_sums_floats(c))
        sums = self._new(xq, (nsum + 2 * (3 + c + cs),), torch.float32)
        rc = self.lib.pdf_pt_layer_backward(
            n, k, c, self._ptr(xq), self._ptr(xk), self._ptr(xv), self._ptr(p), self._ptr(idx),
            self._ptr_array(weights), self._ptr(bn), self._ptr(saved), self._ptr(H), self._ptr(gout),
            self._ptr(gxq), self._ptr(gxk), self._ptr(gxv), self._ptr(G2), self._ptr(G3), self._ptr(Wsm), self._ptr(GR),
            self._ptr(inv_off), self._ptr(inv_entry), int(entry_base), self._ptr(partial),
            self._ptr(sums), self.layer_flags(self.storage_bf16 if storage_bf16 is None else storage_bf16), self._order_ptr(idx),
            self._moments_ptr(idx), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_pt_layer_backward failed with status {rc}")
        # unpack the parameter-gradient sections (layout: csrc/fused_layer.hip, pdf_pt_layer_backward)
        o1 = 0
        o2 = o1 + 3 * cs + cs * cs
        o3 = o2 + 2 * c + cs + cs * c
        o4 = o3 + 8 + 4 * c
        g = dict(
            beta2=sums[o1:o1 + cs], gamma2=sums[o1 + cs:o1 + 2 * cs], bw2=sums[o1 + 2 * cs:o1 + 3 * cs],
            Ww2=sums[o1 + 3 * cs:o2].view(cs, cs),
            beta1=sums[o2:o2 + c], gamma1=sums[o2 + c:o2 + 2 * c], bw1=sums[o2 + 2 * c:o2 + 2 * c + cs],
            Ww1=sums[o2 + 2 * c + cs:o3].view(cs, c),
            betap=sums[o3:o3 + 3], gammap=sums[o3 + 3:o3 + 6], bp2=sums[o3 + 8:o3 + 8 + c],
            Wp2=sums[o3 + 8 + c:o4].view(c, 3),
            bp1=sums[o4:o4 + 3], Wp1=sums[o4 + 3:o4 + 12].view(3, 3),
        )
        return gxq, gxk, gxv, g

    # -- dense per-point Linear on the matrix cores (csrc/rowlin.hip) ---------------------------------------
    # -- libs/pointops2 window attention, backward: atomic-free segmented sums (csrc/window_attention_bwd.hip) ---------------------
    # PDFOPS_WA_ATOMICS=1: the fp32-atomic launchers of rounds 1-4 (csrc/window_attention.hip) -- A/B and fallback for other shapes
    wa_atomic_free = os.environ.get("PDFOPS_WA_ATOMICS", "0") != "1"

    def _wa_ok(self, d, L, *tensors):
        return self.wa_atomic_free and d == 16 and L <= 64 and all(t.data_ptr() % 16 == 0 for t in tensors)

    def _wa_rows(self, n, h, d, L, seg_off, seg_edge, other, rel, w, X, table, out, ldx=None, xscale=1.0, ldo=None, oscale=1.0, order=None):
        """pdf_wa_segment_rows[_ordered]; X / out may be column slices of wider rows (ldx / ldo = their row strides in floats); ``order``:
        visiting order of the owners (int32 permutation) or None."""
        c = h * d
        if order is not None and (order.shape[0] != n or order.dtype != torch.int32):
            order = None
        self._call("wa_segment_rows_ordered", n, h, d, L, seg_off, seg_edge, other, rel, w, X, c if ldx is None else int(ldx), float(xscale), table, out,
                   c if ldo is None else int(ldo), float(oscale), order)

    wa_window_order = os.environ.get("PDFOPS_WA_WINDOW_ORDER", "1") != "0"   # (A/B: 0 = owners in storage order)

    def _wa_order(self, offsets):
        return window_order_of(offsets) if self.wa_window_order else None

    def _wa_permute(self, w, edge):
        """w (M, h) float32 -> w[edge] (torch's index_select takes 25-95 us for these 12-96-byte rows; this is one pass at copy speed)"""
        w = w.contiguous()
        _check(w, torch.float32, "edge scalars"); _check(edge, torch.int32, "edge ids")
        out = torch.empty((edge.shape[0], w.shape[1]), dtype=torch.float32, device=w.device)
        self._call("wa_permute_edges", int(edge.shape[0]), int(w.shape[1]), w, edge, out)
        return out

    def _wa_table_grad(self, n, h, d, L, seg_off, seg_edge, rel, w, x, like, ldx=None, xscale=1.0):
        g = torch.empty((L, h, d, 3), dtype=torch.float32, device=like.device)
        ws = torch.empty((max(int(self.lib.pdf_wa_table_grad_ws_floats(n, h, L)), 1),), dtype=torch.float32, device=like.device)
        self._call("wa_table_grad", n, h, d, L, seg_off, seg_edge, rel, w, x, h * d if ldx is None else int(ldx), float(xscale), ws, g)
        return g

    def attention_step1_v2_backward(self, grad_out, q, k, index1, offsets, n_max):
        n, h, d = q.shape
        if index1.shape[0] == 0 or not self._wa_ok(d, 1, q, k):
            return super().attention_step1_v2_backward(grad_out, q, k, index1, offsets, n_max)
        _check(grad_out, torch.float32, "grad_output")
        key_off, key_edge, key_q = window_csc(index1, offsets, n_keys=k.shape[0])
        gq, gk = torch.empty_like(q), torch.empty_like(k)
        self._wa_rows(n, h, d, 0, offsets, None, index1, None, grad_out, k, None, gq)                      # grad_q = sum g k[index1]
        self._wa_rows(k.shape[0], h, d, 0, key_off, key_edge, key_q, None, grad_out, q, None, gk)          # grad_k = sum g q[query]
        return gq, gk

    def dot_prod_with_idx_v3_backward(self, grad_out, q, offsets, n_max, k, index_k, table_q, table_k, rel_idx):
        n, h, d = q.shape
        L = int(table_q.shape[0])
        if index_k.shape[0] == 0 or not self._wa_ok(d, L, q, k):
            return super().dot_prod_with_idx_v3_backward(grad_out, q, offsets, n_max, k, index_k, table_q, table_k, rel_idx)
        _check(grad_out, torch.float32, "grad_output")
        nk = k.shape[0]
        key_off, key_edge, _key_q, key_rel = window_csc(index_k, offsets, rel_idx, n_keys=nk)
        g_key = self._wa_permute(grad_out, key_edge)   # (the edge scalars in key order, once: two kernels read them sequentially)
        gq, gk = torch.empty_like(q), torch.empty_like(k)
        self._wa_rows(n, h, d, L, offsets, None, None, rel_idx, grad_out, None, table_q, gq)               # grad_q = sum g T_q
        self._wa_rows(nk, h, d, L, key_off, None, None, key_rel, g_key, None, table_k, gk)                 # grad_k = sum g T_k
        gtq = self._wa_table_grad(n, h, d, L, offsets, None, rel_idx, grad_out, q, q)
        gtk = self._wa_table_grad(nk, h, d, L, key_off, None, key_rel, g_key, k, q)
        return gq, gk, gtq, gtk

    def attention_step2_with_rel_pos_value_v2(self, attn, v, offsets, n_max, index1, table, rel_idx):
        """out[q] = sum over the query's edges of attn * (v[index1] + T): the same segmented pass as the backward's row sums (lane = (query,
        16-byte piece), 8 edges in flight) instead of the lane-per-channel walk of csrc/window_attention.hip (235 -> 120 us per call)."""
        m, h = attn.shape
        n, _, d = v.shape
        L = int(table.shape[0])
        if m == 0 or offsets.shape[0] != n + 1 or not self._wa_ok(d, L, v):
            return super().attention_step2_with_rel_pos_value_v2(attn, v, offsets, n_max, index1, table, rel_idx)
        for t, nm in ((attn, "attn"), (v, "v"), (table, "table")):
            _check(t, torch.float32, nm)
        for t, nm in ((offsets, "index0_offsets"), (index1, "index1"), (rel_idx, "rel_idx")):
            _check(t, torch.int32, nm)
        if tuple(table.shape[1:]) != (h, d, 3):
            raise ValueError("attention_step2_with_rel_pos_value_v2: inconsistent shapes")
        out = torch.empty((n, h, d), dtype=torch.float32, device=v.device)
        self._wa_rows(n, h, d, L, offsets, None, index1, rel_idx, attn, v, table, out)
        return out

    def window_keys(self, xyz, ends, lo, hi, window_size, parity):
        """(kf, kc, wk) int64 per point of one window partition (csrc/window_edges.hip we::k_keys; what stratified.window_keys composes from
        ~45 torch ops): xyz (N, 3), ends (scenes) int32 scene ends, lo / hi (3) float32 = per-axis minimum / maximum of xyz."""
        n = int(xyz.shape[0])
        out = torch.empty((3, n), dtype=torch.int64, device=xyz.device)
        self._call("window_keys", n, xyz.contiguous(), ends.int().contiguous(), int(ends.shape[0]), lo.float().contiguous(), hi.float().contiguous(),
                   float(window_size), int(parity), out[0], out[1], out[2])
        return out[0], out[1], out[2]

    def window_edges(self, xyz, kf, kc, wk, downsample_idx, c2w, qs, vmax):
        """The CSR-by-query edge table of one window partition (csrc/window_edges.hip; stratified_transformer_v1m1_origin.py:45-127 + the
        sort by query of :468-536 + :282-292).  kf / kc / wk (N) int64: fine-window key, coarse-window key, packed fine-window cell of every
        point; downsample_idx: the FPS key subset.  -> index_0 (E) int64 ascending, index_1 (E) int32, offsets (N + 1) int32, n_max (int),
        rel_idx (E, 3) int32, flag (1) int32 device tensor (non-zero: a quantised offset left [0, vmax] -- the caller asserts).
        Two point-sized stable sorts (torch / rocPRIM), two launches, ONE host read (edge count + longest row)."""
        n, dev = int(xyz.shape[0]), xyz.device
        i32 = dict(dtype=torch.int32, device=dev)
        kf, kc, wk = kf.contiguous(), kc.contiguous(), wk.contiguous()
        kf_sorted, order_f = torch.sort(kf, stable=True)
        ds = torch.sort(downsample_idx.long())[0]
        kcd_sorted, perm = torch.sort(kc[ds], stable=True)
        order_cd = ds[perm]
        wkd = wk[order_cd].contiguous()
        m = int(ds.shape[0])
        count = torch.empty(n, **i32)
        seg = torch.empty((n, 4), **i32)
        self._call("window_edges_count", n, kf_sorted, kf, m, kcd_sorted, kc, wk, wkd, count, seg)
        offsets = torch.zeros(n + 1, **i32)
        torch.cumsum(count, 0, dtype=torch.int32, out=offsets[1:])
        e, n_max = (int(v) for v in torch.stack([offsets[-1], count.max() if n else offsets[-1]]).tolist())   # the one host read
        index0 = torch.empty(e, dtype=torch.int64, device=dev)
        index1 = torch.empty(e, **i32)
        rel = torch.empty((e, 3), **i32)
        flag = torch.zeros(1, **i32)
        order32 = order_f.int()
        self._call("window_edges_fill", n, offsets, seg, order32, order_cd.int(), wk, wkd, xyz.contiguous(), float(c2w), float(qs), int(vmax),
                   index0, index1, rel, flag)
        # the queries window by window: the visiting order of the attention's segmented row passes (queries / keys of one window gather
        # the same rows: pdf_wa_segment_rows_ordered)
        setattr(offsets, _WORDER, (offsets.data_ptr(), offsets._version, order32))
        return index0, index1, offsets, n_max, rel, flag

    def window_logits_supported(self, q, k, table_q):
        return self.wa_atomic_free and q.shape[0] == k.shape[0] and self._wa_ok(q.shape[2], int(table_q.shape[0]), q, k)

    def window_logits(self, q, k, index1, offsets, table_q, table_k, rel_idx):
        """attention_step1_v2(q, k) + dot_prod_with_idx_v3(q, k, table_q, table_k) as ONE pass over the key rows -> (M, h)."""
        n, h, d = q.shape
        m = index1.shape[0]
        out = torch.empty((m, h), dtype=torch.float32, device=q.device)
        self._call("wa_logits_forward", n, m, h, d, int(table_q.shape[0]), q, k, h * d, 1.0, offsets, index1, table_q, table_k, rel_idx, out)
        return out

    def window_logits_backward(self, g, q, k, index1, offsets, table_q, table_k, rel_idx):
        """-> (grad_q, grad_k, grad_table_q, grad_table_k): the sums of the two ops' gradients, each as ONE segmented pass (rows + table term
        together); the edge scalars are brought into key order once for the two key-side kernels."""
        n, h, d = q.shape
        L = int(table_q.shape[0])
        key_off, key_edge, key_q, key_rel = window_csc(index1, offsets, rel_idx, n_keys=n)
        g_key = self._wa_permute(g, key_edge)
        gq, gk = torch.empty_like(q), torch.empty_like(k)
        self._wa_rows(n, h, d, L, offsets, None, index1, rel_idx, g, k, table_q, gq)          # sum g (k[index1] + T_q)
        self._wa_rows(n, h, d, L, key_off, None, key_q, key_rel, g_key, q, table_k, gk)       # sum g (q[query] + T_k)
        gtq = self._wa_table_grad(n, h, d, L, offsets, None, rel_idx, g, q, q)
        gtk = self._wa_table_grad(n, h, d, L, key_off, None, key_rel, g_key, k, q)
        return gq, gk, gtq, gtk

    def attention_step2_with_rel_pos_value_v2_backward(self, grad_out, attn, v, offsets, n_max, index1, table, rel_idx):
        m, h = attn.shape
        n, _, d = v.shape
        L = int(table.shape[0])
        if m == 0 or offsets.shape[0] != n + 1 or not self._wa_ok(d, L, v, grad_out):
            return super().attention_step2_with_rel_pos_value_v2_backward(grad_out, attn, v, offsets, n_max, index1, table, rel_idx)
        _check(grad_out, torch.float32, "grad_output")
        c = h * d
        key_off, key_edge, key_q = window_csc(index1, offsets, n_keys=n)
        ga = torch.empty((m, h), dtype=torch.float32, device=v.device)
        gv = torch.empty_like(v)
        self._call("wa_grad_attn", n, m, h, d, L, grad_out, c, offsets, index1, v, c, table, rel_idx, ga)
        self._wa_rows(n, h, d, 0, key_off, key_edge, key_q, None, attn, grad_out, None, gv)                # grad_v = sum attn grad_out[query]
        gt = self._wa_table_grad(n, h, d, L, offsets, None, rel_idx, attn, grad_out, v)
        return ga, gv, gt

    # The whole attention core of WindowAttention.forward (stratified_transformer_v1m1_origin.py:296-341) on the (N, 3 C) output of the
    # qkv Linear: q / k / v are read as column slices (no permute copy), the query scale rides on the kernels, and the backward writes
    # the three gradients into the slices of one (N, 3 C) buffer.
    wa_core = os.environ.get("PDFOPS_WA_CORE", "1") != "0"   # (A/B: 0 = the model composes window_logits / softmax / step2 itself)

    def window_attention_core_supported(self, qkv, table_q, table_k, table_v):
        L, h, d, _ = table_q.shape
        return (self.wa_core and self.wa_atomic_free and qkv.dim() == 2 and qkv.shape[1] == 3 * h * d and table_k.shape == table_q.shape == table_v.shape
                and qkv.dtype == torch.float32 and qkv.is_contiguous() and self._wa_ok(d, L, qkv) and (h * d) % 4 == 0)

    def window_attention_core(self, qkv, index1, offsets, table_q, table_k, table_v, rel_idx, scale):
        """-> (out (N, C), attn (M, h) after the softmax)."""
        n = qkv.shape[0]
        L, h, d, _ = table_q.shape
        c, m = h * d, index1.shape[0]
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        logits = torch.empty((m, h), dtype=torch.float32, device=qkv.device)
        self._call("wa_logits_forward_ordered", n, m, h, d, L, q, k, 3 * c, float(scale), offsets, index1, table_q, table_k, rel_idx, logits,
                   self._wa_order(offsets))
        attn = self.segment_softmax(logits, offsets)
        out = torch.empty((n, c), dtype=torch.float32, device=qkv.device)
        self._wa_rows(n, h, d, L, offsets, None, index1, rel_idx, attn, v, table_v, out, ldx=3 * c, order=self._wa_order(offsets))
        return out, attn

    def window_attention_core_backward(self, go, qkv, attn, index1, offsets, table_q, table_k, table_v, rel_idx, scale):
        """-> (grad_qkv (N, 3 C), grad_table_q, grad_table_k, grad_table_v)"""
        n = qkv.shape[0]
        L, h, d, _ = table_q.shape
        c, m = h * d, index1.shape[0]
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        key_off, key_edge, key_q, key_rel = window_csc(index1, offsets, rel_idx, n_keys=n)
        gqkv = torch.empty_like(qkv)
        ga = torch.empty((m, h), dtype=torch.float32, device=qkv.device)
        order = self._wa_order(offsets)   # (query side: window by window)
        self._call("wa_grad_attn_ordered", n, m, h, d, L, go, c, offsets, index1, v, 3 * c, table_v, rel_idx, ga, order)
        attn_key = self._wa_permute(attn, key_edge)
        korder = window_key_order(index1) if self.wa_window_order else None   # (key side: longest rows first, window by window)
        self._wa_rows(n, h, d, 0, key_off, None, key_q, None, attn_key, go, None, gqkv[:, 2 * c:], ldo=3 * c, order=korder)          # grad_v
        gtv = self._wa_table_grad(n, h, d, L, offsets, None, rel_idx, attn, go, qkv)
        g = self.segment_softmax_backward(attn, ga, offsets)
        g_key = self._wa_permute(g, key_edge)
        self._wa_rows(n, h, d, L, offsets, None, index1, rel_idx, g, k, table_q, gqkv[:, :c], ldx=3 * c, ldo=3 * c, oscale=scale, order=order)      # grad_q
        self._wa_rows(n, h, d, L, key_off, None, key_q, key_rel, g_key, q, table_k, gqkv[:, c:2 * c], ldx=3 * c, xscale=scale, ldo=3 * c, order=korder)   # grad_k
        gtq = self._wa_table_grad(n, h, d, L, offsets, None, rel_idx, g, q, qkv, ldx=3 * c, xscale=scale)
        gtk = self._wa_table_grad(n, h, d, L, key_off, None, key_rel, g_key, k, qkv, ldx=3 * c)
        return gqkv, gtq, gtk, gtv

    def _stream(self):
        return c_void_p(raw_stream())

    def rowlin(self, x, w, bias=None, coef=None, relu=False, transpose_w=False, out=None, accumulate=False, stats=False):
        """y = f(x) @ Wt + bias (see include/pdfops.h); x (n,k) with row stride x.stride(0); returns (y, partial|None)."""
        n, k = x.shape
        o = w.shape[1] if transpose_w else w.shape[0]
        y = out if out is not None else torch.empty((n, o), dtype=torch.float32, device=x.device)
        partial = None
        if stats:
            partial = torch.empty((int(self.lib.pdf_rowlin_partial_floats(n, o)),), dtype=torch.float32, device=x.device)
            partial._pdf_rows = int(self.lib.pdf_rowlin_partial_rows(n, k, o))
        rc = self.lib.pdf_rowlin_forward(n, k, o, x.data_ptr(), x.stride(0), w.data_ptr(), int(transpose_w),
                                         None if bias is None else bias.data_ptr(),
                                         None if coef is None else coef.data_ptr(),
                                         None if coef is None else coef.data_ptr() + 4 * k, int(relu), y.data_ptr(), y.stride(0),
                                         int(accumulate), None if partial is None else partial.data_ptr(), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_forward failed with status {rc}")
        return y, partial

    @staticmethod
    def _ptrs(tensors, n=3):
        return (c_void_p * n)(*[t.data_ptr() if t is not None else None for t in list(tensors) + [None] * (n - len(tensors))])

    def rowlin_multi(self, xs, ws, biases=None, coef=None, relu=False, transpose_w=False, nout=1):
        """nout = 3: [f(xs[0]) @ Wt_i + b_i]; nout = 1: [sum_i xs[i] @ Wt_i (+ b_0)]   (pdf_rowlin_multi)"""
        n, k = xs[0].shape
        o = ws[0].shape[1] if transpose_w else ws[0].shape[0]
        ys = [torch.empty((n, o), dtype=torch.float32, device=xs[0].device) for _ in range(nout)]
        rc = self.lib.pdf_rowlin_multi(n, k, o, len(xs), nout, self._ptrs(xs), xs[0].stride(0), self._ptrs(ws), int(transpose_w),
                                       None if biases is None else self._ptrs(biases),
                                       None if coef is None else coef.data_ptr(), None if coef is None else coef.data_ptr() + 4 * k,
                                       int(relu), self._ptrs(ys), o, 0, current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_multi failed with status {rc}")
        return ys

    def rowlin_dgrad_bn_backward(self, gs, ws, bx, coef, training=True, relu=True):
        """Input gradient sum_i gs[i] @ ws[i] of Linear layers reading relu(bn(bx)), followed by that BatchNorm's backward with the
        reduction pass folded into the product's epilogue (pdf_rowlin_dgrad_bstats + pdf_bn_act_backward_presummed).
        -> (d bx, d gamma, d beta), or None when the streaming kernels do not cover the shape."""
        n, k = gs[0].shape
        o = ws[0].shape[1]
        require_current_device(bx, *gs)
        dy = self._new(bx, (n, o), torch.float32)
        partial = self._new(bx, (int(self.lib.pdf_rowlin_partial_floats(n, o)),), torch.float32)
        rows = c_int(0)
        sums = self._new(bx, (2 * o,), torch.float32)
        rc = self.lib.pdf_rowlin_dgrad_bstats(n, k, o, len(gs), self._ptrs(gs), gs[0].stride(0), self._ptrs(ws), dy.data_ptr(), o,
                                              bx.data_ptr(), bx.stride(0), coef.data_ptr(), int(bool(relu)), partial.data_ptr(),
                                              ctypes.byref(rows), current_mma_input(), self._stream())
        if rc == -3:   # PDF_ERR_UNSUPPORTED
            return None
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_dgrad_bstats failed with status {rc}")
        rc = self.lib.pdf_bn_act_backward_presummed(n, o, dy.data_ptr(), bx.data_ptr(), coef.data_ptr(), int(bool(training)), int(bool(relu)),
                                                    partial.data_ptr(), rows.value, sums.data_ptr(), dy.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_bn_act_backward_presummed failed with status {rc}")
        return dy, sums[o:], sums[:o]

    def wgrad_workspace(self, n, k, o, ng, device):
        """Slab workspace of the weight-gradient entry points (fixed-order reduction instead of float atomics)."""
        return torch.empty((max(int(self.lib.pdf_rowlin_wgrad_ws_floats(int(n), int(k), int(o), int(ng))), 1),), dtype=torch.float32, device=device)

    def rowlin_wgrad_multi(self, gs, x, coef, relu):
        n, o = gs[0].shape
        k = x.shape[1]
        dws = [torch.empty((o, k), dtype=torch.float32, device=x.device) for _ in gs]
        dbs = [torch.empty((o,), dtype=torch.float32, device=x.device) for _ in gs]
        ws = self.wgrad_workspace(n, k, o, len(gs), x.device)
        rc = self.lib.pdf_rowlin_wgrad_multi(n, k, o, len(gs), self._ptrs(gs), gs[0].stride(0), x.data_ptr(), x.stride(0),
                                             None if coef is None else coef.data_ptr(), None if coef is None else coef.data_ptr() + 4 * k,
                                             int(relu), self._ptrs(dws), self._ptrs(dbs), ws.data_ptr(), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_wgrad_multi failed with status {rc}")
        return dws, dbs

    # -- rigid KPConv of the StratifiedTransformer stem (csrc/kpconv.hip): gather-accumulate and its adjoint
    def kpconv_supported(self, kp, cin):
        return bool(self.lib.pdf_kpconv_supported(int(kp), int(cin)))

    def kpconv_gather(self, query, support, neighbors, x, k_points, extent):
        """-> weighted (N, KP * C_in): per query and kernel point the influence-weighted sum of its neighbours' feature rows."""
        n, m = neighbors.shape
        kp, cin = k_points.shape[0], x.shape[1]
        out = torch.empty((n, kp * cin), dtype=torch.float32, device=x.device)
        rc = self.lib.pdf_kpconv_gather(n, m, kp, cin, query.data_ptr(), support.data_ptr(), neighbors.data_ptr(), x.data_ptr(),
                                        k_points.data_ptr(), ctypes.c_float(extent), out.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_kpconv_gather failed with status {rc}")
        return out

    def kpconv_scatter(self, query, support, neighbors, grad_weighted, k_points, extent, rows, cin):
        """-> grad_x (rows, C_in): the adjoint of kpconv_gather."""
        n, m = neighbors.shape
        gx = torch.zeros((rows, cin), dtype=torch.float32, device=grad_weighted.device)
        rc = self.lib.pdf_kpconv_scatter(n, m, k_points.shape[0], cin, query.data_ptr(), support.data_ptr(), neighbors.data_ptr(),
                                         grad_weighted.data_ptr(), k_points.data_ptr(), ctypes.c_float(extent), gx.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_kpconv_scatter failed with status {rc}")
        return gx

    def rowlin_wgrad_group(self, gs, xs, coefs, relus, need_bias):
        """Up to five weight gradients of one shape with their own inputs in one launch + one reduction (include/pdfops.h:
        pdf_rowlin_wgrad_group): dW_i = G_i^T f_i(X_i), f_i = relu_i?(x * scale_i + shift_i) for a coef_i = [scale | shift | ...], else the
        identity.  -> (list of dW, list of db | None); None when the shape is outside the streaming kernels."""
        n, o = gs[0].shape
        k = xs[0].shape[1]
        dev = gs[0].device
        dws = [torch.empty((o, k), dtype=torch.float32, device=dev) for _ in gs]
        dbs = [torch.empty((o,), dtype=torch.float32, device=dev) if nb else None for nb in need_bias]
        ws = self.wgrad_workspace(n, k, o, len(gs), dev)
        P = lambda ts: (c_void_p * len(ts))(*[None if t is None else (t if isinstance(t, int) else t.data_ptr()) for t in ts])
        sc = [None if cf is None else cf.data_ptr() for cf in coefs]
        sh = [None if cf is None else cf.data_ptr() + 4 * k for cf in coefs]
        rc = self.lib.pdf_rowlin_wgrad_group(n, k, o, len(gs), P(gs), gs[0].stride(0), P(xs), xs[0].stride(0), P(sc), P(sh),
                                             (c_int * len(gs))(*[int(bool(r)) for r in relus]), P(dws), P(dbs), ws.data_ptr(), current_mma_input(), self._stream())
        if rc == -3:   # PDF_ERR_UNSUPPORTED
            return None
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_wgrad_group failed with status {rc}")
        return dws, dbs

    def rowlin_wgrad(self, g, x, coef, relu, need_bias):
        n, o = g.shape
        k = x.shape[1]
        dw = torch.empty((o, k), dtype=torch.float32, device=x.device)
        db = torch.empty((o,), dtype=torch.float32, device=x.device) if need_bias else None
        ws = self.wgrad_workspace(n, k, o, 1, x.device)
        rc = self.lib.pdf_rowlin_wgrad(n, k, o, g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0),
                                       None if coef is None else coef.data_ptr(),
                                       None if coef is None else coef.data_ptr() + 4 * k, int(relu), dw.data_ptr(),
                                       None if db is None else db.data_ptr(), ws.data_ptr(), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_rowlin_wgrad failed with status {rc}")
        return dw, db

    def bn_coef_from_partial(self, partial, n, c, bn, training):
        """coef (4c) = scale|shift|mean|rstd of BatchNorm ``bn`` from rowlin column partials (training) or running stats."""
        coef = torch.empty((4 * c,), dtype=torch.float32, device=partial.device if partial is not None else bn.weight.device)
        if training:
            rows = partial._pdf_rows
            rc = self.lib.pdf_bn_coef_from_partial(partial.data_ptr(), rows, n, c, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                   bn.running_mean.data_ptr(), bn.running_var.data_ptr(), ctypes.c_float(bn.eps),
                                                   ctypes.c_float(bn.momentum),
                                                   coef.data_ptr(), self._stream())
            if rc != 0:
                raise PdfOpsError(f"pdf_bn_coef_from_partial failed with status {rc}")
        else:
            rstd = torch.rsqrt(bn.running_var + bn.eps)
            sc = bn.weight.detach() * rstd
            coef = torch.cat([sc, bn.bias.detach() - bn.running_mean * sc, bn.running_mean, rstd])
        return coef

    def bn_coef(self, x, bn, training):
        """coef (4c) of BatchNorm ``bn`` over the rows of x: batch statistics (training, updates running stats) or running."""
        n, c = x.shape
        coef = torch.empty((4 * c,), dtype=torch.float32, device=x.device)
        partial = torch.empty((int(self.lib.pdf_bn_partial_floats(n, c)),), dtype=torch.float32, device=x.device) if training else None
        rc = self.lib.pdf_bn_coef(n, c, x.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                  bn.running_var.data_ptr(), int(bool(training)), ctypes.c_float(bn.eps),
                                  ctypes.c_float(bn.momentum), coef.data_ptr(),
                                  None if partial is None else partial.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_bn_coef failed with status {rc}")
        return coef

    def bn_apply(self, x, res, coef, relu):
        n, c = x.shape
        y = torch.empty_like(x)
        rc = self.lib.pdf_bn_apply(n, c, x.data_ptr(), None if res is None else res.data_ptr(), coef.data_ptr(), int(relu),
                                   y.data_ptr(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_bn_apply failed with status {rc}")
        return y

    def td_tables(self, p_src, p_new, idx, new_offset, inverse=None):
        """Geometry-only tables of the fused TransitionDown: rel4 (m,16,4), Z (n,32), scene_sums (b,16) (csrc/transition_down.hip).
        ``inverse`` = (inv_off, inv_entry, entry_base) of idx: Z is then summed in destination order (bit-reproducible)."""
        m, n, b = p_new.shape[0], p_src.shape[0], new_offset.shape[0]
        rel4 = torch.empty((m, 16, 4), dtype=torch.float32, device=p_src.device)
        Z = torch.zeros((n, 32), dtype=torch.float32, device=p_src.device)
        sums = torch.zeros((b, 16), dtype=torch.float32, device=p_src.device)
        off, ent, base = inverse if inverse is not None else (None, None, 0)
        rc = self.lib.pdf_td_tables(m, b, p_src.data_ptr(), p_new.data_ptr(), idx.data_ptr(), new_offset.data_ptr(), rel4.data_ptr(), Z.data_ptr(),
                                    sums.data_ptr(), n, None if off is None else off.data_ptr(), None if ent is None else ent.data_ptr(), int(base),
                                    self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_td_tables failed with status {rc}")
        return rel4, Z, sums

    # -- whole Bottleneck as one host call per direction (csrc/block.hip); thin methods so that bench.py can time them
    def bottleneck_forward(self, n, k, c, ptrs, training, eps, momentum, storage_bf16=0):
        rc = self.lib.pdf_bottleneck_forward(n, k, c, (c_void_p * len(ptrs))(*ptrs), int(training), ctypes.c_float(eps),
                                             ctypes.c_float(momentum), self.layer_flags(storage_bf16), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_bottleneck_forward failed with status {rc}")

    def bottleneck_backward(self, n, k, c, ptrs, training, entry_base=0, storage_bf16=0):
        rc = self.lib.pdf_bottleneck_backward(n, k, c, (c_void_p * len(ptrs))(*ptrs), int(training), int(entry_base), self.layer_flags(storage_bf16), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_bottleneck_backward failed with status {rc}")

    # -- Bottleneck halves as single host calls (csrc/block.hip) -------------------------------------------
    def _ptable(self, tensors):
        return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])

    def block_call(self, name, n, c, tensors, training, eps=None, momentum=None):
        fn = getattr(self.lib, "pdf_block_" + name)
        if eps is None:
            rc = fn(n, c, self._ptable(tensors), int(bool(training)), current_mma_input(), self._stream())
        else:
            rc = fn(n, c, self._ptable(tensors), int(bool(training)), ctypes.c_float(eps), ctypes.c_float(momentum), current_mma_input(), self._stream())
        if rc != 0:
            raise PdfOpsError(f"pdf_block_{name} failed with status {rc}")

    # -- BatchNorm + residual + ReLU over (n, c) rows ------------------------------------------------------
    def bn_supported(self, c):
        return bool(self.lib.pdf_bn_supported(int(c)))

    def bn_act_forward(self, x, res, gamma, beta, running_mean, running_var, training, eps, momentum, relu):
        n, c = x.shape
        require_current_device(x, res, gamma)
        coef = self._new(x, (4 * c,), torch.float32)
        partial = self._new(x, (int(self.lib.pdf_bn_partial_floats(n, c)),), torch.float32)
        y = self._new(x, (n, c), torch.float32)
        P = lambda t: None if t is None else self._ptr(t)
        rc = self.lib.pdf_bn_act_forward(n, c, P(x), P(res), P(gamma), P(beta), P(running_mean), P(running_var),
                                         int(bool(training)), ctypes.c_float(eps), ctypes.c_float(momentum), int(bool(relu)),
                                         P(coef), P(partial), P(y), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_bn_act_forward failed with status {rc}")
        return y, coef

    def bn_act_backward(self, gy, x, res, coef, training, relu, need_res):
        n, c = x.shape
        require_current_device(gy, x)
        partial = self._new(x, (int(self.lib.pdf_bn_partial_floats(n, c)),), torch.float32)
        sums = self._new(x, (2 * c,), torch.float32)
        gx = self._new(x, (n, c), torch.float32)
        gres = self._new(x, (n, c), torch.float32) if need_res else None
        P = lambda t: None if t is None else self._ptr(t)
        rc = self.lib.pdf_bn_act_backward(n, c, P(gy), P(x), P(res), P(coef), int(bool(training)), int(bool(relu)), P(partial),
                                          P(sums), P(gx), P(gres), c_void_p(raw_stream()))
        if rc != 0:
            raise PdfOpsError(f"pdf_bn_act_backward failed with status {rc}")
        return gx, gres, sums[c:], sums[:c]  # gx, gres, d gamma, d beta

    def radius_neighbors_self(self, nsample, radius, xyz, offset):
        """-> idx (n, nsample) int32, dist2 (n, nsample): the first nsample points of the scene in index order within radius."""
        _check(xyz, torch.float32, "xyz"); _check(offset, torch.int32, "offset")
        n, b = xyz.shape[0], offset.shape[0]
        idx = self._new(xyz, (n, nsample), torch.int32)
        dist2 = self._new(xyz, (n, nsample), torch.float32)
        if b > 64:   # the grid workspace is sized for <= 64 scenes: in-order scan
            order = torch.arange(n, dtype=torch.int32, device=xyz.device)
            return self.ball_query(nsample, radius, 0.0, xyz, xyz, offset, offset, order=order)
        nbytes = int(self.lib.pdf_knn_workspace_bytes(b, n, 0))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
        self._call("radius_neighbors_self", n, int(nsample), float(radius), xyz, offset, b, idx, dist2, ws, nbytes)
        return idx, dist2

    def graph_forest(self, n, eu, ev, nodes, weight=None, active=None, want_chosen=True):
        """One scene's region graph (pseudo-label pass, pointpdf_v1m1_base.py:309-380): directed entries (eu[e], ev[e]) with ``weight[e]``
        (None: all equal) among the ``active`` ones (bool / uint8, None: all); ``nodes``: ids covering every endpoint (repeats allowed).
        -> (chosen (E,) bool or None: the entries of the minimum spanning forest under the order (weight, entry index);
            comp (n,) int32: the root of the component of every listed node, the node's own id elsewhere)."""
        _check(eu, torch.int64, "eu"); _check(ev, torch.int64, "ev"); _check(nodes, torch.int64, "nodes")
        E = int(eu.shape[0])
        if int(ev.shape[0]) != E or (weight is not None and int(weight.shape[0]) != E) or (active is not None and int(active.shape[0]) != E):
            raise ValueError("graph_forest: eu, ev, weight, active disagree on the number of entries")
        if weight is not None:
            _check(weight, torch.float32, "weight")
        if active is not None:
            active = active.view(torch.uint8) if active.dtype == torch.bool else active
            _check(active, torch.uint8, "active")
        comp = torch.arange(int(n), dtype=torch.int32, device=eu.device)
        chosen = torch.empty((E,), dtype=torch.uint8, device=eu.device) if want_chosen else None
        nbytes = int(self.lib.pdf_graph_forest_workspace_bytes(int(n), E, int(nodes.shape[0])))
        ws = torch.empty((nbytes // 8 + 1,), dtype=torch.int64, device=eu.device)
        nul = ctypes.c_void_p(None)
        self._call("graph_forest", int(n), E, eu, ev, nul if weight is None else weight, nul if active is None else active, nodes,
                   int(nodes.shape[0]), comp, nul if chosen is None else chosen, ws, ws.numel() * 8)
        return (chosen.view(torch.bool) if chosen is not None else None), comp

    def gmm2_1d(self, x, iters=100, tol=1e-3, reg=1e-6):
        """Two-component 1-D Gaussian mixture of the float values ``x`` by EM in double, on the device (stands in for
        sklearn.mixture.GaussianMixture(n_components=2).fit, pointpdf_v1m1_base.py:343-345).  -> (8,) float64 on the device: means (2),
        variances (2), weights (2), iterations run, final mean log-likelihood."""
        _check(x, torch.float32, "x")
        xs = torch.sort(x.reshape(-1))[0]
        m = int(xs.shape[0])
        resp = torch.empty((max(2 * m, 1),), dtype=torch.float64, device=x.device)
        out = torch.empty((8,), dtype=torch.float64, device=x.device)
        self._call("gmm2_1d", m, xs, resp, out, int(iters), float(tol), float(reg))
        return out

    def grid_hash(self, coord, offset, grid_size, min_grid, float32_division=False):
        """-> grid (n,3) int64 scene-relative voxel coordinates, key (n) int64 holding the uint64 FNV key bits.
        GridSample front half, pointcept/datasets/transform.py:813-823, 911-925."""
        _check(coord, torch.float32, "coord"); _check(offset, torch.int32, "offset"); _check(min_grid, torch.int64, "min_grid")
        n = coord.shape[0]
        grid = self._new(coord, (n, 3), torch.int64)
        key = self._new(coord, (n,), torch.int64)
        gx, gy, gz = (float(g) for g in grid_size)
        self._call("grid_hash", n, offset.shape[0], coord, offset, gx, gy, gz, 1 if float32_division else 0, min_grid, grid, key)
        return grid, key

    def vote_accumulate(self, logits, score, index, pred, score_sum, score_cnt):
        """One test-time fragment into the running vote (engines/test.py:218-229, 243-251); index entries must be distinct."""
        _check(logits, torch.float32, "logits"); _check(index, torch.int64, "index"); _check(pred, torch.float32, "pred")
        if score is not None:
            _check(score, torch.float32, "score"); _check(score_sum, torch.float32, "score_sum"); _check(score_cnt, torch.float32, "score_cnt")
        n, c = logits.shape
        if pred.shape[1] != c:
            raise ValueError("pred and logits disagree on the number of classes")
        nul = ctypes.c_void_p(None)
        self._call("vote_accumulate", n, c, logits, nul if score is None else score, index, pred,
                   nul if score is None else score_sum, nul if score is None else score_cnt)

    def group_forward(self, feat, xyz, new_xyz, idx, with_xyz):
        _check(feat, torch.float32, "feat"); _check(idx, torch.int32, "idx")
        m, ns = idx.shape
        c = feat.shape[1]
        out = self._new(feat, (m, ns, c + (3 if with_xyz else 0)), torch.float32)
        if with_xyz:
            _check(xyz, torch.float32, "xyz"); _check(new_xyz, torch.float32, "new_xyz")
            self._call("group_forward_ordered", m, ns, c, 1, feat, xyz, new_xyz, idx, order_of(idx), out)
        else:
            self._call("group_forward_ordered", m, ns, c, 0, feat, feat, feat, idx, order_of(idx), out)
        return out

    def group_backward(self, grad_output, idx, n, c, with_xyz):
        _check(grad_output, torch.float32, "grad_output")
        if not with_xyz and self.use_inverse:
            return self.grouping_backward(grad_output, idx, n)
        m, ns = idx.shape
        if with_xyz and self.use_inverse and grad_output.numel() > 0 and n > 0:
            # rows are [g_rel_xyz (3) | g_feat (c)]: the feature part is summed per source point over the inverse table
            off, ent, base = inverse_table(idx, n)
            gf = self._new(grad_output, (n, c), torch.float32)
            self._call("seg_sum_rows_strided", n, c, grad_output.view(-1)[3:], 3 + c, off, ent, base, 1.0, gf)
            return gf
        gf = self._new(grad_output, (n, c), torch.float32, zero=True)
        self._call("group_backward", m, ns, c, 1 if with_xyz else 0, grad_output, idx, gf)
        return gf

    def interpolation_weights(self, dist2):
        _check(dist2, torch.float32, "dist2")
        n, k = dist2.shape
        w = self._new(dist2, (n, k), torch.float32)
        self._call("interpolation_weights", n, k, dist2, w)
        return w


_lock = threading.Lock()
_hip = None
_override = None  # set only by tests / the CPU-baseline leg (oracle injection)


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise PdfOpsError(
            f"pointcloudpdf_amd: {path} is missing. Build it with `python -m pointcloudpdf_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no fallback implementation."
        )
    return ctypes.CDLL(path)


def hip_backend():
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                _hip = HipBackend(load_library())
    return _hip


# Input precision of the matrix-core products of the streaming Linear kernels (include/pdfops.h: `mma_input`): 0 fp32 operands (the
# parity path), 1 fp16, 2 bfloat16 -- fp32 storage and accumulation in every mode.  The mode is an ARGUMENT of every C entry point that
# runs those products; on the Python side it is state of the CALLING THREAD (a thread-local, so a forward on one thread and an autograd
# backward on another never see each other's mode -- round 3 kept it in a process-wide word inside the library).  ``dense.fp32_path``
# selects 1 / 2 for a forward that runs under torch.autocast(float16 / bfloat16); every autograd node remembers the mode of its forward
# and runs its backward in it.
_MMA = threading.local()
MMA_INPUT_OF_DTYPE = {torch.float16: 1, torch.bfloat16: 2}


def current_mma_input():
    return getattr(_MMA, "mode", 0)


class mma_input:
    """``with mma_input(mode):`` -- product-input mode of the launches this THREAD issues inside the block."""

    def __init__(self, mode):
        self.mode = int(mode)
        if self.mode not in (0, 1, 2):
            raise PdfOpsError(f"mma_input({mode}): expected 0 (fp32), 1 (fp16) or 2 (bfloat16) operands")

    def __enter__(self):
        self.prev = current_mma_input()
        _MMA.mode = self.mode
        return self

    def __exit__(self, *exc):
        _MMA.mode = self.prev
        return False


def backend_for(t):
    """The backend that must serve tensor ``t``.  CPU tensors raise unless a test injected a backend."""
    if _override is not None:
        return _override
    if t.device.type == "cuda":
        return hip_backend()
    raise PdfOpsError(
        f"pointcloudpdf_amd: got a tensor on '{t.device}'. The product path is HIP-only (MI355X); "
        "move inputs to a ROCm device."
    )


def _set_backend_for_testing(backend):
    """TEST / CPU-BASELINE ONLY: route every op through ``backend`` (e.g. the CPU oracle). Returns the previous one."""
    global _override
    prev, _override = _override, backend
    return prev
