"""Geometry pre-pass: everything on the PointTransformer path that depends on coordinates only.

The reference recomputes identical kNN queries 2-6x per level and calls FPS / interpolation-kNN from
inside the layers (4 FPS + 31 kNN calls per forward, 13 distinct).  None of those depend on
features, so one ``Geometry`` object per batch memoises them:

    level l      : coords p_l (N_l,3), cumulative offsets o_l (device int32 + host list)
    down(l)      : FPS indices of level l  -> level l+1   (point_transformer_seg.py:96-102)
    knn(k, a, b) : kNN table of level-b queries over level-a points (query.py:9-24)
    interp(a, b) : 3-NN indices + inverse-distance weights coarse a -> fine b (interpolation.py:14-17)

Results are bit-identical to calling the ops every time (same kernels, same inputs).  Coordinate tensors
handed out by a Geometry carry a tag so that the ``pointops`` functions recognise them and hit the memo
table; untagged tensors simply take the uncached path.  Host copies of the offsets remove the
``.item()`` syncs of the reference's TransitionDown / FPS wrapper (point_transformer_seg.py:96-99,
libs/pointops/functions/sampling.py:15-18).
"""
import weakref

import os

import torch

from . import _native

_TAG = "_pdf_geom_tag"


def tag_of(t):
    """(geometry, level) if ``t`` is an unmodified coordinate tensor handed out by a live Geometry, else None."""
    tag = getattr(t, _TAG, None)
    if tag is None:
        return None
    ref, level, ptr, version = tag
    geom = ref()
    if geom is None or t.data_ptr() != ptr or t._version != version:
        return None
    return geom, level


def propagate_tag(src, dst):
    """Copy the geometry tag from ``src`` onto a bit-identical copy ``dst`` (used by ModelHook clones)."""
    tag = getattr(src, _TAG, None)
    if tag is not None and tag_of(src) is not None:
        ref, level, _, _ = tag
        setattr(dst, _TAG, (ref, level, dst.data_ptr(), dst._version))
    return dst


_SHARE_GRIDS = os.environ.get("PDFOPS_KNN_SHARE_GRIDS", "1") != "0"   # 0: one grid build per table (rounds 1-3; A/B runs)


class _Level:
    __slots__ = ("p", "o", "o_host", "n_max")

    def __init__(self, p, o, o_host):
        self.p, self.o, self.o_host = p, o, list(o_host)
        sizes = [self.o_host[0]] + [self.o_host[i] - self.o_host[i - 1] for i in range(1, len(self.o_host))]
        self.n_max = max(sizes) if sizes else 0


class Geometry:
    def __init__(self, coord, offset, offset_host=None):
        """coord (N,3) f32 contiguous; offset (B,) cumulative ends (any int dtype / device)."""
        if coord.dtype != torch.float32:
            coord = coord.float()
        # a fresh tensor OBJECT over the caller's storage: the geometry tag lives on the Python object, and the same
        # coordinate tensor may back several Geometry instances (e.g. pre-passes of later steps already in flight)
        coord = coord.contiguous().view(-1, 3)
        if offset_host is None:
            offset_host = [int(v) for v in offset.detach().cpu().tolist()]  # the ONE host sync per batch
        o_dev = offset.to(device=coord.device, dtype=torch.int32).contiguous()
        self.levels = []
        self._memo = {}
        self._grids = {}          # level -> (kNN grid workspace of the level's points, largest query count it serves); pre-pass only
        self._share_grids = False
        self._add_level(coord, o_dev, offset_host)

    # ------------------------------------------------------------------ levels
    def _add_level(self, p, o, o_host):
        lvl = len(self.levels)
        self.levels.append(_Level(p, o, o_host))
        setattr(p, _TAG, (weakref.ref(self), lvl, p.data_ptr(), p._version))
        return lvl

    def coord(self, level):
        return self.levels[level].p

    def offset(self, level):
        return self.levels[level].o

    def offset_host(self, level):
        return self.levels[level].o_host

    def sizes(self, level):
        """Per-scene point counts of a level as a device int64 tensor (memoised)."""
        key = ("sizes", level)
        if key not in self._memo:
            o = self.levels[level].o.long()
            self._memo[key] = torch.diff(o, prepend=o.new_zeros(1))
        return self._memo[key]

    @property
    def device(self):
        return self.levels[0].p.device

    # ------------------------------------------------------------------ memoised ops
    def down(self, level, stride):
        """FPS level -> level+1.  Returns (new_level, fps_idx int32 (M,)).  Sizes: floor(n_b/stride) per scene
        (point_transformer_seg.py:96-99)."""
        key = ("down", level, stride)
        if key in self._memo:
            return self._memo[key]
        if level != len(self.levels) - 1:
            raise RuntimeError("Geometry.down: levels must be created in order")
        L = self.levels[level]
        n_o, count, prev = [], 0, 0
        for e in L.o_host:
            count += (e - prev) // stride
            prev = e
            n_o.append(count)
        # device copy derived ON the device from the level's offsets (stream-ordered; no host buffer whose lifetime a
        # queued async H2D copy would depend on)
        n_o_dev = torch.cumsum(torch.div(self.sizes(level), stride, rounding_mode="floor"), 0).to(torch.int32)
        be = _native.backend_for(L.p)
        fps_idx = be.farthest_point_sampling(L.p, L.o, n_o_dev, L.n_max, count)
        n_p = L.p.index_select(0, fps_idx.long()).contiguous()
        new_level = self._add_level(n_p, n_o_dev, n_o)
        self._memo[key] = (new_level, fps_idx)
        return self._memo[key]

    def knn(self, nsample, src_level, query_level):
        """kNN of level ``query_level`` points over level ``src_level`` points -> (idx int32, dist2 f32)."""
        key = ("knn", nsample, src_level, query_level)
        if key not in self._memo:
            S, Q = self.levels[src_level], self.levels[query_level]
            be = _native.backend_for(S.p)
            grid = self._grid(be, src_level, Q.p.shape[0])
            self._memo[key] = be.knn_query(nsample, S.p, Q.p, S.o, Q.o, grid=grid) if grid is not None else be.knn_query(nsample, S.p, Q.p, S.o, Q.o)
            if S.p.is_cuda:   # the forward gathers visit the queries in Morton order (neighbouring queries share rows: L2 hits)
                _native.attach_order(self._memo[key][0], self.order(query_level), self.order(src_level))
        return self._memo[key]

    def _grid(self, be, level, m):
        """The kNN grid over a level's points, built once for all the tables that have the level as their SOURCE (self query, the down-sampling
        query of the next level, the interpolation query of the previous one: 14 builds per pre-pass became 5).  Held only while the
        pre-pass runs (``precompute`` drops the workspaces: ~8 MB per scene and level)."""
        if not hasattr(be, "knn_grid") or not getattr(self, "_share_grids", False):   # (views of a grouped pre-pass / static copies never build)
            return None
        ws = self._grids.get(level)
        if ws is None or ws[1] < m:
            L = self.levels[level]
            m_max = max(m, L.p.shape[0], self.levels[level - 1].p.shape[0] if level > 0 else 0)
            t = be.knn_grid(L.p, L.o, m_max)
            if t is None:
                return None
            ws = self._grids[level] = (t, m_max)
        return ws[0]

    def rel_moments(self, nsample, level):
        """Per-scene sums (scenes, 9) float64 of the relative coordinates of the self kNN table (nsample, level, level): the
        feature-independent half of the BatchNorm behind the layer's Linear(3, 3) (csrc/geom_moments.hip).  The batch's sums (over its
        scenes) are attached to the idx tensor, where the fused layer looks them up."""
        key = ("mom", nsample, level, level)
        if key not in self._memo:
            L = self.levels[level]
            idx, _ = self.knn(nsample, level, level)
            self._memo[key] = _native.backend_for(L.p).knn_rel_moments(nsample, L.p, L.o, idx)
            _native.attach_moments(idx, self._memo[key].sum(0))
        return self._memo[key]

    def order(self, level):
        """Morton order of a level's points, scene by scene: a permutation (N_l,) int32 of the level's rows.  Used as the VISITING order
        of the queries in the forward gathers (csrc/gather_ops.hip); nothing is stored in this order."""
        key = ("order", level)
        if key not in self._memo:
            L = self.levels[level]
            p = L.p
            # Quantisation grid PER SCENE (its own bounding box): the order of a scene's points is then a function of that scene alone, so a
            # batch gets the same visiting order -- and with it the same rounding of every per-workgroup partial sum -- whether its
            # pre-pass ran alone or as part of a group (round 3 quantised over the bounding box of everything in the call).
            be = _native.backend_for(p)
            if hasattr(be, "scene_morton_keys"):   # two launches (csrc/geom_moments.hip) instead of ~20 elementwise torch ops
                self._memo[key] = torch.argsort(be.scene_morton_keys(p, L.o), stable=True).to(torch.int32)
                return self._memo[key]
            b = len(L.o_host)
            scene = torch.repeat_interleave(torch.arange(b, device=p.device), self.sizes(level), output_size=p.shape[0])
            sidx = scene.unsqueeze(1).expand(-1, 3)
            lo = torch.full((b, 3), float("inf"), dtype=p.dtype, device=p.device).scatter_reduce_(0, sidx, p, "amin", include_self=True)
            hi = torch.full((b, 3), float("-inf"), dtype=p.dtype, device=p.device).scatter_reduce_(0, sidx, p, "amax", include_self=True)
            cell = torch.clamp((hi - lo).amax(dim=1) / 1023.0, min=1e-9)
            q = ((p - lo[scene]) / cell[scene].unsqueeze(1)).long().clamp_(0, 1023)

            def spread(v):   # 10 bits -> every third bit
                v = (v | (v << 16)) & 0x030000FF
                v = (v | (v << 8)) & 0x0300F00F
                v = (v | (v << 4)) & 0x030C30C3
                return (v | (v << 2)) & 0x09249249

            code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
            self._memo[key] = torch.argsort(code + (scene << 30), stable=True).to(torch.int32)
        return self._memo[key]

    def knn_dist(self, nsample, src_level, query_level):
        """(idx, sqrt(dist2)) as ``pointops.knn_query`` returns them (query.py:24); the square root is taken once per table, not
        once per call (18 calls per forward reuse 5 self tables)."""
        key = ("knn_dist", nsample, src_level, query_level)
        if key not in self._memo:
            idx, dist2 = self.knn(nsample, src_level, query_level)
            self._memo[key] = torch.sqrt(dist2)
        return self.knn(nsample, src_level, query_level)[0], self._memo[key]

    def interp(self, coarse_level, fine_level, k=3):
        """(idx (N_fine,k) int32, weight (N_fine,k) f32) of pointops.interpolation (interpolation.py:13-17)."""
        key = ("interp", k, coarse_level, fine_level)
        if key not in self._memo:
            idx, dist2 = self.knn(k, coarse_level, fine_level)
            self._memo[key] = (idx, interpolation_weights(dist2))
        return self._memo[key]

    def td(self, nsample, src_level, query_level):
        """Geometry-only tables of the fused TransitionDown (csrc/transition_down.hip) for the grouping of level ``src_level``
        points around the level ``query_level`` points: (rel4 (M,k,4) masked relative coordinates, Z (N,32) = per SOURCE point
        [sum of the relative coordinates of the rows that gather it (3) | number of such rows | 0...], per-scene moment sums
        (b,16), consts (16) = the batch's [sum rel^T rel (9) | sum rel (3) | 0...])."""
        key = ("td", nsample, src_level, query_level)
        if key not in self._memo:
            idx, _ = self.knn(nsample, src_level, query_level)
            S, Q = self.levels[src_level], self.levels[query_level]
            assert nsample == 16, "fused TransitionDown tables: nsample 16"
            be0 = _native.backend_for(S.p)
            # the inverse of the grouping table: Z in destination order here, the sparse input gradient of the backward likewise (no atomics)
            inv = self.inverse(nsample, src_level, query_level) if (S.p.is_cuda and be0.use_inverse) else None
            rel4, Z, _ = be0.td_tables(S.p, Q.p, idx, Q.o, inverse=inv) if inv is not None else be0.td_tables(S.p, Q.p, idx, Q.o)
            # per-scene [xx xy xz yy yz zz | sx sy sz] of the relative coordinates: one kernel (csrc/geom_moments.hip, fp64 sums; a python
            # loop over the scenes -- a slice + sum + store per scene and table -- was 15 ms of host time per group of 24 scenes)
            scene_sums = Z.new_zeros(len(Q.o_host), 16)
            be = _native.backend_for(S.p)
            if hasattr(be, "knn_rel_moments") and os.environ.get("PDFOPS_TD_MOM_KERNEL", "1") != "0":
                mom = be.knn_rel_moments(nsample, S.p, Q.o, idx, new_xyz=Q.p)          # (b, 9) = [S (3) | M (6)]
                if S.p.device not in _MOM_TO_TD:
                    _MOM_TO_TD[S.p.device] = torch.tensor([3, 4, 5, 6, 7, 8, 0, 1, 2], device=S.p.device)
                scene_sums[:, :9] = mom[:, _MOM_TO_TD[S.p.device]].float()
            else:
                r0, r1, r2 = rel4[..., 0], rel4[..., 1], rel4[..., 2]
                mom = torch.stack([(r0 * r0).sum(1), (r0 * r1).sum(1), (r0 * r2).sum(1), (r1 * r1).sum(1), (r1 * r2).sum(1), (r2 * r2).sum(1),
                                   r0.sum(1), r1.sum(1), r2.sum(1)], -1).double()
                prev = 0
                for si, e in enumerate(Q.o_host):
                    if e > prev:
                        scene_sums[si, :9] = mom[prev:e].sum(0).float()
                    prev = e
            self._memo[key] = (rel4, Z, scene_sums, _td_consts(scene_sums))
        return self._memo[key]

    def inverse(self, nsample, src_level, query_level):
        """Inverse of the kNN table (nsample, src_level, query_level): (inv_off, inv_entry, entry_base), see csrc/seg_gather.hip.
        Also cached on the idx tensor itself, which is where the backward passes look it up."""
        key = ("inv", nsample, src_level, query_level)
        if key not in self._memo:
            idx, _ = self.knn(nsample, src_level, query_level)
            self._memo[key] = _native.inverse_table(idx, self.levels[src_level].p.shape[0])
        return self._memo[key]

    def radius(self, radius, max_neighbor):
        """Radius table of the level-0 points with themselves: (N, max_neighbor) int32 GLOBAL rows, -1 padded, the point itself included
        (``pseudo_label.radius_neighbors(raw=True)``) -- the neighbour table of the PDF pseudo-label pass (pointpdf_v1m1_base.py:131-137).
        It reads coordinates only, so it belongs to the pre-pass like the kNN tables (0.66 ms of the step at 2 x 150k points otherwise)."""
        key = ("radius", float(radius), int(max_neighbor))
        if key not in self._memo:
            from .pseudo_label import radius_neighbors

            self._memo[key] = radius_neighbors(self.levels[0].p, self.levels[0].o, radius, max_neighbor, raw=True).int()
        return self._memo[key]

    def radius_cached(self, radius, max_neighbor):
        """The table ``radius()`` made ahead of the step, or None (the caller then runs the query itself)."""
        return self._memo.get(("radius", float(radius), int(max_neighbor)))

    def memo_size(self):
        """Number of memoised geometry ops (FPS + kNN + interpolation tables)."""
        return sum(1 for k in self._memo if k[0] in ("down", "knn", "interp"))  # ("td" tables are derived data)

    # ------------------------------------------------------------------ pre-pass
    def precompute(self, strides=(1, 4, 4, 4, 4), nsamples=(8, 16, 16, 16, 16), interp_k=3, recognizer=True, radius=None):
        """Run every coordinate-only op of one PointTransformer-Seg (+ PDF U-decoder) forward now, on the current
        stream: 4 FPS, 5 self-kNN, 4 down-sampling kNN, 4 (+1 for the U-decoder's level-5 self query) interpolation
        tables -- the 13 distinct kNN tables behind the reference's 31 calls."""
        lvl = 0
        self._share_grids = _SHARE_GRIDS
        if radius is not None:   # (radius, max_neighbor) of the pseudo-label pass: pseudo_label.make_pseudo_mask_fn(...).prepass_plan
            self.radius(*radius)
        self.knn_dist(nsamples[0], 0, 0)
        for i in range(1, len(strides)):
            new_level, _ = self.down(lvl, strides[i])
            self.knn(nsamples[i], lvl, new_level)        # TransitionDown grouping (point_transformer_seg.py:103-111)
            self.knn_dist(nsamples[i], new_level, new_level)  # PointTransformerLayer self query (:48-50), incl. the distances it returns
            self.interp(new_level, lvl, interp_k)        # TransitionUp fusion (:163-167)
            lvl = new_level
        if recognizer:
            self.interp(lvl, lvl, interp_k)              # PTRecognizer.dec5 interpolates level 5 onto itself (pt_v1.py:37)
        if self.levels[0].p.is_cuda:                     # tables of the fused TransitionDown (device path only)
            for i in range(1, len(strides)):
                self.td(nsamples[i], i - 1, i)
            if _native.hip_backend().use_moments:        # BatchNorm statistics of the layers' geometry branch from coordinate sums
                for i in range(len(strides)):
                    self.rel_moments(nsamples[i], i)
            if _native.hip_backend().use_inverse:        # inverse tables for the gather-form backward passes (device path only)
                for i in range(len(strides)):
                    self.inverse(nsamples[i], i, i)      # PointTransformerLayer g_xk / g_xv
                for i in range(1, len(strides)):
                    self.inverse(interp_k, i, i - 1)     # interpolation backward (TransitionUp, U-decoder)
                if recognizer:
                    self.inverse(interp_k, lvl, lvl)
        self._grids.clear()
        self._share_grids = False
        return self

    # ------------------------------------------------------------------ fixed-address form (hipGraph replay of a captured step)
    def flat_tensors(self):
        """Every device tensor a forward / backward over this Geometry reads, as an ordered list of (slot, tensor): level coordinates and
        offsets, the memoised tables, the batch's coordinate sums (an attachment of the self-kNN index tensors) -- with the inverse
        tables normalised to (offsets from 0, the batch's own entry list, entry base 0), so that a batch cut out of a grouped pre-pass
        (``split``: views into the group's arrays + an integer base) and a batch with its own pre-pass give the same list of shapes.
        The order is a function of the plan only (sorted keys)."""
        out = []
        for l, lv in enumerate(self.levels):
            out += [(("p", l), lv.p), (("o", l), lv.o)]
        for key in sorted(self._memo, key=repr):
            v = self._memo[key]
            kind = key[0]
            if kind == "inv":
                off, ent, base = v
                nent = self._memo[("knn",) + key[1:]][0].numel()
                off0 = off[:1]
                pos = (_arange(nent, off.device) + off0).clamp_(max=max(ent.numel() - 1, 0)).long()
                out += [((key, 0), off - off0), ((key, 1), ent.index_select(0, pos) - base if base else ent.index_select(0, pos))]
            elif kind == "interp":     # (the index tensor is the kNN entry's own object)
                out.append(((key, 1), v[1]))
            else:
                for j, t in enumerate(v if isinstance(v, tuple) else (v,)):
                    if isinstance(t, torch.Tensor):
                        out.append(((key, j), t))
            if kind == "mom":
                m = _native.moments_of(self._memo[("knn",) + key[1:]][0])
                out.append(((("mom_batch",) + key[1:]), m if m is not None else v.sum(0)))
        return out

    def flat_sources(self):
        """``flat_tensors()`` without materialising anything: (slot, source tensor, src_offset tensor | None, sub tensor | None, sub_const)
        per slot, for the one-launch staging copy (csrc/stage_copy.hip).  The inverse tables' fix-ups happen inside that copy: the offset
        array minus its own first element; the entry list as the window of the (group's) array that starts at that first offset, minus
        the batch's entry base."""
        out = []
        for l, lv in enumerate(self.levels):
            out += [(("p", l), lv.p, None, None, 0), (("o", l), lv.o, None, None, 0)]
        for key in sorted(self._memo, key=repr):
            v = self._memo[key]
            kind = key[0]
            if kind == "inv":
                off, ent, base = v
                out += [((key, 0), off, None, off, 0), ((key, 1), ent, off, None, int(base))]
            elif kind == "interp":
                out.append(((key, 1), v[1], None, None, 0))
            else:
                for j, t in enumerate(v if isinstance(v, tuple) else (v,)):
                    if isinstance(t, torch.Tensor):
                        out.append(((key, j), t, None, None, 0))
            if kind == "mom":
                m = _native.moments_of(self._memo[("knn",) + key[1:]][0])
                out.append(((("mom_batch",) + key[1:]), m if m is not None else v.sum(0), None, None, 0))
        return out

    def pack(self, layout):
        """This Geometry's tensors as ONE flat byte buffer in ``layout``'s order (``StaticGeometry.layout``): ~70 small copies, issued on
        the current stream -- the pre-pass side stream, so the training stream needs a single copy per step."""
        items = self.flat_tensors()
        if len(items) != len(layout.items):
            raise ValueError("Geometry.pack: this geometry does not have the layout's tables")
        flat = torch.empty((layout.nbytes,), dtype=torch.uint8, device=self.device)
        for (slot, t), (lslot, shape, dtype, o, nb) in zip(items, layout.items):
            if slot != lslot or tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError(f"Geometry.pack: {slot} {tuple(t.shape)} {t.dtype} does not match the captured layout {lslot} {shape} {dtype} "
                                 "(hipGraph replay needs identical scene sizes)")
            flat[o:o + nb].view(dtype).view(shape).copy_(t)
        return flat

    def split(self, scene_counts):
        """Per-batch Geometry objects of a pre-pass that was run over several batches at once (scenes of batch 0, then
        batch 1, ...; ``scene_counts[b]`` scenes each).  Every op on this path works scene by scene, so the tables of a
        batch are row slices of the group's tables with the index base of the batch subtracted -- bit-identical to a
        pre-pass of the batch alone.  This is how FPS latency (a chain of ~25k dependent arg-max steps per scene, one
        workgroup per scene) is amortised: one launch carries the scenes of ``len(scene_counts)`` upcoming steps."""
        assert sum(scene_counts) == len(self.levels[0].o_host), "scene_counts must cover the group's scenes"
        # Index tables are rebased for the WHOLE group first (one subtraction per table against a per-row vector of batch bases): the
        # per-batch entries below are then plain row slices.  Rebasing per batch and table was 13 x D small kernels and ~4 ms of host time
        # per group of 12 batches.
        bounds, s0 = [], 0          # per batch: [(first row, end row) per level]
        for nsc in scene_counts:
            s1 = s0 + nsc
            bounds.append([((lv.o_host[s0 - 1] if s0 > 0 else 0), lv.o_host[s1 - 1]) for lv in self.levels])
            s0 = s1
        dev = self.levels[0].p.device
        base_rows = {}              # (source level, query level) -> per query row: first source row of its batch

        def base_vector(src_level, query_level):
            key = (src_level, query_level)
            if key not in base_rows:
                both = torch.tensor([[b[src_level][0] for b in bounds], [b[query_level][1] - b[query_level][0] for b in bounds]],
                                    dtype=torch.int64).to(dev, non_blocking=True)   # (one small copy; expanded on the device)
                base_rows[key] = torch.repeat_interleave(both[0].to(torch.int32), both[1], output_size=bounds[-1][query_level][1])
            return base_rows[key]

        # per-batch sums of the per-scene tables (TransitionDown constants, coordinate sums of the self tables): one segmented sum per
        # table for all batches (index_add over the scenes' batch ids), not a reduction per batch and table
        batch_sums = {}
        if len(scene_counts) > 1:
            batch_of_scene = torch.repeat_interleave(torch.arange(len(scene_counts)), torch.tensor(scene_counts)).to(dev, non_blocking=True)
            for key, val in self._memo.items():
                if key[0] == "td":
                    per = torch.zeros((len(scene_counts), 16), dtype=torch.float64, device=dev).index_add_(0, batch_of_scene, val[2].double())
                    if dev not in _TD_PERM:
                        _TD_PERM[dev] = torch.tensor([0, 1, 2, 1, 3, 4, 2, 4, 5, 6, 7, 8], device=dev)
                    consts = val[2].new_zeros((len(scene_counts), 16))
                    consts[:, :12] = per[:, _TD_PERM[dev]].float()
                    batch_sums[key] = consts
                elif key[0] == "mom":
                    batch_sums[key] = torch.zeros((len(scene_counts), 9), dtype=torch.float64, device=dev).index_add_(0, batch_of_scene, val)
        rebased = {}
        if len(scene_counts) > 1:
            for key, val in self._memo.items():
                if key[0] == "down":
                    new_level, fps_idx = val
                    rebased[key] = fps_idx - base_vector(key[1], new_level)
                elif key[0] == "knn":
                    idx = val[0]
                    rebased[key] = torch.where(idx >= 0, idx - base_vector(key[2], key[3])[:, None], idx)
                elif key[0] == "radius":
                    rebased[key] = torch.where(val >= 0, val - base_vector(0, 0)[:, None], val)
        out, s0 = [], 0
        for bi, nsc in enumerate(scene_counts):
            s1 = s0 + nsc
            g = Geometry.__new__(Geometry)
            g.levels, g._memo = [], {}
            rows = []  # (first row, end row) of this batch at every level
            for lv in self.levels:
                r0 = lv.o_host[s0 - 1] if s0 > 0 else 0
                r1 = lv.o_host[s1 - 1]
                rows.append((r0, r1))
                o = lv.o[s0:s1]
                g._add_level(lv.p[r0:r1], (o - r0) if r0 else o, [e - r0 for e in lv.o_host[s0:s1]])

            def rebase(idx, base):
                return idx if base == 0 else torch.where(idx >= 0, idx - base, idx)

            for key, val in self._memo.items():
                kind = key[0]
                if kind == "sizes":
                    g._memo[key] = val[s0:s1]
                elif kind == "down":
                    new_level, fps_idx = val
                    q0, q1 = rows[new_level]
                    g._memo[key] = (new_level, rebased[key][q0:q1] if key in rebased else rebase(fps_idx[q0:q1], rows[key[1]][0]))
                elif kind == "knn":
                    (idx, dist2), (q0, q1) = val, rows[key[3]]
                    g._memo[key] = (rebased[key][q0:q1] if key in rebased else rebase(idx[q0:q1], rows[key[2]][0]), dist2[q0:q1])
                elif kind == "order":   # scene-major: the batch's rows occupy the same positions of the sorted list
                    r0, r1 = rows[key[1]]
                    g._memo[key] = val[r0:r1] - r0 if r0 else val[r0:r1]
                elif kind == "knn_dist":
                    q0, q1 = rows[key[3]]
                    g._memo[key] = val[q0:q1]
                elif kind == "radius":   # global rows of the group -> rows of the batch
                    r0, r1 = rows[0]
                    g._memo[key] = rebased[key][r0:r1] if key in rebased else rebase(val[r0:r1], r0)
                elif kind == "interp":   # (the index tensor OBJECT of the kNN entry: the inverse table is cached on it)
                    (idx, weight), (q0, q1) = val, rows[key[3]]
                    g._memo[key] = (g._memo[("knn",) + key[1:]][0], weight[q0:q1])
                elif kind == "td":   # rel4 by query rows, Z by source rows; the 12 sums are per batch
                    (rel4, Z, scene_sums, _), (q0, q1), (r0, r1) = val, rows[key[3]], rows[key[2]]
                    g._memo[key] = (rel4[q0:q1], Z[r0:r1], scene_sums[s0:s1], batch_sums[key][bi] if key in batch_sums else _td_consts(scene_sums[s0:s1]))
                elif kind == "mom":   # per-scene sums: the batch's scenes
                    g._memo[key] = val[s0:s1]
                elif kind == "inv":   # absolute positions into the group's shared entry array; entry ids rebased by the batch's first entry
                    (off, ent, base), (q0, _), (r0, r1) = val, rows[key[3]], rows[key[2]]
                    tab = (off[r0:r1 + 1], ent, base + q0 * key[1])
                    g._memo[key] = tab
                    _native.attach_inverse(g._memo[("knn",) + key[1:]][0], r1 - r0, tab)
                else:
                    raise RuntimeError(f"Geometry.split: unknown memo entry {key}")
            for key, val in g._memo.items():   # visiting orders and coordinate sums travel with the batch's index tensors
                if key[0] == "knn" and ("order", key[3]) in g._memo:
                    _native.attach_order(val[0], g._memo[("order", key[3])], g._memo.get(("order", key[2])))
                if key[0] == "mom":
                    _native.attach_moments(g._memo[("knn",) + key[1:]][0], batch_sums[key][bi] if key in batch_sums else val.sum(0))
            out.append(g)
            s0 = s1
        return out

    def tensors(self):
        out = [lv.p for lv in self.levels] + [lv.o for lv in self.levels]
        for v in self._memo.values():
            out += [t for t in (v if isinstance(v, tuple) else (v,)) if isinstance(t, torch.Tensor)]
        # tensors that live ONLY as attachments of an index tensor (the batch's coordinate sums: `val.sum(0)` is a fresh allocation of
        # the pre-pass stream; orders / inverse tables are normally memo entries too, listed again harmlessly): the consumer stream
        # must record them as well, or the pre-pass stream's allocator may hand the block out while main-stream kernels still read it
        for t in list(out):
            for tag in (_native._MOM, _native._ORD, _native._INV):
                att = getattr(t, tag, None)
                if att is None:
                    continue
                for a in att[2:]:
                    for u in (a if isinstance(a, tuple) else (a,)):
                        if isinstance(u, torch.Tensor):
                            out.append(u)
        return out


def _td_consts(scene_sums):
    """(b,16) per-scene [xx xy xz yy yz zz | sx sy sz] -> the batch's [sum rel^T rel (9) | sum rel (3) | 0 x4]."""
    t = scene_sums.double().sum(0)
    # one gather through a per-device index tensor made ONCE (an H2D copy per call would block the host behind the whole pre-pass stream;
    # torch.stack of nine 0-dim views, the previous form, cost 250 us of host time per call, 14 ms per group of 12 batches)
    key = scene_sums.device
    if key not in _TD_PERM:
        _TD_PERM[key] = torch.tensor([0, 1, 2, 1, 3, 4, 2, 4, 5, 6, 7, 8], device=key)
    out = scene_sums.new_zeros(16)
    out[:12] = t[_TD_PERM[key]].float()
    return out


_TD_PERM = {}
_MOM_TO_TD = {}


_ARANGE = {}


def _arange(n, device):
    key = (int(n), device)
    if key not in _ARANGE:
        _ARANGE[key] = torch.arange(int(n), device=device, dtype=torch.int32)
    return _ARANGE[key]


class GeometryLayout:
    """Slots, shapes, dtypes and byte offsets (256-byte aligned) of ``Geometry.flat_tensors()`` inside one flat buffer."""

    def __init__(self, geom):
        self.items, o = [], 0
        for slot, t in geom.flat_tensors():
            nb = t.numel() * t.element_size()
            self.items.append((slot, tuple(t.shape), t.dtype, o, nb))
            o += (nb + 255) & ~255
        self.nbytes = o
        self.levels = [list(lv.o_host) for lv in geom.levels]
        self.memo = {key: (v if not isinstance(v, tuple) else tuple(None if isinstance(t, torch.Tensor) else t for t in v))
                     for key, v in geom._memo.items()}   # non-tensor members (level numbers) of the memo entries


class StaticGeometry(Geometry):
    """A Geometry whose every tensor is a view of ONE flat device buffer at fixed addresses: a training step captured into a hipGraph
    against it (engine.CapturedStep) replays on any batch with the same scene sizes after ``load(packed)`` -- one device copy of the
    buffer another Geometry's ``pack(layout)`` produced."""

    def __init__(self, template):
        self.layout = GeometryLayout(template)
        self.flat = template.pack(self.layout)
        views = {slot: self.flat[o:o + nb].view(dtype).view(shape) for slot, shape, dtype, o, nb in self.layout.items}
        self.levels, self._memo = [], {}
        for l, o_host in enumerate(self.layout.levels):
            self._add_level(views[("p", l)], views[("o", l)], o_host)
        for key, proto in self.layout.memo.items():
            kind = key[0]
            if kind == "inv":
                self._memo[key] = (views[(key, 0)], views[(key, 1)], 0)
            elif kind == "interp":
                continue
            elif isinstance(proto, tuple):
                self._memo[key] = tuple(views[(key, j)] if t is None else t for j, t in enumerate(proto))
            else:
                self._memo[key] = views[(key, 0)]
        for key in self.layout.memo:
            if key[0] == "interp":
                self._memo[key] = (self._memo[("knn",) + key[1:]][0], views[(key, 1)])
        for key, val in self._memo.items():   # the attachments the backend looks up on the index tensors
            if key[0] == "knn" and ("order", key[3]) in self._memo:
                _native.attach_order(val[0], self._memo[("order", key[3])], self._memo.get(("order", key[2])))
            if key[0] == "mom":
                _native.attach_moments(self._memo[("knn",) + key[1:]][0], views[("mom_batch",) + key[1:]])
            if key[0] == "inv":
                _native.attach_inverse(self._memo[("knn",) + key[1:]][0], self.levels[key[2]].p.shape[0], val)

    def load(self, packed):
        """Overwrite the tables with another batch's (``Geometry.pack(self.layout)``): one copy on the current stream."""
        self.flat.data.copy_(packed)   # (.data: the views' version counters -- which the geometry tags and attachments check -- stay put)
        return self

    def stage(self, geom, extra=()):
        """Overwrite the tables with those of ``geom`` (same scene sizes; any Geometry, e.g. a batch cut out of a grouped pre-pass) in ONE
        launch on the current stream (csrc/stage_copy.hip), reading ``geom``'s tensors where they lie.  ``extra``: further
        (source, destination) tensor pairs to move in the same launch (the batch's own tensors).  Nothing is allocated; the version
        counters of the static views do not move."""
        import ctypes

        be = _native.hip_backend()
        src = geom.flat_sources()
        if len(src) != len(self.layout.items):
            raise ValueError("StaticGeometry.stage: the geometry does not have the captured layout's tables")
        n = len(src) + len(extra)
        segs = (be.CopySeg * n)()
        base = self.flat.data_ptr()
        for i, ((slot, t, soff, sub, subc), (lslot, shape, dtype, o, nb)) in enumerate(zip(src, self.layout.items)):
            windowed = soff is not None
            if slot != lslot or t.dtype != dtype or not t.is_contiguous() or (not windowed and tuple(t.shape) != shape):
                raise ValueError(f"StaticGeometry.stage: {slot} {tuple(t.shape)} {t.dtype} does not match the captured layout {lslot} {shape} "
                                 f"{dtype} (hipGraph replay needs identical scene sizes)")
            sg = segs[i]
            sg.src, sg.dst, sg.nbytes = t.data_ptr(), base + o, nb
            sg.src_offset = soff.data_ptr() if windowed else None
            sg.sub = sub.data_ptr() if sub is not None else None
            sg.sub_const = subc
            sg.src_elems = t.numel() if windowed else 0
        for i, (a, b) in enumerate(extra):
            if a.shape != b.shape or a.dtype != b.dtype or not (a.is_contiguous() and b.is_contiguous()) or (a.numel() * a.element_size()) % 4:
                raise ValueError("StaticGeometry.stage: extra pairs must be contiguous tensors of one shape and dtype (4-byte multiples)")
            sg = segs[len(src) + i]
            sg.src, sg.dst, sg.nbytes = a.data_ptr(), b.data_ptr(), a.numel() * a.element_size()
            sg.src_offset, sg.sub, sg.sub_const, sg.src_elems = None, None, 0, 0
        _native.require_current_device(self.flat)
        rc = be.lib.pdf_stage_copy(n, segs, ctypes.c_void_p(_native.raw_stream()))
        if rc != 0:
            raise _native.PdfOpsError(f"pdf_stage_copy failed with status {rc}")
        return self


_WAIT_ALWAYS = os.environ.get("PDFOPS_WAIT_QUERY") == "0"


class _LazyTicket:
    """Ticket j of a group whose pre-pass is being built on the prefetcher's worker thread."""

    def __init__(self, future, j):
        self.future, self.j = future, j

    def resolve(self):
        return self.future.result()[self.j]


class GeometryPrefetcher:
    """Computes the Geometry of upcoming batches on side HIP streams while the current batch trains.

    FPS is a chain of tens of thousands of dependent arg-max steps that occupies one workgroup per scene; the rest of
    the chip would idle behind it.  Coordinates are known as soon as a batch is collated, so its pre-pass can run
    ``depth`` steps ahead (the way a DataLoader prefetches): every step still pays for exactly one full pre-pass,
    only its latency is taken off the critical path.  ``get()`` makes the consumer stream wait for the pre-pass and
    registers the tables with it (caching-allocator stream safety)."""

    def __init__(self, depth=3, threaded=False, **plan):
        """``threaded``: build the pre-pass on a worker thread (its ~200 launches and tensor ops per group are ~20 ms of Python / dispatch
        time: on the training thread they are 1.7 ms per step of a host side that is as long as the device side; the DataLoader-worker
        pattern).  Tickets then carry a future; ``get`` joins it."""
        self.depth, self.plan = depth, plan
        self.streams = [torch.cuda.Stream() for _ in range(max(depth, 1))]   # (stream priorities: measured, no effect on this stack)
        self._n = 0
        self.pool = None
        self.packer = None   # callable Geometry -> flat tensor, run on the pre-pass stream (set to engine.CapturedStep.pack)
        if threaded:
            from concurrent.futures import ThreadPoolExecutor

            self.pool = ThreadPoolExecutor(max_workers=1)
            self.device = torch.cuda.current_device()

    def submit(self, coord, offset, offset_host=None):
        stream = self.streams[self._n % len(self.streams)]
        self._n += 1
        stream.wait_stream(torch.cuda.current_stream())  # inputs were produced on the caller's stream
        with torch.cuda.stream(stream):
            geom = Geometry(coord, offset, offset_host).precompute(**self.plan)
            if self.packer is not None:
                geom.packed = self.packer(geom)
            done = torch.cuda.Event()
            done.record(stream)
        return geom, done, stream

    def submit_group(self, batches, ready=None):
        """One pre-pass over the scenes of several upcoming batches (dicts with coord / offset / offset_host) -> one
        ticket per batch.  See Geometry.split.  ``ready``: an event recorded on the caller's stream once the batches' tensors were
        complete (default: the caller's stream as it stands NOW -- a caller that submits after queueing a training step passes the event
        it recorded before that step, or the pre-pass would wait for the step)."""
        stream = self.streams[self._n % len(self.streams)]
        self._n += 1
        if self.pool is not None:
            if ready is None:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream())   # the batches' tensors were produced on the caller's stream

            def work():
                torch.cuda.set_device(self.device)
                stream.wait_event(ready)
                return self._group_on(stream, batches)

            fut = self.pool.submit(work)
            return [_LazyTicket(fut, j) for j in range(len(batches))]
        if ready is not None:
            stream.wait_event(ready)
        else:
            stream.wait_stream(torch.cuda.current_stream())
        return self._group_on(stream, batches)

    def _group_on(self, stream, batches):
        with torch.cuda.stream(stream):
            for b in batches:   # (allocated on the caller's / the loader's copy stream, read here)
                for k in ("coord", "offset"):
                    if torch.is_tensor(b[k]) and b[k].is_cuda:
                        b[k].record_stream(stream)
            coord = torch.cat([b["coord"] for b in batches])
            counts, o_host, base = [], [], 0
            for b in batches:
                counts.append(len(b["offset_host"]))
                o_host += [base + int(e) for e in b["offset_host"]]
                base = o_host[-1]
            bases = [0] + [o_host[sum(counts[:i + 1]) - 1] for i in range(len(batches) - 1)]
            offset = torch.cat([b["offset"].to(torch.int32) + int(bs) for b, bs in zip(batches, bases)])
            group = Geometry(coord, offset, o_host).precompute(**self.plan)
            geoms = group.split(counts)
            if self.packer is not None:   # fixed-address replay (engine.CapturedStep): the batch's tables as one flat buffer, made here
                for g in geoms:
                    g.packed = self.packer(g)
            done = torch.cuda.Event()
            done.record(stream)
        return [(g, done, stream) for g in geoms]

    @staticmethod
    def get(ticket):
        if isinstance(ticket, _LazyTicket):
            ticket = ticket.resolve()
        geom, done, stream = ticket
        cur = torch.cuda.current_stream()
        # The pre-pass runs a group ahead, so its event has usually fired long ago: asked on the HOST first (round 6).  A device-side wait
        # is a barrier packet in the training queue, and with the pre-pass queues busy beside it that packet cost ~0.45 ms of idle time at
        # the start of every step (profiles/r06_z_timeline.txt: the gap in front of the staging launch; PDFOPS_WAIT_QUERY=0: always wait).
        if _WAIT_ALWAYS or not done.query():
            cur.wait_event(done)
        for t in geom.tensors():
            t.record_stream(cur)
        packed = getattr(geom, "packed", None)
        if packed is not None:
            packed.record_stream(cur)
        return geom


def interpolation_weights(dist2):
    """weight = (1/(sqrt(d2)+1e-8)) / sum  -- libs/pointops/functions/interpolation.py:14-17."""
    be = _native.backend_for(dist2)
    if hasattr(be, "interpolation_weights"):
        return be.interpolation_weights(dist2)
    dist_recip = 1.0 / (torch.sqrt(dist2) + 1e-8)
    return dist_recip / torch.sum(dist_recip, dim=1, keepdim=True)
