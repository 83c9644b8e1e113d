"""CPU suite: host logic of the training-loop pieces in pointcloudpdf_amd/engine.py (no GPU, no kernels): the grouped look-ahead
schedule of GroupedGeometryLoader against a recording stand-in for the pre-pass, the checkpoint layout of FusedSGD against
torch.optim.SGD, the frozen-state guard of CapturedStep."""
import copy
import types

import pytest
import torch


class RecordingPrefetcher:
    """Stands in for geometry.GeometryPrefetcher: records what was submitted when; a ticket resolves to (group number, position)."""

    def __init__(self, log):
        self.log, self.groups = log, 0

    def submit_group(self, batches, ready=None):
        g = self.groups
        self.groups += 1
        self.log.append(("submit", g, [b["id"] for b in batches]))
        return [(g, j, b["id"]) for j, b in enumerate(batches)]

    def get(self, ticket):
        self.log.append(("get", ticket[2]))
        return ticket


def _batches(n, scenes=2):
    for i in range(n):
        yield dict(id=i, coord=torch.zeros(4, 3), offset=torch.tensor([2, 4][:scenes], dtype=torch.int32), offset_host=[2, 4][:scenes])


@pytest.mark.parametrize("n,group,first,delay", [(30, 6, None, 2), (25, 10, 5, 2), (7, 12, None, 2), (9, 3, 1, 0), (12, 4, None, 5), (0, 4, None, 2)])
def test_grouped_loader_schedule(n, group, first, delay):
    """Every batch comes out once, in order, with ITS OWN tables; a group's pre-pass is submitted before its first batch is asked for,
    one group ahead at most, and only after `submit_delay` steps of the running group were handed out (the consumer enqueues a step
    between two `next` calls, so the submission's host work never sits in front of those steps)."""
    from pointcloudpdf_amd import engine

    log = []
    loader = engine.GroupedGeometryLoader(_batches(n), group=group, first_group=first, submit_delay=delay, prefetcher=RecordingPrefetcher(log))
    seen = []
    for b in loader:
        log.append(("yield", b["id"]))
        g, j, bid = b["pdf_geometry"]
        assert bid == b["id"]
        seen.append(b["id"])
    assert seen == list(range(n))
    submits = [(k, e) for k, e in enumerate(log) if e[0] == "submit"]
    covered = [i for _, e in submits for i in e[2]]
    assert covered == list(range(n))                                   # exactly one pre-pass per batch
    sizes = [len(e[2]) for _, e in submits]
    if n:
        assert sizes[0] == min(first or group, n) and all(s <= group for s in sizes[1:])
    for gi, (pos, e) in enumerate(submits):
        yielded_before = [x[1] for x in log[:pos] if x[0] == "yield"]
        assert all(i not in yielded_before for i in e[2])               # submitted ahead of use
        if gi >= 1:
            prev = submits[gi - 1][1][2]
            want = min(delay, len(prev) - 1)                            # handed out `delay` batches of the running group first
            assert yielded_before == list(range(prev[0] + want)), (gi, yielded_before, prev, want)
        if gi >= 2:                                                     # one group ahead, never two
            assert submits[gi - 2][1][2][-1] in yielded_before


def test_grouped_loader_respects_the_scene_budget_and_the_serial_mode():
    from pointcloudpdf_amd import engine

    log = []
    cap = engine.GroupedGeometryLoader.MAX_SCENES
    loader = engine.GroupedGeometryLoader(_batches(80), group=50, prefetcher=RecordingPrefetcher(log))
    assert [b["id"] for b in loader] == list(range(80))
    assert all(2 * len(e[2]) <= cap for e in log if e[0] == "submit")    # 2 scenes per batch
    serial = engine.GroupedGeometryLoader(_batches(5), group=0)
    out = list(serial)
    assert [b["id"] for b in out] == list(range(5)) and all("pdf_geometry" not in b for b in out)


def test_grouped_loader_adds_the_host_copy_of_the_offsets():
    """collate_fn hands over `offset` only (datasets/utils.py:34-39); the pre-pass wants the scene ends on the host as well."""
    from pointcloudpdf_amd import engine

    src = [dict(id=0, coord=torch.zeros(5, 3), offset=torch.tensor([2, 5]))]
    out = list(engine.GroupedGeometryLoader(src, group=0))
    assert out[0]["offset_host"] == [2, 5]


class _Lib:
    def pdf_sgd_chunk(self):
        return 4096


def test_fused_sgd_state_dict_round_trip():
    """engine.FusedSGD's checkpoint in torch.optim.SGD and back (the param groups lacked torch.optim.SGD's other keys and
    `step()` of the torch optimizer raised KeyError 'dampening')."""
    from pointcloudpdf_amd import engine

    g = torch.Generator().manual_seed(0)
    shapes = [(5,), (3, 4), (17,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fused = engine.FusedSGD(pa, lr=0.05, momentum=0.9, weight_decay=1e-2, backend=types.SimpleNamespace(lib=_Lib()))
    for p in pa:
        fused.state[p]["momentum_buffer"] = torch.randn(p.shape, generator=g)
    ref = torch.optim.SGD(pb, lr=1.0, momentum=0.1)
    ref.load_state_dict(copy.deepcopy(fused.state_dict()))
    assert ref.param_groups[0]["lr"] == 0.05 and ref.param_groups[0]["momentum"] == 0.9 and ref.param_groups[0]["weight_decay"] == 1e-2
    for p in pb:
        p.grad = torch.ones_like(p)
    ref.step()                                                         # (KeyError 'dampening' before the defaults were added)
    for x, y in zip(pa, pb):
        buf = fused.state[x]["momentum_buffer"]
        want = x.detach() - 0.05 * (0.9 * buf + (1.0 + 1e-2 * x.detach()))
        assert torch.allclose(y.detach(), want, rtol=1e-6, atol=1e-7)
    # and back: torch.optim.SGD's checkpoint in FusedSGD
    fused2 = engine.FusedSGD([torch.nn.Parameter(p.detach().clone()) for p in pb], lr=9.0, momentum=0.0, backend=types.SimpleNamespace(lib=_Lib()))
    fused2.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert fused2.param_groups[0]["lr"] == 0.05 and fused2.param_groups[0]["dampening"] == 0
    for q, y in zip(fused2.param_groups[0]["params"], pb):
        assert torch.equal(fused2.state[q]["momentum_buffer"], ref.state[y]["momentum_buffer"])
    # a larger group added later re-sizes the pointer tables
    rows = fused2._rows
    fused2.add_param_group(dict(params=[torch.nn.Parameter(torch.zeros(2)) for _ in range(rows + 3)]))
    assert fused2._rows == rows + 3 and all(t.shape[0] == rows + 3 for t in fused2._tabs)
    # unsupported torch.optim.SGD options are refused, not ignored
    fused2.param_groups[0]["nesterov"] = True
    fused2.param_groups[0]["params"][0].grad = torch.zeros_like(fused2.param_groups[0]["params"][0])
    with pytest.raises(RuntimeError, match="nesterov"):
        fused2.step()


def test_captured_step_state_guard_is_host_logic():
    """CapturedStep bakes alpha / the epoch gate / train mode into the graph: `matches` and `__call__` compare the live values with the
    captured ones (checked here on the comparison itself; the replay is a GPU test)."""
    from pointcloudpdf_amd import engine

    step = engine.OpenSegStep(backbone="PointTransformer-Seg26")
    step.train()
    cap = engine.CapturedStep.__new__(engine.CapturedStep)
    cap.step, cap.autocast = step, None
    a = cap._python_state()
    step.recognizer.alpha = float(step.recognizer.alpha) * 0.5
    b = cap._python_state()
    step.eval()
    c = cap._python_state()
    assert a != b and b != c and a[0] == 2 * b[0]


def test_train_step_keeps_one_graph_per_size_class_host_logic(monkeypatch):
    """TrainStep._capture_for (which capture IS this batch's step): one per scene-size signature up to ``max_captures``, every capture on
    the first one's stream, further signatures eager, stale captures (schedule state moved) released and their size class captured
    again.  The captures are stand-ins here (the replays themselves: tests/test_gpu_model.py)."""
    from pointcloudpdf_amd import engine

    made = []
    state = {"alpha": 1.0}

    class FakeCapture:
        def __init__(self, step, batch, geom=None, autocast=None, loss_scale=1.0, stream=None):
            self.sizes, self.stream_arg = list(batch["offset_host"]), stream
            self.stream = stream if stream is not None else object()
            self.frozen = state["alpha"]
            made.append(self)

        def _python_state(self):
            return state["alpha"]

        def matches(self, batch):
            return list(batch["offset_host"]) == self.sizes and self._python_state() == self.frozen

    monkeypatch.setattr(engine, "CapturedStep", FakeCapture)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda *a, **k: None)
    lin = torch.nn.Linear(2, 2)
    train = engine.TrainStep(lin, optimizer=None, graph=True, max_captures=2)
    A, B, C = ({"offset_host": s} for s in ([10, 20], [11, 20], [12, 20]))
    a = train._capture_for(A, None)
    assert a is made[0] and train.captured is a and a.stream_arg is None
    assert train._capture_for(A, None) is a and len(made) == 1          # same sizes: the same capture
    b = train._capture_for(B, None)
    assert b is made[1] and b.stream_arg is a.stream                     # recorded on the first capture's stream
    assert train._capture_for(C, None) is None and len(made) == 2       # both slots taken: eager
    assert train._capture_for(A, None) is a and train._capture_for(B, None) is b
    state["alpha"] = 0.1                                                 # the schedule moved: both graphs are stale
    c = train._capture_for(C, None)
    assert c is made[2] and train.captures == [c] and train.captured is c and c.stream_arg is a.stream
    a2 = train._capture_for(A, None)
    assert a2 is made[3] and train.captures == [c, a2]
    assert train._capture_for(B, None) is None

    def boom(*a, **k):
        raise RuntimeError("capture failed")
    monkeypatch.setattr(engine, "CapturedStep", boom)
    monkeypatch.setattr(engine, "release_autograd_state", lambda step: None)
    one = engine.TrainStep(lin, optimizer=None, graph=True)
    assert one._capture_for(A, None) is None and "capture failed" in one.capture_error and one.graph is False and one.captures == []
