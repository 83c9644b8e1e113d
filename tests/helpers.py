"""Shared helpers for the parity tests."""
import numpy as np
import torch

from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.model_hook import BaseModelHook
from pointcloudpdf_amd.point_transformer import PointTransformerSeg50
from pointcloudpdf_amd.recognizer import PTRecognizer

# must match tests/golden/make_golden.py
MODEL_CASES = {"b2_2048_1600": ([2048, 1600], 0.25), "b1_3000": ([3000], 0.2), "b1_8192": ([8192], 0.12)}
ROW_STRIDE = 4
GRAD_ROWS = 16
HOOK_CONFIG = {
    **{f"backbone.enc{i}": ["forward_output"] for i in range(1, 6)},
    **{f"backbone.dec{i}.1": ["forward_output"] for i in range(1, 6)},
    "backbone": ["forward_output"],
}
REL_TOL = 1e-4  # north_star: float features / logits within 1e-4 rel
# Parameter gradients: the reference's OWN fp32 gradients deviate 2e-3 .. 8e-3 (max-norm relative) from an fp64
# evaluation of the same network (train-mode BatchNorm backward cancels heavily; measured in DESIGN.md "Numerics"),
# so that is the floor any fp32 implementation can be compared at.
GRAD_TOL = 2e-2
# Gradients that are formed AFTER the backward pass has crossed the coarsest levels (enc*, dec5 of the backbone, dec5 of
# the U-decoder) are ill-conditioned on the small fixture scenes: level 5 holds 6..14 points, and a train-mode BatchNorm
# over 14 rows turns a 1e-7 forward perturbation anywhere upstream into a percent-level gradient change (measured:
# switching ONE encoder layer between two implementations that agree to 1e-6 moves dec5.0.linear1's gradient by 6e-2,
# an A/B of two such implementations).  Those are only sanity-bounded; every layer's own gradients are checked tightly in isolation
# (tests/test_gpu_fused_layer.py, tests/test_gpu_ops.py).
# (Round 3 note: `rgrad_dec4.linear2.0.weight` of fixture b1_3000 -- its forward input is the 11-row level-5 tensor -- sits next to one
# ReLU edge: 8.3e-4 from the fp64 evaluation on one side, 2.199e-2 on the other.  Round 2 saw the far side in about one of ten runs
# (float atomics); round 3, bit-reproducible, saw it DETERMINISTICALLY while the 1024-wide product of the dec5 head was formed as two
# separately rounded 512-wide passes, and not with one accumulation chain over both windows (csrc/rowlin.hip) -- the association the
# reference's single GEMM has.  The name stays in the strict set.)
WELL_CONDITIONED = ("cls", "dec1", "dec2", "dec3", "dec4", "confidence")


def well_conditioned(name):
    return name.startswith(WELL_CONDITIONED)


LOOSE_GRAD_TOL = 0.2
# With 8,192 points (BASELINE config 1) level 5 holds 32 points and the amplification is gone: every gradient of the HIP path sits
# within 1.5e-2 of the fp64 evaluation (the reference's own fp32 run: 2e-3; measured on MI355X, tools/grad_report.py), so the
# sanity bound shrinks to 5e-2 there; at 100k / 150k points the full distribution is checked (tests/test_gpu_fullsize.py).
LOOSE_GRAD_TOL_BY_POINTS = {8192: 5e-2}


def thin(a):
    return a[::ROW_STRIDE] if a.ndim >= 2 and a.shape[0] > 1000 else a


def max_rel(a, b):
    """max |a-b| / max |b|  (the 'rel' of the 1e-4 bar: error relative to the tensor's scale)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def l2_rel(a, b):
    """||a-b||_2 / ||b||_2.  For GRADIENTS of two fp32 implementations of the same layer: a ReLU whose pre-activation sits
    within rounding distance of zero (|bn(x)| ~ 1e-6 happens for ~1 in 10^5 elements) may open in one implementation and
    close in the other; that moves ONE channel's gradients by ~1e-2 of the tensor maximum in whichever implementation --
    fp64 sides with either (tools/pt_vs_unfused.py).  The Frobenius norm keeps such isolated kink flips at ~1e-3 while a
    systematically wrong kernel still shows up at 1e-2 and above; forward values and buffers stay on max_rel."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def assert_close(a, b, tol=REL_TOL, what=""):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    r = max_rel(a, b)
    assert r <= tol, f"{what}: max rel err {r:.3e} > {tol}"


class Wrap(torch.nn.Module):
    """Gives hooks the ``backbone.`` prefix that DefaultSegmentor provides."""

    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, d):
        return self.backbone(d)


def build_models(device="cpu"):
    model = Wrap(PointTransformerSeg50(in_channels=6, num_classes=13))
    recog = PTRecognizer()
    synthetic.fill_parameters_deterministic(model.backbone, seed=1)
    synthetic.fill_parameters_deterministic(recog, seed=2)
    return model.to(device), recog.to(device)


def run_case(name, train, device="cpu"):
    """Forward (+ backward in train mode) of backbone + U-decoder on a golden case; returns a dict shaped like the fixture."""
    sizes, gs = MODEL_CASES[name]
    batch = synthetic.make_batch(sizes, first_scene_id=100, grid_size=gs, device=device)
    model, recog = build_models(device)
    model.train(train)
    recog.train(train)
    mh = BaseModelHook(HOOK_CONFIG, clone_tensor=True, exclude_clone={"backbone": ["forward_output"]})
    mh.set_model(model)
    data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"])
    with mh:
        logits = model(data)
        conf = recog(mh)
    out = {"logits": logits, "conf": conf}
    for i in range(1, 6):
        p, x, o = mh[f"backbone.enc{i}"]["forward_output"]
        out[f"enc{i}_p"], out[f"enc{i}_x"], out[f"enc{i}_o"] = p, x, o
        out[f"dec{i}_x"] = mh[f"backbone.dec{i}.1"]["forward_output"][1]
    geom = model.backbone._last_geometry
    out["geometry"] = geom
    ce = torch.nn.CrossEntropyLoss(ignore_index=-1)
    seg_loss = ce(logits, batch["segment"])
    pseudo_mask = (torch.arange(logits.shape[0], device=logits.device) % 7) == 3
    segment_pseudo = batch["segment"].clone()
    segment_pseudo[pseudo_mask] = 13
    full = torch.cat([logits, conf], -1)
    rec_loss = ce(full, segment_pseudo) * 0.1
    out["seg_loss"], out["rec_loss"] = seg_loss, rec_loss
    out["score"] = full.softmax(-1)[:, -1]
    out["msp_score"] = -logits.log_softmax(-1).max(-1)[0]
    if train:
        (seg_loss + rec_loss).backward()
        out["named"] = dict(model.backbone.named_parameters())
        out["rnamed"] = dict(recog.named_parameters())
        out["state"] = model.backbone.state_dict()
    return out


def check_case_against_golden(out, g, train, tol=REL_TOL):
    """Compare a run_case() result with a golden fixture (np.load result).  Every check is made on this ONE evaluation: the HIP step is
    bit-reproducible since round 3 (no float atomics left on the path: tests/test_gpu_model.py::test_training_step_is_bit_reproducible),
    so there is nothing a second evaluation could change."""
    n = out["logits"].shape[0]
    # geometry: bit-exact
    geom = out["geometry"]
    for key in g.files:
        if key.startswith("fps_"):
            _, _, n_in, m_out = key.split("_")
            lvl = [i for i in range(len(geom.levels)) if geom.levels[i].p.shape[0] == int(n_in)][0]
            _, fps_idx = geom.down(lvl, 4)
            assert np.array_equal(fps_idx.cpu().numpy(), g[key]), f"FPS indices differ at {key}"
        elif key.startswith("knn_"):
            _, k, n_src, m_q = key.split("_")
            src = [i for i in range(len(geom.levels)) if geom.levels[i].p.shape[0] == int(n_src)][0]
            qry = [i for i in range(len(geom.levels)) if geom.levels[i].p.shape[0] == int(m_q)][0]
            idx, _ = geom.knn(int(k), src, qry)
            assert np.array_equal(idx.cpu().numpy(), g[key]), f"kNN indices differ at {key}"
    for i in range(1, 6):
        assert np.array_equal(out[f"enc{i}_o"].cpu().numpy(), g[f"enc{i}_o"])
        assert np.array_equal(thin(out[f"enc{i}_p"].detach().cpu().numpy()), g[f"enc{i}_p"]), f"enc{i} coords"
        assert_close(thin(out[f"enc{i}_x"].detach().cpu().numpy()), g[f"enc{i}_x"], tol, f"enc{i}_x")
        assert_close(thin(out[f"dec{i}_x"].detach().cpu().numpy()), g[f"dec{i}_x"], tol, f"dec{i}_x")
    for k in ["logits", "conf", "score", "msp_score", "seg_loss", "rec_loss"]:
        assert_close(out[k], g[k], tol, k)
    if train:
        report = {}
        for key in g.files:
            if key.endswith("#sum"):
                continue
            if key.startswith("grad_") or key.startswith("rgrad_"):
                named = out["named"] if key.startswith("grad_") else out["rnamed"]
                name = key.split("_", 1)[1]
                grad = named[name].grad.detach().cpu().numpy()
                part = grad[:GRAD_ROWS] if grad.ndim >= 2 else grad
                k64 = ("g64_" if key.startswith("grad_") else "rg64_") + name
                truth = g[k64]                      # the reference network evaluated in fp64
                scale = np.abs(truth).max() + 1e-30
                if scale < 1e-6:
                    continue  # analytically-zero gradients (biases in front of a train-mode BatchNorm): pure rounding noise
                ours = np.abs(part - truth).max() / scale
                ref32 = np.abs(g[key] - truth).max() / scale  # how far the reference's own fp32 run is from fp64
                report[name] = (ours, ref32)
                # as close to the fp64 evaluation as the reference's fp32 run is (x4 + 5e-3 slack), never worse than GRAD_TOL
                strict = well_conditioned(name)
                loose = LOOSE_GRAD_TOL_BY_POINTS.get(n, LOOSE_GRAD_TOL)
                bound = min(GRAD_TOL, 4 * ref32 + 5e-3) if strict else loose
                assert ours <= bound, f"{key}: ours-vs-fp64 {ours:.3e}, reference-fp32-vs-fp64 {ref32:.3e}"
                l2 = np.sqrt((grad.astype(np.float64) ** 2).sum())
                s64 = g[k64 + "#sum"]
                assert abs(l2 - s64[1]) <= (GRAD_TOL if strict else loose) * s64[1] + 1e-12, f"{key}: L2 norm {l2} vs {s64[1]}"
            elif key.startswith("buf_"):
                assert_close(out["state"][key[4:]], g[key], tol, key)
        out["grad_report"] = report
    return n


# ---- PointPdfV1.forward / trigger_operation fixture (tests/golden/make_golden.py: PDF_CASE / PDF_MODES / PDF_GRADS) ----
PDF_CASE = ("b2_2048_1600", [2048, 1600], 0.25)
PDF_MODES = {
    "train_pre": (True, 0, 2, False, True),
    "train_post": (True, 2, 2, False, True),
    "train_decay": (True, 4, 2, True, True),
    "eval_seg": (False, 2, 2, False, True),
    "eval_test": (False, 2, 2, False, False),
}
PDF_GRADS = ["model.backbone.cls.0.weight", "model.backbone.dec1.1.linear1.weight", "model.backbone.dec2.0.linear1.1.weight",
             "recognizer.recognizer.confidence.0.weight", "recognizer.recognizer.dec1.linear2.0.weight", "recognizer.recognizer.dec3.linear2.1.weight"]


def run_pdf_case(mode, device="cpu"):
    """DefaultSegmentor + PointPdf-v1m1 built through the registries and wired the way OpenSegTrainer.model_forward does
    (engines/train.py:373-380); returns (model_output, recognizer_output, step module)."""
    from pointcloudpdf_amd import engine
    from pointcloudpdf_amd.registry import MODELS, RECOGNIZER

    train, epoch, start_epoch, step_lw, with_segment = PDF_MODES[mode]
    _, sizes, gs = PDF_CASE
    batch = synthetic.make_batch(sizes, first_scene_id=100, grid_size=gs, device=device)
    ce = [dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)]

    class Step(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = MODELS.build(dict(type="DefaultSegmentor", backbone=dict(type="PointTransformer-Seg50", in_channels=6, num_classes=13), criteria=ce))
            self.recognizer = RECOGNIZER.build(dict(type="PointPdf-v1m1", recognizer=dict(type="PointTransformer-Recognizer"), criteria=ce,
                                                    loss_weight=0.1, step_loss_weight=step_lw, num_classes=13, start_epoch=start_epoch,
                                                    kp_ball_radius=0.1, kp_max_neighbor=34, condition_from="msp", beta=1.5, seed_from="ml",
                                                    seed_range=0.01, num_seed=20, slide_window=True,
                                                    pseudo_mask_fn=engine.default_pseudo_mask))

    step = Step()
    synthetic.fill_parameters_deterministic(step, seed=1)
    step = step.to(device)
    step.train(train)
    mh = BaseModelHook(HOOK_CONFIG, clone_tensor=True, exclude_clone={"backbone": ["forward_output"]}).set_model(step.model)
    step.recognizer.model_hooks = mh
    step.recognizer.set_epoch(epoch)
    d = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"])
    if with_segment:
        d["segment"] = batch["segment"]
    with mh:
        mo = step.model(d)
        ro = step.recognizer(d)
    if train:
        loss = mo["loss"] + ro["loss"] if "loss" in ro else mo["loss"]
        loss.backward()
    return mo, ro, step


def check_pdf_case(mode, mo, ro, step, g, tol=REL_TOL):
    """Against tests/golden/model_pointpdf_forward.npz (the reference's own DefaultSegmentor + PointPdfV1 classes)."""
    assert sorted(mo.keys()) == list(g[f"{mode}_model_keys"]), (mode, sorted(mo.keys()))
    assert sorted(ro.keys()) == list(g[f"{mode}_rec_keys"]), (mode, sorted(ro.keys()))
    for k, v in mo.items():
        assert tuple(v.shape) == g[f"{mode}_model_{k}"].shape
        assert_close(v, g[f"{mode}_model_{k}"], tol, f"{mode} model {k}")
    for k, v in ro.items():
        assert tuple(v.shape) == g[f"{mode}_rec_{k}"].shape, (mode, k, v.shape)
        assert_close(v, g[f"{mode}_rec_{k}"], tol, f"{mode} recognizer {k}")
    assert abs(float(step.recognizer.alpha) - float(g[f"{mode}_alpha"])) < 1e-12
    assert [p.requires_grad for p in step.recognizer.recognizer.parameters()] == list(g[f"{mode}_rec_requires_grad"])
    if PDF_MODES[mode][0]:
        named = dict(step.named_parameters())
        for k in PDF_GRADS:
            has = bool(g[f"{mode}_hasgrad_{k}"])
            assert (named[k].grad is not None) == has, (mode, k)
            if has:
                grad = named[k].grad.detach().cpu().numpy()
                ref = g[f"{mode}_grad_{k}"]
                part = grad[:GRAD_ROWS] if grad.ndim >= 2 else grad
                assert np.abs(part - ref).max() <= GRAD_TOL * (np.abs(ref).max() + 1e-30), (mode, k)
                l2 = np.sqrt((grad.astype(np.float64) ** 2).sum())
                assert abs(l2 - g[f"{mode}_grad_{k}#sum"][1]) <= GRAD_TOL * g[f"{mode}_grad_{k}#sum"][1] + 1e-12, (mode, k)


# ---- StratifiedTransformer (ST-v1m1) + ST-v1m1-Recognizer fixture (tests/golden/make_golden.py: ST_CFG / ST_SIZES / ST_GRADS) ----
ST_CFG = dict(downsample_scale=8, depths=[2, 2, 6, 2], channels=[48, 96, 192, 384], num_heads=[3, 6, 12, 24],
              window_size=[0.16, 0.32, 0.64, 1.28], up_k=3, grid_sizes=[0.04, 0.08, 0.16, 0.32], quant_sizes=[0.01, 0.02, 0.04, 0.08],
              rel_query=True, rel_key=True, rel_value=True, num_layers=4, concat_xyz=True, num_classes=13, ratio=0.25, k=16,
              prev_grid_size=0.04, sigma=1.0, stem_transformer=True, kp_ball_radius=0.04 * 2.5, kp_max_neighbor=34)
ST_SIZES, ST_GRID = [3000, 2500], 0.04
ST_HOOKS = {**{f"backbone.upsamples.{i}": ["forward_input", "forward_output"] for i in range(3)}, "backbone": ["forward_output"]}
ST_GRADS = ["stem_layer.0.kpconv.weight", "layers.0.blocks.0.attn.qkv.weight", "layers.0.blocks.1.attn.relative_pos_query_table",
            "layers.1.blocks.0.attn.relative_pos_value_table", "layers.2.blocks.3.mlp.fc1.weight", "layers.0.downsample.linear.weight",
            "layers.3.blocks.1.attn.proj.weight", "upsamples.0.linear2.1.weight", "upsamples.2.linear1.0.weight", "classifier.0.weight"]
ST_REC_GRADS = ["upsamples.0.linear1.1.weight", "upsamples.2.linear2.1.weight", "confidence.3.weight"]


def run_stratified_case(mode, device="cpu", geometry=None):
    """``geometry``: None = everything inside the forward; "fps" / "windows" = a StratifiedGeometry computed ahead of it."""
    from pointcloudpdf_amd import stratified  # noqa: F401  (registers ST-v1m1 / ST-v1m1-Recognizer)
    from pointcloudpdf_amd.registry import MODELS

    train, dpr = {"train": (True, 0.0), "eval": (False, 0.3)}[mode]
    batch = synthetic.make_batch(ST_SIZES, first_scene_id=300, grid_size=ST_GRID, device=device)
    model = Wrap(MODELS.build(dict(type="ST-v1m1", drop_path_rate=dpr, **ST_CFG)))
    recog = MODELS.build(dict(type="ST-v1m1-Recognizer", up_k=3, channels=ST_CFG["channels"], num_layers=4))
    synthetic.fill_parameters_deterministic(model.backbone, seed=11)
    synthetic.fill_parameters_deterministic(recog, seed=12)
    model, recog = model.to(device), recog.to(device)
    model.train(train); recog.train(train)
    mh = BaseModelHook(ST_HOOKS, clone_tensor=True, exclude_clone={"backbone": ["forward_output"]}).set_model(model)
    data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"])
    if geometry is not None:
        bb = model.backbone
        data["st_geometry"] = bb.make_geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute(
            bb.layers_by_level() if geometry == "windows" else None)
    with mh:
        logits = model(data)
        conf = recog(mh)
    out = dict(logits=logits, conf=conf, hooks=mh, model=model, recog=recog, geometry=data.get("st_geometry"))
    if train:
        ce = torch.nn.CrossEntropyLoss(ignore_index=-1)
        loss = ce(logits, batch["segment"]) + 0.1 * ce(torch.cat([logits, conf], -1), batch["segment"].clamp(min=0))
        loss.backward()
        out["loss"] = loss
    return out


def check_stratified_case(mode, out, g, tol=REL_TOL, grad_tol=GRAD_TOL):
    assert_close(out["logits"], g[f"{mode}_logits"], tol, f"ST {mode} logits")
    assert_close(out["conf"], g[f"{mode}_conf"], tol, f"ST {mode} conf")
    for i in range(3):
        fi, fo = out["hooks"][f"backbone.upsamples.{i}"]["forward_input"], out["hooks"][f"backbone.upsamples.{i}"]["forward_output"]
        shapes = np.array([tuple(t.shape) + (0,) * (2 - t.dim()) for t in fi])
        assert np.array_equal(shapes, g[f"{mode}_up{i}_in_shapes"]), (i, shapes)
        assert_close(thin(fo[0].detach().cpu().numpy()), g[f"{mode}_up{i}_out"], tol, f"ST {mode} upsample {i}")
    if mode == "train":
        assert_close(out["loss"], g["train_loss"], tol, "ST loss")
        named, rnamed = dict(out["model"].backbone.named_parameters()), dict(out["recog"].named_parameters())
        for prefix, names, table in (("grad_", ST_GRADS, named), ("rgrad_", ST_REC_GRADS, rnamed)):
            for k in names:
                grad = table[k].grad.detach().cpu().numpy()
                ref = g[prefix + k]
                part = grad[:GRAD_ROWS] if grad.ndim >= 2 else grad
                assert np.abs(part - ref).max() <= grad_tol * (np.abs(ref).max() + 1e-30), (k, np.abs(part - ref).max(), np.abs(ref).max())
                l2 = np.sqrt((grad.astype(np.float64) ** 2).sum())
                assert abs(l2 - g[prefix + k + "#sum"][1]) <= grad_tol * g[prefix + k + "#sum"][1] + 1e-12, k
