"""Which piece of the backward breaks hipGraph capture (capture_end segfaults for the whole step)?  Each case = one module's forward +
backward captured alone in a subprocess.   python tools/graph_probe2.py [case]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["seg26", "seg50", "step26", "step50", "bottleneck32", "bottleneck64", "bottleneck512", "td", "tu", "tu_head", "cls_ce", "interp", "zero_big", "clone_bwd"]


def run_case(case):
    import torch
    from pointcloudpdf_amd import synthetic, pointops
    from pointcloudpdf_amd.geometry import Geometry
    from pointcloudpdf_amd import point_transformer as pt
    from pointcloudpdf_amd import segmentor  # noqa: F401

    dev = torch.device("cuda", 0)
    npts = int(os.environ.get("POINTS", "20000"))
    batch = synthetic.make_batch([npts, npts], first_scene_id=3, device=dev)
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
    g = torch.Generator(device="cuda").manual_seed(1)
    mods = []

    def mk(m):
        m = m.to(dev)
        synthetic.fill_parameters_deterministic(m, seed=3)
        m.train()
        mods.append(m)
        return m

    if case in ("seg26", "seg50"):
        from pointcloudpdf_amd.registry import MODELS
        m = mk(MODELS.build(dict(type="DefaultSegmentor", backbone=dict(type="PointTransformer-S" + case[1:], in_channels=6, num_classes=13),
                                 criteria=[dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1)])))
        data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"], offset_host=batch["offset_host"], segment=batch["segment"],
                    pdf_geometry=geom)
        fn = lambda: m(dict(data))["loss"]
    elif case in ("step26", "step50"):
        from pointcloudpdf_amd import engine
        m = mk(engine.OpenSegStep(backbone="PointTransformer-Seg" + case[4:]))
        data = dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"], offset_host=batch["offset_host"], segment=batch["segment"],
                    pdf_geometry=geom)
        fn = lambda: m(dict(data))["loss"]
    elif case.startswith("bottleneck"):
        c = int(case[len("bottleneck"):])
        lvl = {32: 0, 64: 1, 512: 4}[c]
        blk = mk(pt.Bottleneck(c, c, 8, 8 if c == 32 else 16))
        x = torch.randn(geom.coord(lvl).shape[0], c, device=dev, generator=g).requires_grad_(True)
        fn = lambda: blk([geom.coord(lvl), x, geom.offset(lvl)])[1].sum()
    elif case == "td":
        td = mk(pt.TransitionDown(32, 64, 4, 16))
        x = torch.randn(geom.coord(0).shape[0], 32, device=dev, generator=g).requires_grad_(True)
        fn = lambda: td([geom.coord(0), x, geom.offset(0)])[1].sum()
    elif case == "tu":
        tu = mk(pt.TransitionUp(64, 32))
        x1 = torch.randn(geom.coord(0).shape[0], 32, device=dev, generator=g).requires_grad_(True)
        x2 = torch.randn(geom.coord(1).shape[0], 64, device=dev, generator=g).requires_grad_(True)
        fn = lambda: tu([geom.coord(0), x1, geom.offset(0)], [geom.coord(1), x2, geom.offset(1)]).sum()
    elif case == "tu_head":
        tu = mk(pt.TransitionUp(512))
        x = torch.randn(geom.coord(4).shape[0], 512, device=dev, generator=g).requires_grad_(True)
        fn = lambda: tu([geom.coord(4), x, geom.offset(4)]).sum()
    elif case == "cls_ce":
        import torch.nn as nn
        from pointcloudpdf_amd.registry import MODELS  # noqa: F401
        cls = mk(nn.Sequential(nn.Linear(32, 32), nn.BatchNorm1d(32), nn.ReLU(inplace=True), nn.Linear(32, 13)))
        from pointcloudpdf_amd.segmentor import CrossEntropyLoss
        ce = CrossEntropyLoss(ignore_index=-1)
        x = torch.randn(40000, 32, device=dev, generator=g).requires_grad_(True)
        fn = lambda: ce(pt._seq(cls, x), batch["segment"])
    elif case == "interp":
        x2 = torch.randn(geom.coord(1).shape[0], 64, device=dev, generator=g).requires_grad_(True)
        fn = lambda: pointops.interpolation(geom.coord(1), geom.coord(0), x2, geom.offset(1), geom.offset(0)).sum()
    elif case == "zero_big":
        x = torch.randn(1 << 20, device=dev).requires_grad_(True)
        fn = lambda: (torch.zeros(3 << 20, device=dev)[:1 << 20] + x).sum()
    elif case == "clone_bwd":
        x = torch.randn(1 << 20, device=dev).requires_grad_(True)
        fn = lambda: (x.clone() * 2 + x).sum()
    params = [p for m in mods for p in m.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            for p in params:
                p.grad = None
            fn().backward()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        loss = fn()
        loss.backward()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    print(f"CASE {case}: ok loss {float(loss):.6g}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
            ok = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
            print(ok[0] if ok else f"CASE {c}: FAILED rc={r.returncode} {r.stderr.strip().splitlines()[-12:]}", flush=True)
