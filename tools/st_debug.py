"""Localise a CPU-oracle vs HIP difference in the StratifiedTransformer: first module whose output differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import helpers, oracle
from pointcloudpdf_amd import _native, synthetic
from pointcloudpdf_amd.registry import MODELS
from pointcloudpdf_amd import stratified  # noqa
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
def run(device, backend):
    prev = _native._set_backend_for_testing(backend) if backend is not None else None
    try:
        train, dpr = {"train": (True, 0.0), "eval": (False, 0.3)}[mode]
        batch = synthetic.make_batch(helpers.ST_SIZES, first_scene_id=300, grid_size=helpers.ST_GRID, device=device)
        model = MODELS.build(dict(type="ST-v1m1", drop_path_rate=dpr, **helpers.ST_CFG))
        synthetic.fill_parameters_deterministic(model, seed=11)
        model = model.to(device); model.train(train)
        outs = {}
        def mk(name):
            def hook(m, i, o):
                t = o[0] if isinstance(o, (tuple, list)) else o
                if isinstance(t, torch.Tensor) and t.is_floating_point():
                    outs.setdefault(name, []).append(t.detach().float().cpu())
            return hook
        for n, m in model.named_modules():
            if n: m.register_forward_hook(mk(n))
        logits = model(dict(coord=batch["coord"], feat=batch["feat"], offset=batch["offset"]))
        return outs, logits.detach().cpu()
    finally:
        if backend is not None: _native._set_backend_for_testing(prev)
torch.backends.cuda.matmul.allow_tf32 = False
ro, rl = run("cpu", oracle.backend())
go, gl = run("cuda", None)
print("logits", helpers.max_rel(gl.numpy(), rl.numpy()))
shown = 0
for n in ro:
    for j, (a, b) in enumerate(zip(ro[n], go[n])):
        if a.shape != b.shape:
            print("SHAPE", n, j, a.shape, b.shape); shown += 1; continue
        e = helpers.max_rel(b.numpy(), a.numpy())
        if e > 1e-4 and shown < 25:
            print(f"{n}[{j}] {e:.2e} shape {tuple(a.shape)}"); shown += 1
