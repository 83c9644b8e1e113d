import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, helpers
from pointcloudpdf_amd.point_transformer import PointTransformerLayer
torch.backends.cuda.matmul.allow_tf32 = False
for name in ["b2_2048_1600"]:
    g = np.load(os.path.join(ROOT, "tests/golden", f"model_{name}_train.npz"))
    for fused in (True, False):
        PointTransformerLayer.fused = fused
        out = helpers.run_case(name, True, device="cuda")
        rep = {}
        for key in g.files:
            if key.startswith("grad_") and not key.endswith("#sum"):
                nm = key[5:]
                grad = out["named"][nm].grad.detach().cpu().numpy()
                part = grad[:16] if grad.ndim >= 2 else grad
                t = g["g64_" + nm]; sc = np.abs(t).max() + 1e-30
                if sc < 1e-6: continue
                rep[nm] = (np.abs(part - t).max() / sc, np.abs(g[key] - t).max() / sc)
        print("fused" if fused else "composed", "logits err", helpers.max_rel(out["logits"].detach().cpu().numpy(), g["logits64"]))
        for k, v in rep.items():
            print(f"   {k:45s} ours {v[0]:.2e}   ref32 {v[1]:.2e}")
