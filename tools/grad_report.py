import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, helpers
torch.backends.cuda.matmul.allow_tf32 = False
for name in ("b1_8192", "b2_2048_1600"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"model_{name}_train.npz"))
    out = helpers.run_case(name, True, device="cuda")
    helpers.check_case_against_golden(out, g, True)
    print(name, "logits", helpers.max_rel(out["logits"].detach().cpu().numpy(), g["logits"]))
    for k, (a, b) in out["grad_report"].items():
        print(f"   {k:48s} ours-vs-fp64 {a:.2e}   reference-fp32-vs-fp64 {b:.2e}")
