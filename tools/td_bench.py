"""TransitionDown (stride 4) forward+backward alone on the level shapes of a 2 x 100k batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import Geometry
from pointcloudpdf_amd.point_transformer import TransitionDown
reps = int(os.environ.get("REPS", "8"))
batch = synthetic.make_batch([100000, 100000], device="cuda")
geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
planes = [32, 64, 128, 256, 512]
for lv in range(4):
    cin, cout = planes[lv], planes[lv + 1]
    td = TransitionDown(cin, cout, 4, 16).cuda().train()
    p, o = geom.coord(lv), geom.offset(lv)
    x = torch.randn(p.shape[0], cin, device="cuda", requires_grad=True)
    go = torch.randn(geom.coord(lv + 1).shape[0], cout, device="cuda")
    tf = tb = 0.0
    for it in range(reps + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); y = td([p, x, o])[1]; e[1].record(); y.backward(go); e[2].record(); torch.cuda.synchronize()
        if it >= 2: tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"TD level {lv}->{lv + 1}: n={p.shape[0]} m={go.shape[0]} {cin}->{cout}  fwd {tf / reps * 1e3:8.1f} us  bwd {tb / reps * 1e3:8.1f} us", flush=True)
