"""Worker of test_two_ranks_share_one_gpu_and_exchange_gradients (tests/test_gpu_model.py): one data-parallel rank of a 2-rank job whose
ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device).  argv: mode (eager | graph), output path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from pointcloudpdf_amd import engine, synthetic
from pointcloudpdf_amd.geometry import Geometry

mode, path = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
step = engine.OpenSegStep(backbone="PointTransformer-Seg26").to(dev)
if rank == 0:
    synthetic.fill_parameters_deterministic(step, seed=3)     # rank 1 keeps its random init: the exchange's broadcast must overwrite it
step.train()
sync = engine.FlatGradAllReduce(step)                         # broadcasts rank 0's parameters
batch = synthetic.make_batch([6000, 5000], first_scene_id=10 * rank + 1, device=dev)   # whole scenes per rank, different ones
geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
if mode == "graph":
    cap = engine.CapturedStep(step, batch, geom=geom)
    out = cap(batch, geom)
else:
    out = step(dict(batch, pdf_geometry=geom))
    out["loss"].backward()
params = dict(step.named_parameters())
local = {n: p.grad.detach().clone().cpu() for n, p in params.items() if p.grad is not None}
sync.sync()
synced = {n: p.grad.detach().clone().cpu() for n, p in params.items() if p.grad is not None}
torch.save(dict(loss=float(out["loss"]), local=local, synced=synced, weights={n: p.detach().cpu() for n, p in list(params.items())[:4]}), path)
dist.barrier()
dist.destroy_process_group()
