import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from pointcloudpdf_amd import engine, synthetic
dev = torch.device("cuda")
step = engine.OpenSegStep().to(dev); synthetic.fill_parameters_deterministic(step, seed=1); step.train()
opt = torch.optim.SGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
b = synthetic.make_batch([100000, 100000], device=dev)
def one():
    opt.zero_grad(set_to_none=True)
    out = step(dict(coord=b["coord"], feat=b["feat"], offset=b["offset"], offset_host=b["offset_host"], segment=b["segment"]))
    out["loss"].backward(); opt.step()
for _ in range(2): one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    one(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("top ops by count")
for e in rows[:45]:
    print(f"{e.count:6d}  cpu_total {e.cpu_time_total/1e3:8.2f} ms  self_cpu {e.self_cpu_time_total/1e3:8.2f} ms  dev {e.device_time_total/1e3:8.2f} ms  {e.key[:80]}")
print("total self cpu ms", sum(e.self_cpu_time_total for e in ka) / 1e3)
# who issues the device-to-device copies / which autograd nodes own the launches
from collections import Counter
own, launches = Counter(), Counter()
for ev in prof.events():
    if ev.name in ("hipMemcpyAsync", "hipMemcpyWithStream", "hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel", "hipMemsetAsync"):
        p, chain = ev.cpu_parent, []
        while p is not None:
            chain.append(p.name); p = p.cpu_parent
        key = " < ".join(chain[:3]) if chain else "(none)"
        (own if "Memcpy" in ev.name else launches)[key] += 1
print("memcpy owners")
for k, v in own.most_common(25): print(f"{v:6d}  {k[:160]}")
print("launch owners")
for k, v in launches.most_common(40): print(f"{v:6d}  {k[:160]}")
