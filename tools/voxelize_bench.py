#!/usr/bin/env python3
"""GPU probe: GridSample on the device (pdf_grid_hash + device sorts) for a batch of raw scenes vs the numpy restatement on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from pointcloudpdf_amd import voxelize, _native
from oracle import voxel
from test_voxelize import dense_scene

be = _native.hip_backend()
n, b, gs = 1000000, 4, 0.04
scenes = [dense_scene(60 + i, n) for i in range(b)]
coord = torch.from_numpy(np.concatenate(scenes)).cuda()
off = torch.tensor([n * (i + 1) for i in range(b)], dtype=torch.int32, device="cuda")
offh = [n * (i + 1) for i in range(b)]
min_grid = torch.zeros(b, 3, dtype=torch.int64, device="cuda")


def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


us_hash = t(lambda: be.grid_hash(coord, off, [gs] * 3, min_grid))
us_all = t(lambda: voxelize.grid_sample(coord, off, gs, offset_host=offh), iters=5)
t0 = time.perf_counter(); voxel.grid_partition(scenes[0], gs); cpu = time.perf_counter() - t0
N = n * b
print(f"{b} scenes x {n} raw points, grid {gs}")
print(f"pdf_grid_hash           {us_hash:9.1f} us   {44 * N / us_hash / 1e3:8.1f} GB/s algorithmic (12 + 24 + 8 B per point)   {N / us_hash:8.1f} M points/s")
print(f"voxelize.grid_sample    {us_all:9.1f} us   {N / us_all:8.1f} M points/s (keys + 2 stable sorts + partition + train pick)")
print(f"numpy restatement, 1 scene, 1 core: {cpu * 1e3:.1f} ms = {n / cpu / 1e6:.2f} M points/s")
