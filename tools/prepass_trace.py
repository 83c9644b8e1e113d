"""The geometry pre-pass of a 12-batch group alone (for `rocprofv3 --kernel-trace --stats -- python3 tools/prepass_trace.py`): which kernels
the coordinate-only work of a training step consists of, without the training stream beside it.  Prints wall ms per group."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointcloudpdf_amd import synthetic
from pointcloudpdf_amd.geometry import GeometryPrefetcher


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-fps-marker", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pf = GeometryPrefetcher(depth=1)
    for r in range(a.reps + 1):
        batches = [synthetic.make_batch([100000, 100000], first_scene_id=100 * r + 2 * i, device=dev) for i in range(a.group)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tickets = pf.submit_group(batches)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"group of {a.group}: host {1e3 * (t1 - t0):.1f} ms, done after {1e3 * (t2 - t0):.1f} ms", flush=True)
        del tickets


if __name__ == "__main__":
    main()
