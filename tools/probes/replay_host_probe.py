"""Host time of one replay of the captured training step (2 x 100k points) with ROCm's graph packet capture off (the package's default:
the runtime walks the graph's ~920 kernel nodes on the calling thread) and on (pre-recorded AQL packets), next to the device time of the
replay.  Usage on the GPU box: python tools/probes/replay_host_probe.py   (runs both settings in child processes)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from pointcloudpdf_amd import engine, synthetic
    from pointcloudpdf_amd.geometry import Geometry

    dev = torch.device("cuda")
    n = int(os.environ.get("PROBE_POINTS", "100000"))
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    batch = synthetic.make_batch([n] * 2, first_scene_id=10, device=dev)
    geom = Geometry(batch["coord"], batch["offset"], batch["offset_host"]).precompute()
    cap = engine.CapturedStep(step, batch, geom=geom, debug_graph=True)
    census = cap.node_census()
    for _ in range(3):
        cap(batch, geom)
    torch.cuda.synchronize()
    reps = 20
    host, devt = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        cap.graph.replay()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        host.append((t1 - t0) * 1e3)
        devt.append(e0.elapsed_time(e1))
    host.sort(); devt.sort()
    print(f"  graph nodes {census}; host time of graph.replay() median {host[reps // 2]:.2f} ms (min {host[0]:.2f}); "
          f"device time of the replay median {devt[reps // 2]:.2f} ms (min {devt[0]:.2f})", flush=True)
    # back-to-back replays without a synchronisation between them: what the training loop sees
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cap.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  {reps} replays queued back to back: host enqueue {1e3 * (t1 - t0) / reps:.2f} ms per replay, wall {1e3 * (t2 - t0) / reps:.2f} ms per replay", flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
    else:
        for setting in ("0", "1"):
            print(f"DEBUG_CLR_GRAPH_PACKET_CAPTURE={setting}", flush=True)
            env = dict(os.environ, PROBE_CHILD="1", DEBUG_CLR_GRAPH_PACKET_CAPTURE=setting)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
