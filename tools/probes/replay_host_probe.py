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
    # ... and with ONE eagerly launched kernel between two replays (what the optimizer / the staging copy are in the training loop):
    # the device timeline of a training run shows ~0.45 ms of idle time at every graph <-> eager transition (profiles/r06_z_timeline.txt)
    x = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cap.graph.replay()
        x.add_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  {reps} x (replay, one eager elementwise kernel): wall {1e3 * (t2 - t0) / reps:.2f} ms per iteration", flush=True)
    # how far AHEAD of the device the host is when replay() returns: host work of w ms between the replay and the next launch shows up as
    # device idle time once w exceeds that lead (the training loop does ~1 ms of Python per step there)
    out = []
    for w in (0.0, 0.1, 0.2, 0.4, 0.8, 1.6, 3.2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cap.graph.replay()
            t = time.perf_counter()
            while time.perf_counter() - t < w * 1e-3:
                pass
            x.add_(1.0)
        torch.cuda.synchronize()
        out.append(f"{w:g} ms -> {1e3 * (time.perf_counter() - t0) / reps:.2f}")
    print("  wall per (replay, host work, eager kernel) by host work: " + "; ".join(out), flush=True)
    # what the training loop has between two replays besides kernels: a pinned host -> device copy (the optimizer's pointer table) and a
    # wait for an event of another stream (the pre-pass's, long complete)
    host = torch.zeros(2436, dtype=torch.int64).pin_memory()
    tab = torch.zeros(2436, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y = torch.zeros(16, device=dev)
        done = torch.cuda.Event()
        done.record(side)
    torch.cuda.synchronize()
    variants = {"kernel only": lambda: x.add_(1.0),
                "pinned H2D copy + kernel": lambda: (tab.copy_(host, non_blocking=True), x.add_(1.0)),
                "wait_event(other stream) + kernel": lambda: (torch.cuda.current_stream().wait_event(done), x.add_(1.0)),
                "event record + kernel": lambda: (torch.cuda.Event().record(), x.add_(1.0)),
                "side-stream kernel + event + wait_event + kernel": None}
    for name, fn in variants.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cap.graph.replay()
            if fn is not None:
                fn()
            else:
                with torch.cuda.stream(side):
                    y.add_(1.0)
                    e = torch.cuda.Event()
                    e.record(side)
                torch.cuda.current_stream().wait_event(e)
                x.add_(1.0)
        torch.cuda.synchronize()
        print(f"  replay + {name}: wall {1e3 * (time.perf_counter() - t0) / reps:.2f} ms per iteration", flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
    else:
        extra = [dict(kv.split("=", 1) for kv in a.split(",")) for a in sys.argv[1:]] or [{}]
        for setting in ("0", "1"):
            for more in extra:
                print(f"DEBUG_CLR_GRAPH_PACKET_CAPTURE={setting} {more}", flush=True)
                env = dict(os.environ, PROBE_CHILD="1", DEBUG_CLR_GRAPH_PACKET_CAPTURE=setting, **more)
                subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
