"""What a torch reduction inside a captured training step does on this stack (ROCm 7.2, MI355X): the TransitionUp head's per-scene sums
as torch ops (PDFOPS_HEAD_TORCH_SUMS=1: rounds 1-4) against the kernels of csrc/scene_rows.hip (default), on 2 x 131,200 points -- level 5
then has 512 rows per scene, the size from which torch reduces dim 0 over several workgroups with a semaphore it clears by hipMemsetAsync
(a memset node in the captured step).  Every replay runs on the same parameters and the same batch, alternately right after another replay
and right after an eager pre-pass of another batch; the gradients are compared with the eager step's.
Usage on the GPU box: python tools/probes/replay_reduction_probe.py  (runs both settings in child processes)"""
import copy
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from pointcloudpdf_amd import engine, synthetic
    from pointcloudpdf_amd.geometry import Geometry

    dev = torch.device("cuda")
    n = int(os.environ.get("PROBE_POINTS", "131200"))
    step = engine.OpenSegStep().to(dev)
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    opt = engine.FusedSGD(step.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    train = engine.TrainStep(step, opt, graph=True)
    pool = [synthetic.make_batch([n] * 2, first_scene_id=10 * i, device=dev) for i in range(2)]
    keys = ("coord", "feat", "offset", "offset_host", "segment")
    inline = lambda b: Geometry(b["coord"], b["offset"], b["offset_host"]).precompute()
    batch = lambda j, g: dict({k: pool[j][k] for k in keys}, pdf_geometry=g)
    geoms = [inline(b) for b in pool]
    train(batch(0, geoms[0]))
    state = copy.deepcopy(step.state_dict())

    def grads(eager=False, before=None):
        step.load_state_dict(state)
        if before is not None:
            before()
        train(batch(1, inline(pool[1]) if eager else geoms[1]), eager=eager)
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in step.parameters()]

    ref = grads(eager=True)
    wrong = {"after a replay": 0, "after an eager pre-pass": 0}
    worst = 0.0
    trials = 12
    for i in range(trials):
        for name, before in (("after an eager pre-pass", lambda: inline(pool[0])), ("after a replay", None)):
            g = grads(before=before)
            bad = [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(g, ref) if not torch.equal(a, b)]
            wrong[name] += 1 if bad else 0
            worst = max([worst] + bad)
    print(f"  level-5 rows per scene {int(geoms[0].levels[-1].p.shape[0]) // 2}; replays with gradients that differ from the eager step's: "
          + ", ".join(f"{k}: {v} of {trials}" for k, v in wrong.items()) + f"; worst relative deviation of a parameter's gradient {worst:.3g}", flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
    else:
        for label, env in (("torch sums in the TransitionUp head (rounds 1-4)", {"PDFOPS_HEAD_TORCH_SUMS": "1"}), ("csrc/scene_rows.hip (round 5)", {})):
            print(label + ":", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PROBE_CHILD="1", **env), check=False)
