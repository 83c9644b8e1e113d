"""CPU suite: the N > 1 path (whole scenes sharded over ranks, one gradient all-reduce, max-over-ranks timing) with
world_size 2 over gloo.  The ops run on the CPU oracle here (test-only injection); the product path is HIP + RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes_per_rank, out_dir, mode="ddp"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import oracle
    from pointcloudpdf_amd import _native, engine, synthetic

    _native._set_backend_for_testing(oracle.backend())
    r, lr, w = engine.init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    step = engine.OpenSegStep(backbone="PointTransformer-Seg26")
    synthetic.fill_parameters_deterministic(step, seed=1)
    step.train()
    if mode == "flat":   # perturb the non-zero ranks: the constructor must broadcast rank 0's parameters
        if rank > 0:
            with torch.no_grad():
                for p in step.parameters():
                    p.add_(0.01)
        sync = engine.FlatGradAllReduce(step)
        ddp = step
    else:
        ddp = engine.wrap_ddp(step, torch.device("cpu"))
    # whole scenes are the sharding unit: global scene list -> this rank's scenes
    scene_ids = engine.shard_scene_ids(len(sizes_per_rank) * world, rank, world)
    sizes = [sizes_per_rank[i // world] for i in scene_ids]
    batch = synthetic.make_batch(sizes, first_scene_id=scene_ids[0], grid_size=0.3)
    out = ddp(batch)
    out["loss"].backward()
    if mode == "flat":
        sync.sync()
    grads = {n: p.grad.clone() for n, p in step.named_parameters() if p.grad is not None}
    # reference: average of per-rank local gradients, computed without DDP
    torch.save(dict(loss=out["loss"].detach(), grads=grads, scene_ids=scene_ids, n=batch["coord"].shape[0]),
               os.path.join(out_dir, f"ddp_{rank}.pt"))
    local = engine.OpenSegStep(backbone="PointTransformer-Seg26")
    synthetic.fill_parameters_deterministic(local, seed=1)
    local.train()
    lo = local(batch)
    lo["loss"].backward()
    torch.save({n: p.grad.clone() for n, p in local.named_parameters() if p.grad is not None}, os.path.join(out_dir, f"local_{rank}.pt"))
    # bench.py's timing contract: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_two_rank_gradient_allreduce(tmp_path, mode):
    """mode "ddp": torch DistributedDataParallel; mode "flat": engine.FlatGradAllReduce (bench.py's default for N > 1)."""
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, [1100, 900], str(tmp_path), mode), nprocs=world, join=True)
    d = [torch.load(tmp_path / f"ddp_{r}.pt") for r in range(world)]
    loc = [torch.load(tmp_path / f"local_{r}.pt") for r in range(world)]
    assert d[0]["scene_ids"] == [0, 2] and d[1]["scene_ids"] == [1, 3]  # disjoint whole scenes
    assert d[0]["n"] == d[1]["n"] == 2000
    assert abs(d[0]["loss"].item() - d[1]["loss"].item()) > 0  # different scenes per rank
    gscale = max(((loc[0][n] + loc[1][n]) / 2).abs().max().item() for n in d[0]["grads"])
    for name, g0 in d[0]["grads"].items():
        g1 = d[1]["grads"][name]
        assert torch.equal(g0, g1), f"{name}: ranks disagree after all-reduce"
        mean_local = (loc[0][name] + loc[1][name]) / 2
        scale = mean_local.abs().max().item() + 1e-12
        # analytically-zero gradients (biases in front of a train-mode BatchNorm) are rounding noise: floor at 1e-5 of the
        # largest gradient in the model
        assert (g0 - mean_local).abs().max().item() <= 1e-5 * scale + 1e-5 * gscale, name


def test_shard_scene_ids_partition():
    from pointcloudpdf_amd import engine

    ids = [engine.shard_scene_ids(16, r, 8) for r in range(8)]
    assert sorted(sum(ids, [])) == list(range(16)) and all(len(i) == 2 for i in ids)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts its own ranks (no torchrun needed) and fails loudly, before any rank is started, when the
    box has fewer GPUs -- it never falls back to timing one GPU under an n_gpus = N label."""
    import subprocess
    import sys

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has GPUs: the launch path itself is covered by tests/test_gpu_model.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but only" in r.stderr and r.stdout.strip() == ""


def test_bench_preflight_fails_loudly_when_the_node_cannot_hold_the_ranks():
    """bench.preflight_host: the check `--gpus N` runs before any rank is started (and rank 0 runs under torchrun) -- N ranks need a
    core each and ~3.5 GiB of host memory each; a node that cannot hold them exits 3 with the reason instead of losing a rank to the
    OOM killer minutes into the run."""
    import io

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    ok = bench.preflight_host(1, 2, 100000, 20, out=io.StringIO())
    assert ok["ranks"] == 1 and ok["host_cores"] >= 1 and ok["lookahead_batches_per_rank"] == 20
    msg = io.StringIO()
    with pytest.raises(SystemExit) as e:
        bench.preflight_host(100000, 2, 100000, 20, out=msg)
    assert e.value.code == 3 and "pre-flight FAILED" in msg.getvalue() and "100000 ranks" in msg.getvalue()


def _eval_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from pointcloudpdf_amd import evaluator

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ev = evaluator.OpenSegEvaluator(6, unknown_label=[4], ignore_index=-1)
    for b in range(3):
        g = torch.Generator().manual_seed(100 * b + rank)
        logits = torch.randn(500, 6, generator=g)
        score = torch.rand(500, generator=g)
        seg = torch.randint(0, 6, (500,), generator=g)
        if rank == 1 and b == 1:
            seg[seg == 4] = 0   # this rank's batch holds no unknown point: its (None, None) pair is gathered and skipped
        ev.update(logits, score, seg)
    torch.save(ev.summary(), os.path.join(out_dir, f"eval_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_evaluator_gathers_every_ranks_recognition_metrics(tmp_path):
    """engines/hooks/evaluator.py:199-221: every rank appends EVERY rank's (aupr, auroc) of a batch, so the summaries agree across
    ranks and equal the one-process evaluation of all batches."""
    from pointcloudpdf_amd import evaluator

    world = 2
    mp.spawn(_eval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    s = [torch.load(tmp_path / f"eval_{r}.pt", weights_only=False) for r in range(world)]
    one = evaluator.OpenSegEvaluator(6, unknown_label=[4], ignore_index=-1)
    for b in range(3):
        for rank in range(world):
            g = torch.Generator().manual_seed(100 * b + rank)
            logits, score, seg = torch.randn(500, 6, generator=g), torch.rand(500, generator=g), torch.randint(0, 6, (500,), generator=g)
            if rank == 1 and b == 1:
                seg[seg == 4] = 0
            one.update(logits, score, seg)
    ref = one.summary()
    for key in ["mIoU", "mAcc", "allAcc", "aupr", "auroc"]:
        assert s[0][key] == s[1][key], key
        assert abs(s[0][key] - ref[key]) <= 1e-12, (key, s[0][key], ref[key])
