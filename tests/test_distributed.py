"""Data-parallel path on CPU: world_size-2 gloo processes (127.0.0.1).  The compute on each rank is
the CPU oracle (the HIP kernels need a GPU); what is under test is the product's sharding, flat
parameter/gradient bucket and the single all-reduce: the averaged per-rank gradients must equal
the single-process gradient of the global batch, and every rank must end with identical buckets."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, build_graphs, experiment, oracle_model


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(seed=42):
    cfg = experiment("baseline", mesh_levels=[0])
    g = build_graphs(cfg)
    torch.manual_seed(seed)
    m = oracle_model(cfg, g)
    gen = torch.Generator().manual_seed(1234)
    X = torch.randn(4, g["G"], 66, generator=gen)
    y = X[..., 33:] + 0.1 * torch.randn(4, g["G"], 33, generator=gen)
    return m, X, y


def _worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphcast_lite_amd.train import FlatParams, allreduce_gradients, shard_batch
    from oracle import train_step as T

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m, X, y = _make()
    flat = FlatParams(m)  # parameters re-pointed into one bucket, .grad slices installed
    Xl, yl = shard_batch(X, y, rank, world)
    assert Xl.shape[0] == X.shape[0] // world
    flat.zero_grad()
    loss = T.train_step_loss(m, Xl, yl, lat_weights=T.get_lat_weights(32, 64))
    loss.backward()  # autograd accumulates into the installed bucket slices
    scale = allreduce_gradients(flat, world)
    g = flat.grad * scale
    # every rank holds the same reduced bucket
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        torch.save({"grad": g, "numel": flat.numel, "loss": loss.detach()}, os.path.join(out_dir, "dp.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_equals_global_batch(tmp_path):
    from graphcast_lite_amd.train import FlatParams
    from oracle import train_step as T

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "dp.pt"))
    m, X, y = _make()
    flat = FlatParams(m)
    assert flat.numel == got["numel"] and flat.numel % 64 == 0  # bucket length: every parameter padded to 256 B
    assert flat.num_params == 53784  # baseline parameter count, shared PReLU counted once
    flat.zero_grad()
    T.train_step_loss(m, X, y, lat_weights=T.get_lat_weights(32, 64)).backward()
    rel = ((got["grad"] - flat.grad).norm() / flat.grad.norm()).item()
    assert rel < 1e-5, rel


def test_flat_params_views_and_shared_parameters():
    from graphcast_lite_amd.train import FlatParams

    m, _, _ = _make()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(before[k], v), k  # values preserved
    p = m.processor.graph_layer.activation.weight
    assert p.data_ptr() >= flat.flat.data_ptr() and p.grad is not None
    flat.flat.mul_(2.0)  # the module sees bucket updates (what the fused Adam relies on)
    assert torch.allclose(p.detach(), before["processor.graph_layer.activation.weight"] * 2)
    assert m.processor.graph_layer.layers[1].weight is p  # alias of the shared PReLU stays an alias


def test_shard_batch_rejects_uneven_split():
    from graphcast_lite_amd.train import shard_batch

    X, y = torch.zeros(5, 3, 2), torch.zeros(5, 3, 1)
    with pytest.raises(ValueError):
        shard_batch(X, y, 0, 2)
    a, b = shard_batch(torch.arange(8).view(8, 1, 1), torch.arange(8).view(8, 1, 1), 1, 4)
    assert a.flatten().tolist() == [2, 3]


def _bcast_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    from graphcast_lite_amd.models import _broadcast_edges_from_rank0

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank pruned differently (different samples); all must end with rank 0's list
    mine = torch.arange(2 * (5 + 3 * rank), dtype=torch.int64).view(2, -1) + 100 * rank
    got = _broadcast_edges_from_rank0(mine)
    torch.save(got, os.path.join(out_dir, f"edges_{rank}.pt"))
    dist.destroy_process_group()


def test_sparse_gat_prune_broadcast_c2(tmp_path):
    """C2: the pruned processing graph of rank 0 reaches every rank (different list lengths)."""
    from graphcast_lite_amd.models import _broadcast_edges_from_rank0

    port = _free_port()
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    e0, e1 = torch.load(os.path.join(tmp_path, "edges_0.pt")), torch.load(os.path.join(tmp_path, "edges_1.pt"))
    assert torch.equal(e0, torch.arange(10, dtype=torch.int64).view(2, 5)) and torch.equal(e0, e1)
    x = torch.zeros(2, 3, dtype=torch.int64)
    assert _broadcast_edges_from_rank0(x) is x  # single process: untouched


# ------------------------------------------------------------------------------------------------
# bench.py's own launcher (CPU, gloo): `python bench.py --gpus N` without torchrun must start N real
# ranks - or fail - and report the world size the process group has, never args.gpus.
# ------------------------------------------------------------------------------------------------
def _run_bench(args, env_extra, timeout=300):
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_launcher_spawns_real_ranks():
    r, lines = _run_bench(["--gpus", "2"], {"GCL_BENCH_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1  # ONE line, from rank 0
    assert lines[0] == {"selftest": True, "n_gpus": 2, "rank_sum": 3.0, "spawned": True}
    r, lines = _run_bench(["--gpus", "3"], {"GCL_BENCH_SELFTEST": "1"})
    assert r.returncode == 0 and lines[0]["n_gpus"] == 3 and lines[0]["rank_sum"] == 6.0


def test_bench_never_reports_more_gpus_than_ranks():
    """Without a launcher and without enough devices the run FAILS (no JSON line); a launcher that started
    a different number of ranks than --gpus says is refused as well."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r, lines = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and not lines
    assert "refusing" in r.stderr
    r, lines = _run_bench(["--gpus", "2"], {"GCL_BENCH_SELFTEST": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and not lines and "WORLD_SIZE=1" in r.stderr
    # a failing rank takes the whole launch down with a non-zero code
    r, lines = _run_bench(["--gpus", "2"], {"GCL_BENCH_SELFTEST": "1", "GCL_DIST_BACKEND": "no_such_backend"})
    assert r.returncode != 0 and not lines


# ------------------------------------------------------------------------------------------------
# TrainStep with world_size 2 on the GPU box: two gloo ranks sharing device 0 (tests/helpers/dp_rank.py)
# against the single-process step on the global batch.
# ------------------------------------------------------------------------------------------------
def _spawn_dp(tmp_path, steps, seed_mode, use_graph):
    import subprocess
    import sys

    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "dp_rank.py"), str(tmp_path),
                                       str(steps), seed_mode, "1" if use_graph else "0"], env=env))
    codes = [p.wait(timeout=900) for p in procs]
    assert codes == [0, 0], codes
    return torch.load(os.path.join(tmp_path, "dp_rank0.pt"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed_mode,use_graph", [("same", False), ("different", True)])
def test_two_rank_train_step_equals_global_batch_step(tmp_path, seed_mode, use_graph):
    """TrainStep(world_size=2): batch shards, ONE all-reduce of the flat gradient bucket, 1/world folded into
    Adam - after 4 optimiser steps the replicas equal the single-process TrainStep on the global batch.
    With `different` seeds per rank only TrainStep's broadcast of rank 0's state makes that true; with
    `use_graph` the forward+backward is replayed from a hipGraph and the collective + Adam follow eagerly."""
    sys_path = os.path.join(ROOT, "tests", "helpers")
    import sys

    sys.path.insert(0, sys_path)
    import dp_rank
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    got = _spawn_dp(tmp_path, 4, seed_mode, use_graph)
    cfg, m = dp_rank.build(42)  # rank 0's seed
    step = TrainStep(m, lr=1e-3, lat_weights=get_lat_weights(32, 64, "cuda:0"), use_graph=False)
    X, y = dp_rank.global_batch(m._num_grid_nodes)
    Xd, yd = X.cuda(), y.cuda()
    for i in range(4):
        step(Xd * (1 + 0.01 * i), yd)
    assert got["t"] == step.opt.t == 4
    for k, v in m.named_parameters():
        a, b = got["params"][k].double(), v.detach().cpu().double()
        # 4 Adam steps: the normalised update g / sqrt(v) amplifies the fp32 summation-order difference between
        # "2 ranks x 2 samples, then all-reduce" and "4 samples" (same bar as the torch-Adam comparison)
        assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-9, k
    if use_graph:
        assert got["launch_mode"].startswith("hipGraph replay")


@pytest.mark.gpu
def test_train_step_raises_when_requested_graph_cannot_be_captured(monkeypatch):
    """use_graph=True is a requirement (RuntimeError on a failed capture); use_graph=None degrades to eager
    launches with a RuntimeWarning and says so in .launch_mode."""
    import sys
    import warnings

    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    import dp_rank
    from graphcast_lite_amd import train as TR

    cfg, m = dp_rank.build(42)
    X, y = dp_rank.global_batch(m._num_grid_nodes, 2)
    Xd, yd = X.cuda(), y.cuda()

    def boom(self, X, y):
        raise RuntimeError("capture refused (test)")

    monkeypatch.setattr(TR.TrainStep, "_capture", boom)
    s1 = TR.TrainStep(m, use_graph=True)
    s1(Xd, yd), s1(Xd, yd)
    with pytest.raises(RuntimeError, match="capture failed"):
        s1(Xd, yd)
    cfg, m2 = dp_rank.build(42)
    s2 = TR.TrainStep(m2, use_graph=None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(4):
            s2(Xd, yd)
    assert any("capture unavailable" in str(x.message) for x in w)
    assert not s2.graph_active and s2.launch_mode.startswith("eager (capture failed")
