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
