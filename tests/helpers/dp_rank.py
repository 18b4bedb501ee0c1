"""One data-parallel rank of `tests/test_distributed.py::test_two_rank_train_step_*` (GPU box only).

Started as a subprocess with RANK / WORLD_SIZE / MASTER_* set.  All ranks share device 0 through the
gloo backend (RCCL refuses two ranks on one device; the driver's multi-GPU run uses RCCL with one
device per rank - the code path through TrainStep is the same).  argv: out_dir steps seed_mode use_graph
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(seed):
    from conftest import experiment
    from graphcast_lite_amd.models import WeatherPrediction

    cfg = experiment("baseline", mesh_levels=[1, 2])
    torch.manual_seed(seed)
    lats, lons = np.linspace(-90, 90, 32), np.linspace(0, 360, 64, endpoint=False)
    return cfg, WeatherPrediction((lats, lons), cfg.graph, cfg.pipeline, cfg.data, torch.device("cuda:0"))


def global_batch(G, B=4):
    g = torch.Generator().manual_seed(1234)
    X = torch.randn(B, G, 66, generator=g)
    y = X[..., 33:] + 0.1 * torch.randn(B, G, 33, generator=g)
    return X, y


def main():
    out_dir, steps, seed_mode, use_graph = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphcast_lite_amd.train import TrainStep, get_lat_weights, shard_batch

    # "same": every rank builds the same weights; "different": only the broadcast in TrainStep makes them equal
    cfg, m = build(42 if seed_mode == "same" else 42 + 7 * rank)
    step = TrainStep(m, lr=1e-3, lat_weights=get_lat_weights(32, 64, "cuda:0"), world_size=world,
                     use_graph=True if use_graph else False)
    X, y = global_batch(m._num_grid_nodes)
    Xl, yl = shard_batch(X, y, rank, world)
    Xl, yl = Xl.cuda(), yl.cuda()
    losses = []
    for i in range(steps):
        losses.append(float(step(Xl * (1 + 0.01 * i), yl)))
    torch.cuda.synchronize()
    flat = step.flat.flat.detach().cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged"
    if use_graph:
        assert step.graph_active and "all-reduce + Adam eager" in step.launch_mode, step.launch_mode
    if rank == 0:
        torch.save({"params": {k: v.detach().cpu() for k, v in m.named_parameters()}, "losses": losses,
                    "launch_mode": step.launch_mode, "t": step.opt.t}, os.path.join(out_dir, "dp_rank0.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
