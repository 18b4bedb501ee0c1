"""cProfile of a few eager training steps (host-side cost per step).  python tools/prof_step.py <config> [batch]"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from graphcast_lite_amd.train import TrainStep, get_lat_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "baseline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
cfg, model, grid = bench.build_model(name, dev)
X, y = bench.synthetic_batch(cfg, model._num_grid_nodes, B, seed=1)
X, y = X.to(dev), y.to(dev)
step = TrainStep(model, lr=1e-3, lat_weights=get_lat_weights(grid[0], grid[1], dev), use_graph=False)
for _ in range(3):
    step(X, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(X, y)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
