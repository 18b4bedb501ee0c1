"""Device-resident input pipeline: the reference's on-the-fly window loader, done on the GPU.

Mirrors `TimeseriesChunkDataset` of `src/data/dataloader_chunked.py:33-223`: the raw fp16 time series
(`data.npy` + `dataset_info.json`, or `chunk_*.npy`) is uploaded ONCE and stays in HBM as fp16 (the
81 GB of `wb2_512x256_19f_ar` fit beside the model in 288 GB); a batch of windows is then one
`gcl_window_pack` launch (dequantise, z-score, (lat, lon)-major transpose) instead of a host-side
numpy pass per sample plus an H2D copy of fp32 windows.  Sample indexing and the train / val / test
splits follow the reference (`:132-172`), windows never cross a chunk boundary.
"""
import glob
import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip


def _upload_fp16(mm: np.ndarray, device, slab_bytes: int = 1 << 30) -> torch.Tensor:
    """Stream a (memory-mapped) fp16 array to the GPU in slabs, so the host never holds it whole."""
    if mm.dtype != np.float16:
        raise TypeError(f"time series must be float16 on disk, got {mm.dtype}")
    out = torch.empty(mm.shape, dtype=torch.float16, device=device)
    per_t = max(1, int(np.prod(mm.shape[1:])) * 2)
    step = max(1, slab_bytes // per_t)
    for t in range(0, mm.shape[0], step):
        out[t:t + step].copy_(torch.from_numpy(np.array(mm[t:t + step])), non_blocking=False)
    return out


class TimeseriesChunkDataset:
    """`src/data/dataloader_chunked.py:33-223` with the series resident on `device`.

    `ds[i]` returns `(X [G, obs*C], Y [G, pred*C])` like the reference's `__getitem__` (on the GPU);
    `ds.batch(indices)` returns `[B, G, ...]` tensors from one kernel launch per chunk touched."""

    def __init__(self, data_dir: str, obs_window: int = 2, pred_steps: int = 1, split: str = "train",
                 n_features: Optional[int] = None, test_fraction: float = 0.2, device="cuda:0"):
        self.data_dir, self.obs_window, self.pred_steps = data_dir, obs_window, pred_steps
        self.split, self.test_fraction, self.device = split, test_fraction, torch.device(device)
        scalers = np.load(os.path.join(data_dir, "scalers.npz"))
        mean, std = scalers["mean"].astype(np.float32), scalers["std"].astype(np.float32)

        single, info_file = os.path.join(data_dir, "data.npy"), os.path.join(data_dir, "dataset_info.json")
        if os.path.exists(single) and os.path.exists(info_file):  # raw memmap without an .npy header (:79-96)
            with open(info_file) as fh:
                info = json.load(fh)
            self.flat_grid = bool(info.get("flat", False))
            shape = ((info["n_time"], info["n_nodes"], info["n_feat"]) if self.flat_grid
                     else (info["n_time"], info["n_lon"], info["n_lat"], info["n_feat"]))
            host_chunks = [np.memmap(single, dtype=np.float16, mode="r", shape=shape)]
        else:
            self.flat_grid = False
            files = sorted(glob.glob(os.path.join(data_dir, "chunk_*.npy")))
            if not files:
                raise FileNotFoundError(f"No data.npy or chunk_*.npy found in {data_dir}")
            host_chunks = [np.load(f, mmap_mode="r") for f in files]
        self.chunk_lengths = [int(c.shape[0]) for c in host_chunks]
        self.total_time = sum(self.chunk_lengths)
        c0 = host_chunks[0]
        if self.flat_grid:
            self.n_nodes, self.n_lon, self.n_lat, self.n_feat_total = int(c0.shape[1]), None, None, int(c0.shape[2])
        else:
            self.n_nodes, self.n_lon, self.n_lat, self.n_feat_total = None, int(c0.shape[1]), int(c0.shape[2]), int(c0.shape[3])
        self.n_feat = n_features if n_features else self.n_feat_total
        self.mean_np, self.std_np = mean[:self.n_feat], std[:self.n_feat]
        self.mean = torch.from_numpy(self.mean_np.copy()).to(self.device)
        self.std = torch.from_numpy(self.std_np.copy()).to(self.device)
        self.chunks = [_upload_fp16(c, self.device) for c in host_chunks]

        self._sample_indices = sample_indices(self.chunk_lengths, obs_window, pred_steps, split, test_fraction)
        where = f"flat_nodes={self.n_nodes}" if self.flat_grid else f"grid={self.n_lon}×{self.n_lat}"
        print(f"[ChunkDataset] {split}: {len(self._sample_indices)} samples, {where}, feat={self.n_feat}, "
              f"obs={obs_window}, pred={pred_steps}")

    def __len__(self):
        return len(self._sample_indices)

    @property
    def grid_nodes(self) -> int:
        return self.n_nodes if self.flat_grid else self.n_lon * self.n_lat

    def batch(self, indices: Sequence[int], out=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """A whole batch of windows in one launch per chunk.  out = (X, Y): fill these buffers in place (e.g.
        `TrainStep.input_buffers()`, so that the captured step needs no copy of the batch)."""
        pairs = [self._sample_indices[int(i)] for i in indices]
        B, G = len(pairs), self.grid_nodes
        by_chunk = {}
        for pos, (ci, t) in enumerate(pairs):
            by_chunk.setdefault(ci, []).append((pos, t))
        X, Y = out if out is not None else (None, None)
        for ci, items in by_chunk.items():
            t0 = torch.tensor([t for _, t in items], dtype=torch.int64).to(self.device)
            if len(by_chunk) == 1:
                return hip.window_pack(self.chunks[ci], t0, self.mean, self.std, self.n_feat, self.obs_window,
                                       self.pred_steps, out=out)
            x, y = hip.window_pack(self.chunks[ci], t0, self.mean, self.std, self.n_feat, self.obs_window, self.pred_steps)
            if X is None:
                X = torch.empty(B, G, x.shape[-1], dtype=torch.float32, device=self.device)
                Y = torch.empty(B, G, y.shape[-1], dtype=torch.float32, device=self.device)
            pos = torch.tensor([p for p, _ in items], dtype=torch.int64, device=self.device)
            X.index_copy_(0, pos, x)
            Y.index_copy_(0, pos, y)
        return X, Y

    def __getitem__(self, idx):
        X, Y = self.batch([idx])
        return X[0], Y[0]


def sample_indices(chunk_lengths: Sequence[int], obs_window: int, pred_steps: int, split: str,
                   test_fraction: float) -> List[Tuple[int, int]]:
    """(chunk, local_t) of every window that fits inside one chunk, then the split by time
    (`src/data/dataloader_chunked.py:132-172`): train = first (1 - test_fraction), test = the rest,
    val / test_only = first / second half of the test part, all = everything."""
    window = obs_window + pred_steps
    idx = [(ci, t) for ci, T in enumerate(chunk_lengths) for t in range(max(0, T - window + 1))]
    cut = int(len(idx) * (1 - test_fraction))
    if split == "train":
        return idx[:cut]
    if split == "test":
        return idx[cut:]
    if split in ("val", "test_only"):
        test = idx[cut:]
        half = len(test) // 2
        return test[:half] if split == "val" else test[half:]
    if split == "all":
        return idx
    raise ValueError(f"Unknown split: {split}")
