"""CPU restatement (numpy) of the reference's window loader - TEST INFRASTRUCTURE ONLY.

`window_sample` follows `TimeseriesChunkDataset.__getitem__` (`src/data/dataloader_chunked.py:179-223`)
step by step: slice the window, keep the first n_feat channels, cast fp16 -> fp32, z-score with the
fp32 scalers, split into observed / target frames and flatten (lat, lon)-major.  It is pinned by a
hand-computed case in `tests/test_oracle.py` (the reference ships no dataset or loader test)."""
import numpy as np


def window_sample(chunk: np.ndarray, local_t: int, obs: int, pred: int, n_feat: int, mean: np.ndarray,
                  std: np.ndarray, flat: bool):
    window = chunk[local_t: local_t + obs + pred]
    if flat:  # (frames, N, C) -> (N, frames*C)
        w = (window[:, :, :n_feat].astype(np.float32) - mean) / std
        n = w.shape[1]
        return (w[:obs].transpose(1, 0, 2).reshape(n, obs * n_feat),
                w[obs:].transpose(1, 0, 2).reshape(n, pred * n_feat))
    w = (window[:, :, :, :n_feat].astype(np.float32) - mean) / std  # (frames, lon, lat, C)
    g = w.shape[1] * w.shape[2]
    return (w[:obs].transpose(2, 1, 0, 3).reshape(g, obs * n_feat),     # lat slow, lon fast
            w[obs:].transpose(2, 1, 0, 3).reshape(g, pred * n_feat))
