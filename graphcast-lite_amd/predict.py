"""Autoregressive rollout / inference caller of the hot path.

Mirrors the AR branch of the reference's inference script (`scripts/predict.py:499-538`; the same
steps appear in the training loop, `src/train.py:203-228`): the one-step model predicts a delta,
the residual is added, static channels are carried forward from the last input step, forcing
channels are taken from the known future (`y`), the step is stored and the observation window is
shifted.  Here the whole per-step glue is one kernel (`gcl_ar_advance`) and the window never
leaves the device.
"""
from typing import Optional, Sequence

import torch

from . import hip


def channel_kinds(C: int, static_channels: Optional[Sequence[int]], forcing_channels: Optional[Sequence[int]], device):
    """int32 [C]: 0 predicted, 1 static (carry forward), 2 forcing (from y).  Forcing wins over
    static when a channel is listed twice, as the reference applies the forcing overwrite last."""
    kind = torch.zeros(C, dtype=torch.int32)
    for ch in static_channels or []:
        kind[ch] = 1
    for ch in forcing_channels or []:
        kind[ch] = 2
    return kind.to(device)


@torch.no_grad()
def rollout(model, X: torch.Tensor, ar_steps: int, y: Optional[torch.Tensor] = None, static_channels=None,
            forcing_channels=None, use_residual: bool = True, attention_threshold: float = 0.0) -> torch.Tensor:
    """X [B,G,obs*C] (or [G,obs*C]) on the GPU -> predictions [B,G,ar_steps*C].

    `y` [B,G,P*C] supplies the forcing channels for the steps it covers (`ar_step < P`), exactly as
    `scripts/predict.py:527-529` does; later steps keep the model's own value."""
    squeeze = X.dim() == 2
    if squeeze:
        X = X.unsqueeze(0)
        y = y.unsqueeze(0) if y is not None else None
    B, G, _ = X.shape
    obs = model.obs_window
    C = X.shape[-1] // obs
    state = X.reshape(B, G, obs, C).contiguous()
    kinds = channel_kinds(C, static_channels, forcing_channels, X.device)
    out = torch.empty(B, G, ar_steps * C, dtype=torch.float32, device=X.device)
    y_steps = y.shape[-1] // C if y is not None else 0
    for s in range(ar_steps):
        delta = model(X=state.view(B, G, obs * C), attention_threshold=attention_threshold)
        if delta.dim() == 2:
            delta = delta.unsqueeze(0)
        y_step = y[:, :, s * C:(s + 1) * C] if (y is not None and forcing_channels and s < y_steps) else None
        state = hip.ar_advance(state, delta, y_step, kinds, out, s * C, use_residual)
    return out[0] if squeeze else out
