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
            forcing_channels=None, use_residual: bool = True, attention_threshold: float = 0.0,
            kinds: Optional[torch.Tensor] = None) -> torch.Tensor:
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
    if kinds is None:  # (a caller that captures the rollout uploads this once, outside the capture)
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


class CapturedRollout:
    """`rollout` replayed from a hipGraph: the whole K-step autoregressive forecast (K model forwards
    + K window advances, a few hundred launches at batch 1) costs the host one graph launch.

    The first two calls for a given input signature run eagerly (graph handles, workspaces and
    allocator pools get set up), the third captures, later calls copy the inputs into the captured
    buffers and replay.  Falls back to eager launches if capture is not possible."""

    def __init__(self, model, ar_steps: int, static_channels=None, forcing_channels=None, use_residual: bool = True):
        self.model, self.ar_steps = model, ar_steps
        self.static_channels, self.forcing_channels, self.use_residual = static_channels, forcing_channels, use_residual
        self._sig, self._graph, self._sX, self._sy, self._out, self._calls, self.enabled = None, None, None, None, None, 0, True
        self._kinds = None

    def _eager(self, X, y):
        C = X.shape[-1] // self.model.obs_window
        if self._kinds is None or self._kinds.numel() != C or self._kinds.device != X.device:
            self._kinds = channel_kinds(C, self.static_channels, self.forcing_channels, X.device)
        return rollout(self.model, X, self.ar_steps, y=y, static_channels=self.static_channels,
                       forcing_channels=self.forcing_channels, use_residual=self.use_residual, kinds=self._kinds)

    @torch.no_grad()
    def __call__(self, X: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self.enabled or getattr(self.model, "using_sparse_gat", False):
            return self._eager(X, y)
        sig = (tuple(X.shape), None if y is None else tuple(y.shape))
        if sig != self._sig:
            self._sig, self._graph, self._calls = sig, None, 0
        if self._graph is None:
            if self._calls < 2:
                self._calls += 1
                return self._eager(X, y)
            try:
                self._sX = X.clone()
                self._sy = y.clone() if y is not None else None
                from . import models as _models

                g = torch.cuda.CUDAGraph()
                _models._graphs.pin = pinned = []  # keep the CSR handles of the captured kernels alive
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._out = self._eager(self._sX, self._sy)
                finally:
                    _models._graphs.pin = None
                self._graph, self._pinned = g, pinned
            except Exception as e:  # capture is an optimisation, never a requirement
                print(f"[CapturedRollout] hipGraph capture unavailable ({type(e).__name__}: {str(e)[:200]}); staying eager",
                      flush=True)
                self.enabled, self._graph = False, None
                torch.cuda.synchronize()
                return self._eager(X, y)
        self._sX.copy_(X)
        if y is not None:
            self._sy.copy_(y)
        self._graph.replay()
        return self._out.clone()
