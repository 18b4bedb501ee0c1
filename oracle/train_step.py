"""CPU restatement of the caller semantics of the training step (`src/train.py`).

TEST INFRASTRUCTURE ONLY.  Pinned by `tests/golden/loss_vectors.npz`, which was produced by
running the reference's own `weighted_mse_loss`, `get_lat_weights`, `build_boundary_mask` and
`update_attention_threshold` in the build container.
"""
import torch


def get_lat_weights(lat_dim: int, lon_dim: int):
    """`src/train.py:53-72`: cos(lat)/mean, tiled `[lon, lat]` then flattened (lon-major, a
    reference quirk kept on purpose - SURVEY.md Appendix B.3).  Pole weights are ~-7e-8."""
    w = torch.cos(torch.deg2rad(torch.linspace(-90, 90, lat_dim)))
    w = w / w.mean()
    return w.view(1, -1).expand(lon_dim, lat_dim).reshape(1, -1, 1)


def build_boundary_mask(n_lon: int, n_lat: int, width: int):
    """`src/train.py:74-82`."""
    m = torch.zeros(n_lon, n_lat)
    m[width:n_lon - width, width:n_lat - width] = 1.0
    return m.reshape(1, -1, 1)


def weighted_mse_loss(pred, target, lat_weights=None, channel_mask=None, spatial_mask=None):
    """`src/train.py:85-102`: sum(w * (pred-target)^2) / max(sum(w), 1e-12), w broadcast to
    `[B, G, C]` from lat `[1,G,1]`, channel `[C]`, spatial `[1,G,1]`."""
    diff = (pred - target) ** 2
    w = torch.ones_like(diff)
    if channel_mask is not None:
        w = w * channel_mask.view(1, 1, -1)
    if spatial_mask is not None:
        w = w * spatial_mask
    if lat_weights is not None:
        w = w * lat_weights
    return (diff * w).sum() / w.sum().clamp_min(1e-12)


def update_attention_threshold(epoch, max_epochs=30, start_epoch=5, final_threshold=0.1356):
    """`src/train.py:132-136`."""
    if epoch < start_epoch:
        return 0.0
    if epoch > max_epochs + start_epoch:
        return final_threshold
    return min(final_threshold, (epoch - start_epoch) * final_threshold / (max_epochs - start_epoch))


def train_step_loss(model, X, y, *, lat_weights=None, channel_mask=None, spatial_mask=None,
                    ar_steps=1, static_channels=None, forcing_channels=None, use_residual=True,
                    threshold=0.0, epoch=0, batch_num=1):
    """Loss of one batch exactly as the inner loop of `train_epoch` builds it
    (`src/train.py:173-231`): AR rollout, residual add, per-step weighted MSE, mean over steps."""
    N, G, _ = X.shape
    obs = model.obs_window
    C = X.shape[-1] // obs
    steps_total = y.shape[-1] // C
    y_steps = y.view(N, G, steps_total, C)
    state = X.view(N, G, obs, C)
    steps = min(ar_steps, steps_total)
    loss = 0
    for s in range(steps):
        delta = model(X=state.reshape(N, G, -1), attention_threshold=threshold, epoch=epoch, batch_num=batch_num)
        if delta.dim() == 2:
            delta = delta.unsqueeze(0)
        out = state[:, :, -1, :] + delta if use_residual else delta
        loss = loss + weighted_mse_loss(out, y_steps[:, :, s, :], lat_weights, channel_mask, spatial_mask)
        if static_channels:
            for ch in static_channels:
                out[:, :, ch] = state[:, :, -1, ch]
        if forcing_channels:
            for ch in forcing_channels:
                out[:, :, ch] = y_steps[:, :, s, ch]
        state = torch.cat([state[:, :, 1:, :], out.unsqueeze(2)], dim=2)
    return loss / steps


@torch.no_grad()
def ar_rollout(model, X, ar_steps, y=None, static_channels=None, forcing_channels=None, use_residual=True):
    """AR branch of the reference's inference loop (`scripts/predict.py:499-538`), batched."""
    B, G, _ = X.shape
    obs = model.obs_window
    C = X.shape[-1] // obs
    state = X.view(B, G, obs, C)
    y_steps = y.view(B, G, -1, C) if y is not None else None
    outs = []
    for s in range(ar_steps):
        delta = model(X=state.reshape(B, G, -1), attention_threshold=0.0)
        if delta.dim() == 2:
            delta = delta.unsqueeze(0)
        step_out = (state[:, :, -1, :] + delta) if use_residual else delta.clone()
        if static_channels:
            for ch in static_channels:
                step_out[:, :, ch] = state[:, :, -1, ch]
        if forcing_channels and y_steps is not None and s < y_steps.shape[2]:
            for ch in forcing_channels:
                step_out[:, :, ch] = y_steps[:, :, s, ch]
        outs.append(step_out)
        state = torch.cat([state[:, :, 1:, :], step_out.unsqueeze(2)], dim=2)
    return torch.cat(outs, dim=-1)


def spatial_corr(pred, true, exclude_channels=None) -> float:
    """`src/train.py:114-130`."""
    if pred.dim() == 3:
        accs = [spatial_corr(pred[b], true[b], exclude_channels) for b in range(pred.shape[0])]
        return sum(accs) / max(len(accs), 1)
    p = (pred - pred.mean(dim=0, keepdim=True)) / (pred.std(dim=0, keepdim=True) + 1e-8)
    t = (true - true.mean(dim=0, keepdim=True)) / (true.std(dim=0, keepdim=True) + 1e-8)
    acc = (p * t).mean(dim=0)
    if exclude_channels:
        keep = [i for i in range(acc.shape[0]) if i not in exclude_channels]
        if keep:
            return acc[keep].mean().item()
    return acc.mean().item()


@torch.no_grad()
def evaluate(model, batches, lat_weights=None, spatial_mask=None, channel_mask=None, static_channels=None,
             forcing_channels=None, use_residual=True):
    """The reference's `test` loop (`src/train.py:241-308`) over a list of (X, y) batches."""
    total, accs, raw = 0.0, [], []
    for X, y in batches:
        C = X.shape[-1] // model.obs_window
        steps = y.shape[-1] // C
        y0 = y.view(y.shape[0], y.shape[1], steps, C)[:, :, 0, :] if steps > 1 else y
        pred = model(X=X, attention_threshold=0.0)
        if pred.dim() == 2:
            pred = pred.unsqueeze(0)
        x_last = X.view(X.shape[0], X.shape[1], model.obs_window, C)[:, :, -1, :]
        outs = x_last + pred if use_residual else pred.clone()
        for ch in static_channels or []:
            outs[:, :, ch] = x_last[:, :, ch]
        for ch in forcing_channels or []:
            outs[:, :, ch] = y0[:, :, ch]
        total += weighted_mse_loss(outs, y0, lat_weights, channel_mask, spatial_mask).item()
        raw.append(((outs - y0) ** 2).mean().item())
        skip = sorted(set(static_channels or []) | set(forcing_channels or []))
        accs.append(spatial_corr(outs, y0, exclude_channels=skip if skip else None))
    n = max(len(raw), 1)
    return total / max(len(batches), 1), sum(accs) / n, (sum(raw) / n) ** 0.5
