"""Training-step semantics of the reference's `src/train.py`, on the HIP path.

`weighted_mse_loss`, `get_lat_weights`, `build_boundary_mask`, `update_attention_threshold` and
`train_epoch` keep the reference's names, arguments and results (src/train.py:53-136,138-239).
`TrainStep` is the data-parallel fast path used by `bench.py`: one flat parameter / gradient /
Adam-state buffer, gradients accumulated in place by the backward kernels, ONE RCCL all-reduce of
the flat gradient bucket per optimiser step, one fused Adam launch.
"""
import os

import torch
import torch.distributed as dist

from . import hip
from .functional import ARStepLossFn, WeightedMSEFn


class FileNames:
    """File names of an experiment directory (src/constants.py:5-15)."""

    EXPERIMENT_CONFIG = "config.json"
    TRAIN_X = "X_train.pt"
    TRAIN_Y = "y_train.pt"
    TEST_X = "X_test.pt"
    TEST_Y = "y_test.pt"
    SAVED_MODEL = "best_model.pth"
    SAVED_RESULTS = "results.json"
    CHECKPOINT = "checkpoint.pth"


def save_checkpoint(path, model, optimiser, epoch, ar_steps, best_val_loss, patience_counter, train_losses,
                    val_losses):
    """Same dictionary layout as the reference (src/train.py:22-34), so either side can resume the
    other's `checkpoint.pth`; `best_model.pth` is a bare `model.state_dict()` (src/train.py:496)."""
    torch.save({
        "epoch": epoch,
        "ar_steps": ar_steps,
        "best_val_loss": best_val_loss,
        "patience_counter": patience_counter,
        "train_losses": train_losses,
        "val_losses": val_losses,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimiser.state_dict(),
    }, path)


def load_checkpoint(path, model, optimiser, device):
    """src/train.py:37-49.  `weights_only=True`: nothing from the file is executed."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    optimiser.load_state_dict(ckpt["optimizer_state_dict"])
    return {
        "start_epoch": ckpt["epoch"] + 1,
        "ar_steps": ckpt["ar_steps"],
        "best_val_loss": ckpt["best_val_loss"],
        "patience_counter": ckpt["patience_counter"],
        "train_losses": ckpt["train_losses"],
        "val_losses": ckpt["val_losses"],
    }


def get_lat_weights(lat_dim, lon_dim, device, flat_lats=None):
    """cos(lat)/mean laid out as the reference lays it out (src/train.py:53-72), `[1, G, 1]`."""
    if flat_lats is not None:
        w = torch.cos(torch.deg2rad(torch.from_numpy(flat_lats.copy()).float()))
        w = w / w.mean()
        return w.view(1, -1, 1).to(device)
    w = torch.cos(torch.deg2rad(torch.linspace(-90, 90, lat_dim)))
    w = w / w.mean()
    return w.view(1, -1).expand(lon_dim, lat_dim).reshape(-1).view(1, -1, 1).to(device)


def build_boundary_mask(n_lon, n_lat, width, device):
    """src/train.py:74-82."""
    m = torch.zeros(n_lon, n_lat)
    m[width:n_lon - width, width:n_lat - width] = 1.0
    return m.reshape(1, -1, 1).to(device)


def combine_spatial_masks(*masks):
    out = None
    for m in masks:
        if m is not None:
            out = m if out is None else out * m
    return out


_lw_cache = {}


def _loss_weights(G: int, C: int, lat_weights, channel_mask, spatial_mask, device):
    """Per-node and per-channel factors of the reference's broadcast weight tensor and sum(w) for
    ONE sample (the caller multiplies by the batch size).  Cached per mask set: building it needs a
    host read-back, which must not happen every step (nor inside a hipGraph capture)."""
    key = (G, C, str(device)) + tuple((id(t), t._version) if t is not None else None
                                      for t in (lat_weights, channel_mask, spatial_mask))
    hit = _lw_cache.get(key)
    if hit is not None:
        return hit[0]
    node_w = None
    for t in (spatial_mask, lat_weights):
        if t is not None:
            t = t.reshape(-1).to(device=device, dtype=torch.float32)
            node_w = t if node_w is None else node_w * t
    chan_w = channel_mask.reshape(-1).to(device=device, dtype=torch.float32) if channel_mask is not None else None
    s_node = node_w.double().sum().item() if node_w is not None else float(G)
    s_chan = chan_w.double().sum().item() if chan_w is not None else float(C)
    out = (node_w.contiguous() if node_w is not None else None,
           chan_w.contiguous() if chan_w is not None else None, s_node * s_chan)
    if len(_lw_cache) > 64:
        _lw_cache.clear()
    _lw_cache[key] = (out, lat_weights, channel_mask, spatial_mask)  # keep the keys' tensors alive
    return out


def weighted_mse_loss(pred, target, lat_weights=None, channel_mask=None, spatial_mask=None, x_last=None):
    """src/train.py:85-102 as one kernel: sum(w (pred-target)^2) / max(sum(w), 1e-12).
    `x_last` (extension) folds the residual add `pred = x_last + delta` into the same pass."""
    if pred.dim() == 2:
        pred, target = pred.unsqueeze(0), target.unsqueeze(0)
    B, G, C = pred.shape
    node_w, chan_w, wsum1 = _loss_weights(G, C, lat_weights, channel_mask, spatial_mask, pred.device)
    inv = 1.0 / max(wsum1 * B, 1e-12)
    return WeightedMSEFn.apply(pred, x_last, target, node_w, chan_w, inv)


def update_attention_threshold(epoch, max_epochs=30, start_epoch=5, final_threshold=0.1356):
    """src/train.py:132-136."""
    if epoch < start_epoch:
        return 0.0
    if epoch > max_epochs + start_epoch:
        return final_threshold
    return min(final_threshold, (epoch - start_epoch) * final_threshold / (max_epochs - start_epoch))


_kind_cache = {}


def _channel_kinds(C, static_channels, forcing_channels, device):
    """Device int32 [C] (0 predicted, 1 static, 2 forcing), cached: uploading it needs a host -> device copy that
    must not happen on every step (nor inside a hipGraph capture)."""
    from .predict import channel_kinds

    key = (C, tuple(static_channels or ()), tuple(forcing_channels or ()), str(device))
    k = _kind_cache.get(key)
    if k is None:
        k = _kind_cache[key] = channel_kinds(C, static_channels, forcing_channels, device)
    return k


def batch_loss(model, X, y, threshold=0.0, epoch=0, batch_num=1, lat_weights=None, current_ar_steps=1,
               channel_mask=None, spatial_mask=None, static_channels=None, forcing_channels=None,
               use_residual=True):
    """Loss of one batch as the reference's inner loop builds it (src/train.py:173-231): per AR step the model
    predicts a delta, the (residual) prediction is scored against that step's target, static / forcing channels are
    overwritten and the window shifts; the step losses are averaged.  Each step is one `ARStepLossFn` (loss + window
    advance fused, 1/steps folded into the loss normaliser, the running sum kept on the device)."""
    N, G, _ = X.shape
    obs = model.obs_window
    C = X.shape[-1] // obs
    steps_total = y.shape[-1] // C
    y_steps = y.view(N, G, steps_total, C)
    state = X.view(N, G, obs, C)
    steps = min(current_ar_steps, steps_total)
    node_w, chan_w, wsum1 = _loss_weights(G, C, lat_weights, channel_mask, spatial_mask, X.device)
    inv = 1.0 / (max(wsum1 * N, 1e-12) * steps)
    kinds = _channel_kinds(C, static_channels, forcing_channels, X.device) if steps > 1 else None
    loss = None
    for s in range(steps):
        delta = model(X=state.reshape(N, G, -1), attention_threshold=threshold, epoch=epoch, batch_num=batch_num)
        if delta.dim() == 2:
            delta = delta.unsqueeze(0)
        loss, state = ARStepLossFn.apply(state, delta, y_steps[:, :, s, :], loss, node_w, chan_w, inv, kinds,
                                         bool(use_residual), s + 1 < steps)
    return loss


def train_epoch(model, train_dataloader, optimiser, loss_fn, device, threshold, epoch, lat_weights=None,
                current_ar_steps=1, channel_mask=None, spatial_mask=None, static_channels=None,
                forcing_channels=None, use_residual=True):
    """src/train.py:138-239 (same signature; `loss_fn` is unused there as well)."""
    model.train()
    total = 0.0
    for i, (X, y) in enumerate(train_dataloader):
        y = y.squeeze(0) if y.dim() == 4 else y
        X, y = X.to(device), y.to(device)
        optimiser.zero_grad()
        loss = batch_loss(model, X, y, threshold, epoch, i, lat_weights, current_ar_steps, channel_mask,
                          spatial_mask, static_channels, forcing_channels, use_residual)
        loss.backward()
        optimiser.step()
        total += loss.detach().item()
    return total / max(len(train_dataloader), 1)


def spatial_corr(pred: torch.Tensor, true: torch.Tensor, exclude_channels=None) -> float:
    """Spatial anomaly correlation (src/train.py:114-130): per feature, the mean over nodes of the
    product of the standardised fields (std unbiased, + 1e-8); per sample, then averaged."""
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


def test(model, test_dataloader, loss_fn, device, lat_weights=None, spatial_mask=None, channel_mask=None,
         static_channels=None, forcing_channels=None, use_residual=True):
    """One-step evaluation, same signature and return value as the reference's `test` (src/train.py:241-308):
    (mean weighted MSE, mean spatial ACC, RMSE of the unweighted errors).  The prediction step with its
    static / forcing carry-forward is the device-side rollout of `predict.py` (one step)."""
    from .predict import rollout

    model.eval()
    total, accs, raw = 0.0, [], []
    with torch.no_grad():
        for X, y in test_dataloader:
            y = y.squeeze(0) if y.dim() == 4 else y
            X, y = X.to(device), y.to(device)
            C = X.shape[-1] // model.obs_window
            steps = y.shape[-1] // C if C > 0 else 1
            y0 = y.view(y.shape[0], y.shape[1], steps, C)[:, :, 0, :].contiguous() if steps > 1 else y
            outs = rollout(model, X, 1, y=y0, static_channels=static_channels, forcing_channels=forcing_channels,
                           use_residual=use_residual)
            total += weighted_mse_loss(outs, y0, lat_weights, channel_mask, spatial_mask).item()
            raw.append(((outs - y0) ** 2).mean().item())
            skip = sorted(set(static_channels or []) | set(forcing_channels or []))
            accs.append(spatial_corr(outs, y0, exclude_channels=skip if skip else None))
    n = max(len(raw), 1)
    return total / max(len(test_dataloader), 1), sum(accs) / n, (sum(raw) / n) ** 0.5


class FlatParams:
    """All trainable parameters of a module re-pointed into ONE flat fp32 buffer, with a flat
    gradient buffer whose slices are installed as `.grad` (so the backward kernels accumulate
    straight into the all-reduce bucket)."""

    def __init__(self, module: torch.nn.Module):
        # `index[i]`: position of params[i] in module.parameters() - the index torch.optim.Adam(
        # model.parameters()) gives it in its state_dict (frozen parameters keep their slot)
        seen, self.params, self.index, pos = set(), [], [], 0
        for p in module.parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.requires_grad:
                self.params.append(p)
                self.index.append(pos)
            pos += 1
        self.num_module_params = pos
        dev = self.params[0].device
        # every parameter starts on a 256-byte boundary (the dense kernels read weight rows as
        # 16-byte vectors); the padding stays zero in the weights, the gradients and Adam's moments
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
        self.numel = total  # bucket length (with alignment padding)
        self.num_params = sum(p.numel() for p in self.params)

    def zero_grad(self):
        if self.grad.is_cuda:
            hip.zero_(self.grad)  # stream memset through the C ABI (no torch fill kernel on the step)
        else:
            self.grad.zero_()     # (CPU buckets exist only in the gloo rehearsal tests of the sharding logic)


def shard_batch(X, y, rank: int, world: int):
    """Contiguous equal shards of the batch dimension: rank r owns samples
    [r*B_local, (r+1)*B_local).  The global batch must divide evenly."""
    B = X.shape[0]
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    k = B // world
    return X[rank * k:(rank + 1) * k], y[rank * k:(rank + 1) * k]


def allreduce_gradients(flat: "FlatParams", world: int) -> float:
    """C1, the only collective on the path: ONE all-reduce (sum) of the flat gradient bucket.
    Returns the factor the optimiser must apply (1/world: the reference loss is a mean over the
    batch, src/train.py:101-102, so the global gradient is the mean of the per-rank gradients
    when every rank holds the same number of samples)."""
    if world > 1:
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)
    return 1.0 / world


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay=0) semantics over a FlatParams bucket.
    The step counter lives on the device so the update can sit inside a captured hipGraph."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=flat.flat.device)
        self.bc_dev = torch.zeros(2, dtype=torch.float32, device=flat.flat.device)

    @property
    def t(self) -> int:
        return int(self.step_dev.item())

    def step(self, grad_scale: float = 1.0):
        hip.adam_step_dev(self.flat.flat, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                          self.eps, self.wd, self.step_dev, self.bc_dev, grad_scale)

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        """The dictionary `torch.optim.Adam(model.parameters()).state_dict()` would hold after the
        same steps (per-parameter `step / exp_avg / exp_avg_sq`), so `save_checkpoint` files written
        from the fused optimiser resume under the reference's `load_checkpoint` (src/train.py:37-49)."""
        t, state = self.t, {}
        for p, i, off in zip(self.flat.params, self.flat.index, self.flat.offsets):
            k = p.numel()
            if t > 0:
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(self.flat.num_module_params))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Inverse of `state_dict`; also accepts what torch.optim.Adam wrote for the same model."""
        group = sd["param_groups"][0]
        if group.get("amsgrad", False) or group.get("maximize", False):
            raise ValueError("FusedAdam implements plain Adam only (amsgrad / maximize are not supported)")
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.wd = group.get("weight_decay", 0.0)
        steps = set()
        self.m.zero_(), self.v.zero_()
        for p, i, off in zip(self.flat.params, self.flat.index, self.flat.offsets):
            k = p.numel()
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                self.m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FusedAdam keeps one step counter; the state holds several: {sorted(steps)}")
        self.step_dev.fill_(steps.pop() if steps else 0)


_DEFER_REDUCTIONS = os.environ.get("GCL_NO_DEFER", "0") in ("0", "")


class TrainStep:
    """One optimiser step on a local batch: forward, loss, backward, [all-reduce], Adam.

    With `use_graph` (default) the launch-heavy part of the step - ~160 kernel launches - is captured
    once into a hipGraph (via torch.cuda.CUDAGraph on the stream the kernels are enqueued on) and
    replayed; inputs are copied into static buffers first.  One GPU: the whole step including Adam
    is in the graph.  Several GPUs (or `split_finish`): zero-grad + forward + loss + backward are in
    the graph and the RCCL all-reduce + Adam stay ordinary stream operations after the replay, so
    no collective is ever captured.  A step then costs the host one graph launch instead of ~160
    kernel launches, which keeps it GPU-bound on hosts with slow launch paths."""

    def __init__(self, model, lr=1e-3, lat_weights=None, channel_mask=None, spatial_mask=None, use_residual=True,
                 ar_steps=1, world_size=1, use_graph=None, split_finish=None, static_channels=None,
                 forcing_channels=None):
        self.model = model
        self.flat = FlatParams(model)
        self.opt = FusedAdam(self.flat, lr=lr)
        self.lat_weights, self.channel_mask, self.spatial_mask = lat_weights, channel_mask, spatial_mask
        self.use_residual, self.ar_steps, self.world = use_residual, ar_steps, world_size
        self.static_channels, self.forcing_channels = static_channels, forcing_channels
        # use_graph=True: the caller REQUIRES the hipGraph path (a failed capture raises);
        # use_graph=None: replay when the capture works, fall back to eager launches with a warning
        self._graph_required = use_graph is True
        if use_graph is None:
            use_graph = os.environ.get("GCL_NO_GRAPH", "0") in ("0", "")
        self.use_graph = bool(use_graph)
        self._sparse = bool(getattr(model, "using_sparse_gat", False))
        self.split_finish = (world_size > 1) if split_finish is None else bool(split_finish or world_size > 1)
        self._graph, self._sX, self._sy, self._sloss, self._eager_calls = None, None, None, None, 0
        self.capture_error = None
        if world_size > 1:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError(f"TrainStep(world_size={world_size}) needs an initialised torch.distributed process group")
            if dist.get_world_size() != world_size:
                raise RuntimeError(f"TrainStep(world_size={world_size}) but the process group has {dist.get_world_size()} ranks")
            self.sync_from_rank0()

    def sync_from_rank0(self):
        """Only gradients are all-reduced, so replicas stay identical only if they START identical: rank 0's
        parameters and Adam state are broadcast once (ranks built from different seeds or checkpoints
        would otherwise diverge silently)."""
        for t in (self.flat.flat, self.opt.m, self.opt.v, self.opt.step_dev):
            dist.broadcast(t, src=0)

    @property
    def graph_active(self) -> bool:
        """True while steps are being replayed from a captured hipGraph."""
        return bool(self.use_graph and self._graph is not None)

    @property
    def launch_mode(self) -> str:
        if self.graph_active:
            return "hipGraph replay" + (" (fwd+bwd; all-reduce + Adam eager)" if self.split_finish else "")
        return "eager" + (f" (capture failed: {self.capture_error})" if self.capture_error else "")

    def _fwd_bwd(self, X, y, threshold=0.0, epoch=0, batch_num=1):
        self.flat.zero_grad()
        loss = batch_loss(self.model, X, y, threshold, epoch, batch_num, self.lat_weights, self.ar_steps,
                          self.channel_mask, self.spatial_mask, self.static_channels, self.forcing_channels,
                          self.use_residual)
        # every parameter gradient lands in the flat bucket (preinstalled .grad slices) and is only read after the whole
        # backward, so the small final passes of the fused dense backward are queued and run in one launch per 16 layers
        defer = _DEFER_REDUCTIONS and loss.is_cuda
        if defer:
            hip.defer_begin()
        try:
            loss.backward()
        except BaseException:
            if defer:
                hip.defer_flush(drop=True)  # a failed backward: forget what it queued, launch nothing
            raise
        if defer:
            hip.defer_flush()
        return loss.detach()

    def _finish(self):
        scale = allreduce_gradients(self.flat, self.world)
        self.opt.step(grad_scale=scale)

    def _eager(self, X, y, threshold=0.0, epoch=0, batch_num=1):
        loss = self._fwd_bwd(X, y, threshold, epoch, batch_num)
        self._finish()
        return loss

    def _capture(self, X, y):
        self._sX, self._sy = X.clone(), y.clone()
        from . import models as _models

        g = torch.cuda.CUDAGraph()
        _models._graphs.pin = pinned = []  # CSR handles the captured kernels point into
        try:
            # thread_local: other threads of the process (the RCCL watchdog) may touch the runtime meanwhile
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._sloss = self._fwd_bwd(self._sX, self._sy)
                if not self.split_finish:
                    self._finish()
        finally:
            _models._graphs.pin = None
        self._graph, self._pinned = g, pinned

    def _replay(self):
        self._graph.replay()
        if self.split_finish:
            self._finish()
        return self._sloss.detach()

    def __call__(self, X, y, threshold=0.0, epoch=0, batch_num=1):
        if not self.use_graph:
            return self._eager(X, y, threshold, epoch, batch_num)
        if self._sparse and batch_num == 0:
            # SparseGATConv prunes the mesh graph on this step (src/models.py:138-149): run it eagerly and
            # drop the captured graph, which was recorded over the old edge list; two eager steps follow so
            # that the CSR of the pruned list (and of its loop-completed form) exists before re-capturing
            self._graph, self._eager_calls = None, 0
            return self._eager(X, y, threshold, epoch, batch_num)
        if self._graph is None:
            # a few eager steps first: workspaces, CSR handles, kernel attributes and the RCCL
            # communicator get set up outside the capture
            if self._eager_calls < 2:
                self._eager_calls += 1
                return self._eager(X, y, threshold, epoch, batch_num)
            try:
                self._capture(X, y)
            except Exception as e:
                self.capture_error = f"{type(e).__name__}: {str(e)[:300]}"
                self.use_graph, self._graph = False, None
                torch.cuda.synchronize()
                if self._graph_required:  # the caller asked for the graph path explicitly: no silent degradation
                    raise RuntimeError(f"TrainStep(use_graph=True): hipGraph capture failed ({self.capture_error})") from e
                import warnings

                warnings.warn(f"[TrainStep] hipGraph capture unavailable ({self.capture_error}); staying eager "
                              f"(see .launch_mode / .graph_active)", RuntimeWarning)
                return self._eager(X, y, threshold, epoch, batch_num)
            return self._replay()  # capture only records; the first replay performs this step
        if X.shape != self._sX.shape:
            return self._eager(X, y, threshold, epoch, batch_num)
        # a producer that fills input_buffers() in place passes those tensors back and saves the two device copies
        if X.data_ptr() != self._sX.data_ptr():
            self._sX.copy_(X)
        if y.data_ptr() != self._sy.data_ptr():
            self._sy.copy_(y)
        return self._replay()

    def input_buffers(self):
        """(X, y) the captured graph reads its batch from, or None before the capture / on the eager path: fill them
        in place and pass them to the step to skip its per-step copy of the batch."""
        if self.use_graph and self._graph is not None:
            return self._sX, self._sy
        return None
