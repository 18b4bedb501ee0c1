"""CPU restatement of the PyG 2.5.3 / torch arithmetic on graphcast-lite's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is imported by the product package
(`graphcast-lite_amd/`); only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may use it, and only as the checker.

PARITY UNPINNED for the PyG layers: `torch_geometric` (pinned `==2.5.3`, reference
`requirements.txt:6`) is not installed in the build image, is not vendored in the reference, and
the reference has no tests or golden outputs for these layers.  The functions below restate the
library's documented behaviour as the reference calls it; they are pinned
only by the reference authors' recorded shapes / parameter counts
(`notebooks/src/main.ipynb:178-208`, `README.md:176`) and by hand-derived micro-cases
(`tests/test_oracle.py`).  The loss, the attention-threshold schedule and the graph layout ARE
pinned by fixtures produced by running the reference's own code (`tests/golden/`).

Everything is plain torch on whatever dtype comes in (fp32 for parity, fp64 for gradcheck), so
autograd supplies the reference backward.  All ops accept an optional leading batch dimension:
`x` is `[n, F]` or `[B, n, F]`, edges index dimension -2.  Op sequence follows what PyG executes
without `torch_scatter`/`pyg_lib` (Appendix A.6): `index_select` -> message -> `scatter_add_`
into zeros, normalisation recomputed on every call.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def remove_self_loops(edge_index: torch.Tensor) -> torch.Tensor:
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep]


def add_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Append (0,0)...(n-1,n-1) AFTER the existing edges (PyG `add_self_loops`)."""
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def gcn_norm(edge_index: torch.Tensor, num_nodes: int, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """PyG `gcn_norm` with unit weights (Appendix A.1 steps 1-3): existing self-loops are
    replaced by exactly one loop per node appended last; deg = in-degree incl. the loop;
    w_e = deg[row]^-1/2 * deg[col]^-1/2 (both factors use IN-degree, also on directed graphs)."""
    ei = add_self_loops(remove_self_loops(edge_index), num_nodes)
    row, col = ei[0], ei[1]
    ones = torch.ones(ei.shape[1], dtype=dtype, device=ei.device)
    deg = torch.zeros(num_nodes, dtype=dtype, device=ei.device).scatter_add_(0, col, ones)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0.0)
    return ei, dis[row] * ones * dis[col]


def _propagate_sum(h: torch.Tensor, ei: torch.Tensor, w: Optional[torch.Tensor], n: int) -> torch.Tensor:
    """out[..., i, :] = sum_{e: col_e = i} w_e * h[..., row_e, :] via index_select + scatter_add_."""
    row, col = ei[0], ei[1]
    msg = h.index_select(-2, row)
    if w is not None:
        msg = msg * w.unsqueeze(-1)
    out = torch.zeros(h.shape[:-2] + (n, h.shape[-1]), dtype=h.dtype, device=h.device)
    idx = col.view((1,) * (h.dim() - 2) + (-1, 1)).expand_as(msg)
    return out.scatter_add_(-2, idx, msg)


def gcn_conv(x: torch.Tensor, edge_index: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """`GCNConv(in,out)` with defaults (Appendix A.1): y = A_hat (x W^T) + b; the linear map has
    no bias of its own, the layer bias is added AFTER aggregation.  Reference call site:
    `src/models.py:419`."""
    n = x.shape[-2]
    ei, w = gcn_norm(edge_index, n, x.dtype)
    h = x @ weight.t()
    y = _propagate_sum(h, ei, w, n)
    if bias is not None:
        y = y + bias
    return y


def segment_softmax(e: torch.Tensor, col: torch.Tensor, n: int) -> torch.Tensor:
    """PyG `softmax(src, index)` over edges sharing a target (Appendix A.2 step 4): subtract the
    per-target max (taken on detached values), exponentiate, divide by (segment sum + 1e-16).
    `e` is `[..., E, H]`."""
    lead = e.shape[:-2]
    H = e.shape[-1]
    idx = col.view((1,) * len(lead) + (-1, 1)).expand_as(e)
    with torch.no_grad():
        m = torch.full(lead + (n, H), float("-inf"), dtype=e.dtype, device=e.device)
        m = m.scatter_reduce(-2, idx, e.detach(), reduce="amax", include_self=True)
    p = (e - m.gather(-2, idx)).exp()
    s = torch.zeros(lead + (n, H), dtype=e.dtype, device=e.device).scatter_add_(-2, idx, p)
    return p / (s.gather(-2, idx) + 1e-16)


def gat_conv(
    x: torch.Tensor,
    edge_index: torch.Tensor,
    weight: torch.Tensor,   # [H*C, in]   (`lin.weight`, shared by source and target)
    att_src: torch.Tensor,  # [1, H, C]
    att_dst: torch.Tensor,  # [1, H, C]
    bias: Optional[torch.Tensor],  # [C]
    heads: int,
    negative_slope: float = 0.2,
):
    """`GATConv(in, C, heads=H, concat=False)` with defaults (Appendix A.2).  Reference call
    sites: `src/models.py:425` (GATConv) and `:135` (SparseGATConv).
    Returns (y [..., n, C], edge_index_with_loops [2, E'], alpha [..., E', H])."""
    n = x.shape[-2]
    H = heads
    C = weight.shape[0] // H
    h = (x @ weight.t()).view(x.shape[:-1] + (H, C))
    a_s = (h * att_src.view(H, C)).sum(-1)  # [..., n, H]
    a_d = (h * att_dst.view(H, C)).sum(-1)
    ei = add_self_loops(remove_self_loops(edge_index), n)
    row, col = ei[0], ei[1]
    e = F.leaky_relu(a_s.index_select(-2, row) + a_d.index_select(-2, col), negative_slope)
    alpha = segment_softmax(e, col, n)  # [..., E', H]
    msg = h.index_select(-3, row) * alpha.unsqueeze(-1)  # [..., E', H, C]
    out = torch.zeros(x.shape[:-1] + (H, C), dtype=x.dtype, device=x.device)
    idx = col.view((1,) * (x.dim() - 2) + (-1, 1, 1)).expand_as(msg)
    out = out.scatter_add_(-3, idx, msg)
    y = out.mean(dim=-2)
    if bias is not None:
        y = y + bias
    return y, ei, alpha


def sparse_gat_prune(edge_index_with_loops: torch.Tensor, alpha: torch.Tensor, threshold: float):
    """`SparseGATConv.forward` tail for `batch_num == 0` (`src/models.py:138-149`): keep the
    columns whose attention is >= threshold.  `alpha` is `[E']` (heads squeezed, so H must be 1)."""
    mask = alpha >= threshold
    return edge_index_with_loops[:, mask], alpha[mask]


def simple_conv_mean(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """`SimpleConv(aggr="mean")` (Appendix A.3): mean of in-neighbours, no self-loop, exact 0
    for nodes without in-edges, no parameters.  Reference call site: `src/models.py:414`."""
    n = x.shape[-2]
    s = _propagate_sum(x, edge_index, None, n)
    ones = torch.ones(edge_index.shape[1], dtype=x.dtype, device=x.device)
    cnt = torch.zeros(n, dtype=x.dtype, device=x.device).scatter_add_(0, edge_index[1], ones)
    return s / cnt.clamp(min=1).unsqueeze(-1)


def pyg_layer_norm(x: torch.Tensor, weight, bias, mode: str = "graph", eps: float = 1e-5) -> torch.Tensor:
    """PyG `LayerNorm(C, eps=1e-5, affine=True, mode)` (Appendix A.4).
    node : `F.layer_norm` over the channel dimension of every row.
    graph: statistics over ALL elements of one sample's `[n, C]` tensor, eps added to the std
           (`(x - mean) / (std_biased + eps)`), then the affine map; per sample under batching."""
    if mode == "node":
        return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)
    if mode == "graph":
        mean = x.mean(dim=(-2, -1), keepdim=True)
        xc = x - mean
        var = (xc * xc).mean(dim=(-2, -1), keepdim=True)
        out = xc / (var.sqrt() + eps)
        if weight is not None:
            out = out * weight + bias
        return out
    raise ValueError(f"Unknown normalization mode: {mode}")


def prelu(x: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """`nn.PReLU()` with one scalar slope (Appendix A.5)."""
    return F.prelu(x, a)


def scatter_mean_rows(values: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """`torch_geometric.utils.scatter(values, index, dim=0, dim_size=n, reduce="mean")` as the
    reference calls it on edge rows (`src/models.py:220-221`): per-target sum / max(count, 1),
    rows without any source are exact zeros.  values `[..., E, D]`, index `[E]`."""
    out = torch.zeros(values.shape[:-2] + (n, values.shape[-1]), dtype=values.dtype, device=values.device)
    out.index_add_(-2, index, values)
    cnt = torch.zeros(n, dtype=values.dtype, device=values.device).index_add_(
        0, index, torch.ones(index.numel(), dtype=values.dtype, device=values.device))
    return out / cnt.clamp(min=1).unsqueeze(-1)
