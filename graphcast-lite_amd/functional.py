"""Module-level autograd Functions over the C ABI (one Function per MLP / conv stack).

Each Function runs a whole reference module (`MLP.forward` src/models.py:106-109; the GCN / GAT /
SparseGAT / SimpleConv branches of `GraphLayer.forward` src/models.py:406-440) as a sequence of
`libgcl_hip` launches and keeps only what its backward needs.  Buffers passed between layers hold
PRE-activation values (see include/gcl.h, "Activation chaining").

Gradient delivery: when a parameter already has a `.grad` tensor the backward ACCUMULATES into it
in place on the device and returns None for that input (fused accumulation - no extra add kernel
per parameter, and gradients land directly in the flat bucket that the data-parallel all-reduce
uses); otherwise it returns a fresh gradient tensor and autograd stores it as usual.
"""
from typing import List, Optional

import torch

from . import hip


def _grad_slot(p: Optional[torch.Tensor], needs: bool):
    """(tensor to write into, accumulate?, return_value_is_none?)"""
    if p is None or not needs:
        return None, False, True
    if p.grad is not None:
        return p.grad, True, True
    return torch.zeros_like(p), False, False


class _Grads:
    """Collects per-parameter gradient destinations for one backward call."""

    def __init__(self, params: List[Optional[torch.Tensor]], needs: List[bool]):
        self.dst, self.acc, self.ret_none = [], [], []
        for p, nd in zip(params, needs):
            d, a, r = _grad_slot(p, nd)
            self.dst.append(d)
            self.acc.append(a)
            self.ret_none.append(r)

    def out(self):
        return tuple(None if r else d for d, r in zip(self.dst, self.ret_none))


def _flat3(x: torch.Tensor):
    """[B,n,F] (or [n,F]) -> contiguous [B,n,F]."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x


# ------------------------------------------------------------------------------------------------
# MLP: Linear -> PReLU -> ... -> Linear (-> LayerNorm node)
# params layout: [W1, b1, a1, W2, b2, a2, ..., WL, bL] (+ [gamma, beta])
# ------------------------------------------------------------------------------------------------
class MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, has_ln: bool, eps: float, *params):
        L = (len(params) - (2 if has_ln else 0) + 1) // 3
        shape = x.shape
        x2 = hip.rows2d(x.detach())
        zs = []
        cur, slope = x2, None
        for k in range(L):
            W, b = params[3 * k], params[3 * k + 1]
            z = hip.linear_fwd(cur, W.detach(), b.detach(), slope)
            zs.append(z)
            cur = z
            slope = params[3 * k + 2].detach() if k < L - 1 else None
        stats = None
        out = cur
        if has_ln:
            out, stats = hip.layernorm_fwd(cur, params[-2].detach(), params[-1].detach(), eps)
        ctx.owner, ctx.has_ln, ctx.L = owner, has_ln, L
        ctx.x2, ctx.zs, ctx.stats = x2, zs, stats
        ctx.params = params
        return out.view(shape[:-1] + (out.shape[-1],))

    @staticmethod
    def backward(ctx, dy):
        params, L = ctx.params, ctx.L
        needs = list(ctx.needs_input_grad[4:])
        G = _Grads(list(params), needs)
        dy2 = hip.rows2d(dy)
        zs = ctx.zs
        if ctx.has_ln:
            gi, bi = len(params) - 2, len(params) - 1
            dgam = G.dst[gi] if G.dst[gi] is not None else torch.zeros_like(params[gi])
            dbet = G.dst[bi] if G.dst[bi] is not None else torch.zeros_like(params[bi])
            dz = hip.layernorm_bwd(dy2, zs[-1], params[gi].detach(), ctx.stats, dgam, dbet, G.acc[gi] and G.acc[bi])
        else:
            dz = dy2
        dx = None
        for k in range(L - 1, -1, -1):
            W = params[3 * k].detach()
            inp = ctx.x2 if k == 0 else zs[k - 1]
            slope = None if k == 0 else params[3 * k - 1].detach()
            wi, bi = 3 * k, 3 * k + 1
            dW = G.dst[wi] if G.dst[wi] is not None else torch.zeros_like(params[wi])
            db = G.dst[bi]
            if k > 0:
                # one fused launch: dz_{k-1}, dW_k, db_k, d(slope_{k-1})
                dz = hip.linear_bwd_all(dz, W, inp, slope, G.dst[3 * k - 1], dW, db, None, G.acc[wi])
            else:
                hip.linear_bwd_dw(dz, inp, None, dW, db, G.acc[wi])
                if ctx.needs_input_grad[0]:
                    dx = hip.linear_bwd_dx(dz, W, None, None, None)
        if dx is not None:
            dx = dx.view(dy.shape[:-1] + (dx.shape[-1],))
        return (dx, None, None, None) + G.out()


# ------------------------------------------------------------------------------------------------
# GCN stack: conv -> PReLU(shared) -> conv -> ... -> conv (-> LayerNorm node)
# params layout: [W1, b1, ..., WL, bL, slope] (+ [gamma, beta]);  slope may be None when L == 1
# ------------------------------------------------------------------------------------------------
class GCNStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, graph, L: int, has_ln: bool, eps: float, *params):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        B, n, _ = x3.shape
        slope_p = params[2 * L]
        ps = []  # pre-activation outputs of every conv
        cur, slope = x3, None
        for k in range(L):
            W, b = params[2 * k].detach(), params[2 * k + 1].detach()
            Fout = W.shape[0]
            ldh = (Fout + 3) // 4 * 4  # padded scratch so the gather can use 16-B loads
            h = hip.linear_fwd(cur.view(B * n, -1), W, None, slope, ld_out=ldh)
            h3 = torch.as_strided(h, (B, n, Fout), (n * ldh, ldh, 1))
            p = hip.aggregate(graph, h3, b)
            ps.append(p)
            cur = p
            slope = slope_p.detach() if slope_p is not None else None
        out, stats = cur, None
        if has_ln:
            o2, stats = hip.layernorm_fwd(cur.view(B * n, -1), params[-2].detach(), params[-1].detach(), eps)
            out = o2.view(B, n, -1)
        ctx.owner, ctx.graph, ctx.L, ctx.has_ln = owner, graph, L, has_ln
        ctx.x3, ctx.ps, ctx.stats, ctx.params, ctx.squeeze = x3, ps, stats, params, squeeze
        return out[0] if squeeze else out

    @staticmethod
    def backward(ctx, dy):
        params, L, graph = ctx.params, ctx.L, ctx.graph
        needs = list(ctx.needs_input_grad[6:])
        G = _Grads(list(params), needs)
        dy3 = _flat3(dy)
        B, n, _ = dy3.shape
        ps = ctx.ps
        if ctx.has_ln:
            gi, bi = len(params) - 2, len(params) - 1
            dgam = G.dst[gi] if G.dst[gi] is not None else torch.zeros_like(params[gi])
            dbet = G.dst[bi] if G.dst[bi] is not None else torch.zeros_like(params[bi])
            dp = hip.layernorm_bwd(dy3.view(B * n, -1), ps[-1].view(B * n, -1), params[gi].detach(), ctx.stats, dgam,
                                   dbet, G.acc[gi] and G.acc[bi]).view(B, n, -1)
        else:
            dp = dy3
        si = 2 * L
        dsl = G.dst[si] if params[si] is not None else None
        slope_t = params[si].detach() if params[si] is not None else None
        dx = None
        if G.dst[2 * L - 1] is not None:  # bias of the last conv: its dp comes from outside this stack
            hip.colsum(dp.reshape(B * n, -1), G.dst[2 * L - 1], G.acc[2 * L - 1])
        for k in range(L - 1, -1, -1):
            W = params[2 * k].detach()
            wi = 2 * k
            inp = (ctx.x3 if k == 0 else ps[k - 1]).view(B * n, -1)
            dh2 = hip.aggregate(graph, dp, None, transpose=True).view(B * n, -1)
            dW = G.dst[wi] if G.dst[wi] is not None else torch.zeros_like(params[wi])
            if k > 0:
                # one fused launch: dp_{k-1} (with PReLU'), dW_k, d(slope) and the bias gradient of
                # conv k-1 (= column sums of dp_{k-1})
                dp = hip.linear_bwd_all(dh2, W, inp, slope_t, dsl, dW, None, G.dst[2 * k - 1], G.acc[wi]).view(B, n, -1)
            else:
                hip.linear_bwd_dw(dh2, inp, None, dW, None, G.acc[wi])
                if ctx.needs_input_grad[0]:
                    dx = hip.linear_bwd_dx(dh2, W, None, None, None).view(B, n, -1)
        if dx is not None and ctx.squeeze:
            dx = dx[0]
        return (dx, None, None, None, None, None) + G.out()


# ------------------------------------------------------------------------------------------------
# GATConv / SparseGATConv layer.  alpha_edges (second output, non-differentiable) is the attention
# of sample 0 in PyG edge order [E', H] - what SparseGATConv thresholds (src/models.py:136-149).
# ------------------------------------------------------------------------------------------------
class GATLayerFn(torch.autograd.Function):
    """One GATConv (with the shared PReLU applied to its input on load)."""

    @staticmethod
    def forward(ctx, x, owner, graph, H: int, want_alpha: bool, slope, W, att_src, att_dst, bias):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        B, n, _ = x3.shape
        Cc = W.shape[0] // H
        sl = slope.detach() if slope is not None else None
        h = hip.linear_fwd(x3.view(B * n, -1), W.detach(), None, sl).view(B, n, H * Cc)
        y, a_s, a_d, alpha = hip.gat_fwd(graph, h, att_src.detach().reshape(-1), att_dst.detach().reshape(-1),
                                         bias.detach(), H, Cc)
        alpha_edges = hip.gat_alpha_edge_order(graph, alpha[0], H) if want_alpha else torch.empty(0, device=x3.device)
        ctx.owner, ctx.graph, ctx.H, ctx.Cc, ctx.squeeze = owner, graph, H, Cc, squeeze
        ctx.x3, ctx.h, ctx.a_s, ctx.a_d, ctx.alpha = x3, h, a_s, a_d, alpha
        ctx.params = (slope, W, att_src, att_dst, bias)
        ctx.mark_non_differentiable(alpha_edges)
        return (y[0] if squeeze else y), alpha_edges

    @staticmethod
    def backward(ctx, dy, _dalpha):
        slope, W, att_src, att_dst, bias = ctx.params
        needs = list(ctx.needs_input_grad[5:])
        G = _Grads([slope, W, att_src, att_dst, bias], needs)
        graph, H, Cc = ctx.graph, ctx.H, ctx.Cc
        dy3 = _flat3(dy)
        B, n, _ = dy3.shape
        d_as = G.dst[2] if G.dst[2] is not None else torch.zeros_like(att_src)
        d_ad = G.dst[3] if G.dst[3] is not None else torch.zeros_like(att_dst)
        acc = G.acc[2] and G.acc[3] and (G.dst[4] is None or G.acc[4])
        if not acc:
            # mixed accumulate states: run non-accumulating into temporaries and add
            t_as, t_ad = torch.zeros_like(att_src), torch.zeros_like(att_dst)
            t_b = torch.zeros_like(bias) if G.dst[4] is not None else None
            dh = hip.gat_bwd(graph, dy3, ctx.h, att_src.detach().reshape(-1), att_dst.detach().reshape(-1), ctx.a_s,
                             ctx.a_d, ctx.alpha, t_as.view(-1), t_ad.view(-1), t_b, False, H, Cc)
            for dst, t, a in ((G.dst[2], t_as, G.acc[2]), (G.dst[3], t_ad, G.acc[3]), (G.dst[4], t_b, G.acc[4])):
                if dst is not None:
                    dst.add_(t) if a else dst.copy_(t)
        else:
            dh = hip.gat_bwd(graph, dy3, ctx.h, att_src.detach().reshape(-1), att_dst.detach().reshape(-1), ctx.a_s,
                             ctx.a_d, ctx.alpha, d_as.view(-1), d_ad.view(-1), G.dst[4], True, H, Cc)
        dh2 = dh.view(B * n, -1)
        inp = ctx.x3.view(B * n, -1)
        sl = slope.detach() if slope is not None else None
        if G.dst[1] is not None:
            hip.linear_bwd_dw(dh2, inp, sl, G.dst[1], None, G.acc[1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = hip.linear_bwd_dx(dh2, W.detach(), inp if sl is not None else None, sl, G.dst[0] if sl is not None else None)
            dx = dx.view(B, n, -1)
            if ctx.squeeze:
                dx = dx[0]
        return (dx, None, None, None, None) + G.out()


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, eps, gamma, beta):
        x2 = hip.rows2d(x.detach())
        y, stats = hip.layernorm_fwd(x2, gamma.detach(), beta.detach(), eps)
        ctx.x2, ctx.stats, ctx.params = x2, stats, (gamma, beta)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        gamma, beta = ctx.params
        G = _Grads([gamma, beta], list(ctx.needs_input_grad[3:]))
        dg = G.dst[0] if G.dst[0] is not None else torch.zeros_like(gamma)
        db = G.dst[1] if G.dst[1] is not None else torch.zeros_like(beta)
        dx = hip.layernorm_bwd(hip.rows2d(dy), ctx.x2, gamma.detach(), ctx.stats, dg, db, G.acc[0] and G.acc[1])
        return (dx.view(dy.shape), None, None) + G.out()


class GraphNormFn(torch.autograd.Function):
    """PyG LayerNorm(mode="graph"): statistics over all n*F elements of each sample."""

    @staticmethod
    def forward(ctx, x, owner, eps, gamma, beta):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        y, stats = hip.graphnorm_fwd(x3, gamma.detach(), beta.detach(), eps)
        ctx.x3, ctx.stats, ctx.params, ctx.eps, ctx.squeeze = x3, stats, (gamma, beta), eps, squeeze
        return y[0] if squeeze else y

    @staticmethod
    def backward(ctx, dy):
        gamma, beta = ctx.params
        G = _Grads([gamma, beta], list(ctx.needs_input_grad[3:]))
        dg = G.dst[0] if G.dst[0] is not None else torch.zeros_like(gamma)
        db = G.dst[1] if G.dst[1] is not None else torch.zeros_like(beta)
        dx = hip.graphnorm_bwd(_flat3(dy), ctx.x3, gamma.detach(), ctx.stats, dg, db, G.acc[0] and G.acc[1], ctx.eps)
        return ((dx[0] if ctx.squeeze else dx), None, None) + G.out()


class MeanAggFn(torch.autograd.Function):
    """SimpleConv(aggr="mean") (src/models.py:414)."""

    @staticmethod
    def forward(ctx, x, graph):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        ctx.graph, ctx.squeeze = graph, squeeze
        y = hip.aggregate(graph, x3, None)
        return y[0] if squeeze else y

    @staticmethod
    def backward(ctx, dy):
        dx = hip.aggregate(ctx.graph, _flat3(dy), None, transpose=True)
        return (dx[0] if ctx.squeeze else dx), None


class AssembleFn(torch.autograd.Function):
    """`_preprocess_input` (src/models.py:776-806)."""

    @staticmethod
    def forward(ctx, x, grid_static, mesh_static):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        ctx.squeeze, ctx.G, ctx.Cdyn = squeeze, x3.shape[1], x3.shape[2]
        out = hip.assemble_input(x3, grid_static, mesh_static)
        return out[0] if squeeze else out

    @staticmethod
    def backward(ctx, dy):
        dy3 = dy if dy.dim() == 3 else dy.unsqueeze(0)
        dx = dy3[:, : ctx.G, : ctx.Cdyn].contiguous()
        return (dx[0] if ctx.squeeze else dx), None, None


class WeightedMSEFn(torch.autograd.Function):
    """Residual add + weighted MSE (src/train.py:203-213, 85-102); the gradient with respect to
    the model output is produced by the same kernel that computes the loss."""

    @staticmethod
    def forward(ctx, delta, x_last, y, node_w, chan_w, inv_wsum: float):
        loss, dd, _ = hip.wmse_fwd_bwd(delta.detach(), x_last, y, node_w, chan_w, inv_wsum, 1.0, want_grad=True)
        ctx.dd = dd
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.dd * g, None, None, None, None, None


class Gather2Fn(torch.autograd.Function):
    """Row gather from two sources with precomputed index maps (see hip.gather2_rows).
    maps = (map_a, map_b, inv_a, inv_b): forward maps have length nd; inv_x[j] = destination row
    fed by row j of source x, or -1.  A source with leading dimension 1 is broadcast over the
    batch and receives the batch-summed gradient."""

    @staticmethod
    def forward(ctx, a, b, maps, nd: int, B: int):
        map_a, map_b, inv_a, inv_b = maps
        a3 = a.detach()
        b3 = b.detach() if b is not None else None
        if not a3.is_contiguous():
            a3 = a3.contiguous()
        if b3 is not None and not b3.is_contiguous():
            b3 = b3.contiguous()
        ctx.maps, ctx.B = maps, B
        ctx.sa, ctx.sb = a3.shape, (b3.shape if b3 is not None else None)
        return hip.gather2_rows(a3, map_a, b3, map_b, nd, B)

    @staticmethod
    def backward(ctx, g):
        map_a, map_b, inv_a, inv_b = ctx.maps
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            bc = ctx.sa[0] == 1 and ctx.B > 1
            da = hip.gather2_rows(g, inv_a, None, None, ctx.sa[1], ctx.B, sum_batch=bc)
        if ctx.sb is not None and ctx.needs_input_grad[1]:
            bc = ctx.sb[0] == 1 and ctx.B > 1
            db = hip.gather2_rows(g, inv_b, None, None, ctx.sb[1], ctx.B, sum_batch=bc)
        return da, db, None, None, None
