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

import os

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
                dz = hip.linear_bwd_all(dz, W, inp, slope, G.dst[3 * k - 1], dW, db, None, G.acc[wi], acc_db=G.acc[bi])
            elif ctx.needs_input_grad[0]:
                # first layer with a differentiable input: dx, dW and db from one read of dz
                dx = hip.linear_bwd_all(dz, W, inp, None, None, dW, db, None, G.acc[wi], acc_db=G.acc[bi],
                                        act=hip.ACT_NONE)
            elif db is not None and G.acc[bi] != G.acc[wi]:  # the dW kernel shares one flag between dW and db
                hip.linear_bwd_dw(dz, inp, None, dW, None, G.acc[wi])
                hip.colsum(dz, db, G.acc[bi])
            else:
                hip.linear_bwd_dw(dz, inp, None, dW, db, G.acc[wi])
        if dx is not None:
            dx = dx.view(dy.shape[:-1] + (dx.shape[-1],))
        return (dx, None, None, None) + G.out()


# ------------------------------------------------------------------------------------------------
# GCN stack: conv -> PReLU(shared) -> conv -> ... -> conv (-> LayerNorm node)
# params layout: [W1, b1, ..., WL, bL, slope] (+ [gamma, beta]);  slope may be None when L == 1
# ------------------------------------------------------------------------------------------------
_LN_COLSUM = os.environ.get("GCL_NO_LN_COLSUM", "0") in ("0", "")
_ROWS_OUT = os.environ.get("GCL_NO_ROWS_OUT", "0") in ("0", "")


def _lat_first_layer_bwd(lat, enc3, dz3, W, dW, acc_dw: bool, want_dx: bool, Pc=None, enc_shape=None):
    """Dense backward of a first processor layer whose input was read through a LatSource: dz3 [B, M, D'] is the
    gradient of the layer's transformed mesh rows (GCN: A^T dp, GAT: dh).  Returns the gradient of the encoder output
    [B, ne, D] (or None when the shared landing buffer took it / no gradient is wanted); dW is written / accumulated in
    place.  Pc: the compact encoder rows [B, Md + r, D] when the forward already copied them."""
    _, _, inv_a, inv_fold = lat.maps
    B, ne, D = enc_shape if enc3 is None else enc3.shape  # (enc3 may be None when Pc is given: only its shape is needed)
    G, Md, r = lat.G, lat.Md, lat.r
    nc = Md + r
    Fo = dz3.shape[-1]
    dz3 = dz3 if dz3.is_contiguous() else dz3.contiguous()
    # compact gradient rows [B, Md + r, Fo]: dependent rows gathered, folded rows summed over the batch
    dzc = torch.empty(B, nc, Fo, dtype=torch.float32, device=dz3.device)
    if Md > 0:
        hip.gather2_rows(dz3, inv_a[G: G + Md], None, None, Md, B, out=dzc[:, :Md])
    if r > 1:
        hip.gather2_rows(dz3, inv_fold, None, None, B * r, B, sum_batch=True, out=dzc[:, Md:], deal=r)
    elif r == 1:
        tmp = hip.gather2_rows(dz3, inv_fold, None, None, B * r, B, sum_batch=True)
        hip.copy_rows(tmp.view(B, r, Fo), dzc[:, Md:])
    if Pc is None:  # the encoder rows behind the mesh latents
        Pc = hip.copy_rows(enc3[:, G:, :], torch.empty(B, nc, D, dtype=torch.float32, device=dz3.device))
    land = lat.landing
    shared = land is not None and land.buf is not None
    if not want_dx:
        hip.linear_bwd_dw(dzc.view(B * nc, Fo), Pc.view(B * nc, D), None, dW, None, acc_dw)
        dxc = None
    else:
        dxc = hip.linear_bwd_all(dzc.view(B * nc, Fo), W, Pc.view(B * nc, D), None, None, dW, None, None, acc_dw,
                                 act=hip.ACT_NONE).view(B, nc, D)
    if shared:
        if dxc is not None:
            hip.copy_rows(dxc, land.buf[:, G:])  # head rows: written by the decoder-input gather's backward
        land.buf, land.mesh_pending, land.handed_over = None, False, False
        return None
    if dxc is None:
        return None
    out = torch.zeros(B, ne, D, dtype=torch.float32, device=dz3.device)
    hip.copy_rows(dxc, out[:, G:])
    return out


class LatSource:
    """The mesh latents of the compact pipeline as a VIEW of the encoder output enc [B, ne, D] (rows per sample:
    [G grid | Md batch-dependent mesh | r folded batch-invariant mesh rows], see MeshLatFn): mesh row i of sample b is
    enc[b, tab[i]] (tab[i] >= 0) or the flat row ~tab[i] of enc viewed as [B * ne, D] (tab[i] < 0).  A GCN stack given
    one reads its first layer's input through the table (gcl_gcn_layer_fwd_tab) - the [B, M, D] latents are never
    written - and runs that layer's dense backward on the COMPACT rows: by linearity the gradient of a folded row is
    (sum_b dz[b, i]) W and its dW term (sum_b dz[b, i])^T enc_row, so 81 % of the rows (64x32 grid) cost one batch sum
    instead of B dense rows.  maps = MeshLatFn's (map_a, map_b, inv_a, inv_fold)."""

    def __init__(self, tab, maps, M: int, G: int, Md: int, r: int, landing=None):
        self.tab, self.maps, self.M, self.G, self.Md, self.r, self.landing = tab, maps, M, G, Md, r, landing
        self._compact = {}

    def compact_tab(self, ne: int):
        """The table relative to the COMPACT rows enc[:, G:, :] copied to [B, Md + r, D] (a GAT layer transforms those
        first): own rows shift by G, flat row b' * ne + G + k becomes b' * (Md + r) + k."""
        t = self._compact.get(ne)
        if t is None:
            tab = self.tab.to(torch.int64)
            f = -tab - 1
            bq = torch.div(f, ne, rounding_mode="floor")
            flat_c = bq * (self.Md + self.r) + (f - bq * ne - self.G)
            t = torch.where(tab >= 0, tab - self.G, -flat_c - 1).to(torch.int32).contiguous()
            self._compact[ne] = t
        return t


class GCNStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, graph, L: int, has_ln: bool, eps: float, out_rows: int, *params):
        """out_rows > 0: only the first `out_rows` rows are returned (the decoder keeps the grid rows,
        src/models.py:870-872); the slice is part of this Function so that its backward receives the gradient of
        the slice and widens it with ONE pass of gcl_pad_rows (no zero-fill + copy by autograd)."""
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        B, n, _ = x3.shape
        lat = getattr(owner, "_lat_src", None)
        if lat is not None:
            owner._lat_src = None
            n = lat.M  # x is the encoder output [B, ne, D]; the stack runs on the M mesh rows it is read into
            if not x3.is_contiguous():
                x3 = x3.contiguous()
            if lat.landing is not None:
                lat.landing.mesh_pending = True  # this Function's backward fills the tail rows of the shared buffer (as MeshLatFn)
        ctx.lat = lat
        slope_p = params[2 * L]
        # activation between the convs: learnable PReLU slope (params[2L]), SiLU, or ReLU as a PReLU
        # with the owner's constant zero slope (src/models.py:154-163, :316)
        akind = getattr(owner, "act_kind", hip.ACT_PRELU)
        slope_t = slope_p.detach() if slope_p is not None else getattr(owner, "const_slope", None)
        ps = []  # pre-activation outputs of every conv
        cur = x3
        pad_last = None
        for k in range(L):
            W, b = params[2 * k].detach(), params[2 * k + 1].detach()
            Fout = W.shape[0]
            ldh = (Fout + 3) // 4 * 4  # padded scratch so the gather can use 16-B loads
            act_k, slope_k = (akind, slope_t) if k > 0 else (hip.ACT_NONE, None)
            if k == 0 and lat is not None:
                p = hip.gcn_layer_fwd_tab(graph, x3, lat.tab, act_k, slope_k, W, b)
                if ldh != Fout:
                    p = p.contiguous()
                ps.append(p)
                cur = p
                continue
            if hip.gcn_layer_fusable(graph, cur, W.shape[1], Fout):
                # ONE kernel (csrc/gcn_layer.hip): gather-aggregate the activated input rows, then the dense
                # transform; an output width that is not a multiple of 4 (33 / 19 variables) is stored ldh wide
                # with zero padding columns, so the layer and its backward stay on 16-byte rows
                # the last conv of a stack whose caller keeps only the first rows (decoder: grid rows) computes only those
                last_rows = int(out_rows) if (k == L - 1 and out_rows and out_rows < n and not has_ln and _ROWS_OUT) else None
                p = hip.gcn_layer_fwd(graph, cur, act_k, slope_k, W, b, rows_out=last_rows)
                if k == L - 1 and ldh != Fout and not has_ln:
                    pad_last = (ldh, Fout, None)
                elif ldh != Fout:
                    p = p.contiguous()  # an odd width that feeds another layer / the LayerNorm: dense rows
                ps.append(p)
                cur = p
                continue
            if k == L - 1 and ldh != Fout and not has_ln:
                # two-kernel path of the same padded layer: zero weight rows / bias entries for the extra columns
                Wp = torch.nn.functional.pad(W, (0, 0, 0, ldh - Fout))
                bp = torch.nn.functional.pad(b, (0, ldh - Fout))
                h = hip.linear_fwd(cur.reshape(B * n, -1), Wp, None, slope_k, act=act_k)
                p = hip.aggregate(graph, h.view(B, n, ldh), bp)
                ps.append(p)
                cur = p[..., :Fout]
                pad_last = (ldh, Fout, Wp)
                continue
            h = hip.linear_fwd(cur.reshape(B * n, -1) if not cur.is_contiguous() else cur.view(B * n, -1), W, None, slope_k,
                               ld_out=ldh, act=act_k)
            h3 = torch.as_strided(h, (B, n, Fout), (n * ldh, ldh, 1))
            p = hip.aggregate(graph, h3, b)
            ps.append(p)
            cur = p
        ctx.akind, ctx.slope_t, ctx.pad_last = akind, slope_t, pad_last
        out, stats = cur, None
        ctx.land = getattr(owner, "_grad_src", None)
        if ctx.land is not None:
            owner._grad_src = None
            if has_ln and not (out_rows and out_rows < n) and not squeeze:
                ctx.land.proc_ready = True
            else:
                ctx.land = None
        if has_ln:
            land = ctx.land
            if land is not None and land.dec_buf is not None and cur.is_contiguous() and land.dec_buf.shape[2] == cur.shape[2]:
                # the output is only consumed by the decoder-input gather: the rows it takes are written straight into
                # the decoder's input, the others not at all (GradLanding.dec_buf); autograd sees a stride-0 token
                stats = hip.layernorm_fwd_map(cur.view(B * n, -1), params[-2].detach(), params[-1].detach(), eps,
                                              land.dec_buf, land.dec_map)
                land.dec_filled = True
                out = cur.new_zeros(()).expand(B, n, cur.shape[2])
            else:
                o2, stats = hip.layernorm_fwd(cur.view(B * n, -1), params[-2].detach(), params[-1].detach(), eps)
                out = o2.view(B, n, -1)
        ctx.owner, ctx.graph, ctx.L, ctx.has_ln = owner, graph, L, has_ln
        ctx.x3, ctx.ps, ctx.stats, ctx.params, ctx.squeeze = x3, ps, stats, params, squeeze
        ctx.n_rows, ctx.out_rows = n, (int(out_rows) if out_rows and out_rows < n else 0)
        if ctx.out_rows:
            out = out[:, :ctx.out_rows, :]
        return out[0] if squeeze else out

    @staticmethod
    def backward(ctx, dy):
        params, L, graph = ctx.params, ctx.L, ctx.graph
        needs = list(ctx.needs_input_grad[7:])
        G = _Grads(list(params), needs)
        land = ctx.land
        dy_map = None
        if land is not None and land.proc_src is not None:
            if all(st == 0 for st in dy.stride()):
                # the incoming `dy` is the gather's stride-0 token: the real gradient is the gather's own incoming
                # gradient, read through its row map
                dy_map = (land.proc_src, land.proc_map)
                dy3 = dy
            else:
                # something ELSE also sent gradient to the processor's output (a hook, a second consumer): autograd has
                # summed it with the token into a real tensor, so the shortcut would drop it - materialise the gather's
                # part and take the ordinary path
                src, pmap = land.proc_src, land.proc_map
                dy3 = _flat3(dy) + hip.gather2_rows(src, pmap, None, None, dy.shape[-2], src.shape[0])
            land.proc_src = land.proc_map = None
            land.proc_ready = False
        else:
            dy3 = _flat3(dy)
        B, n = dy3.shape[0], ctx.n_rows
        ps = ctx.ps
        pad = ctx.pad_last
        dy_rows = dy3  # gradient of the rows that were returned (bias gradient of the last conv sums these)
        if ctx.out_rows and (ctx.has_ln or pad is None):
            dy3 = hip.pad_rows(dy3, n, dy3.shape[2])  # rows that were not returned carry a zero gradient
        bi_last = 2 * L - 1
        cs_done = False
        if ctx.has_ln:
            gi, bi = len(params) - 2, len(params) - 1
            dgam = G.dst[gi] if G.dst[gi] is not None else torch.zeros_like(params[gi])
            dbet = G.dst[bi] if G.dst[bi] is not None else torch.zeros_like(params[bi])
            # the LayerNorm backward also sums its dx over the rows: that IS the bias gradient of the last conv
            cs = G.dst[bi_last] if (pad is None and _LN_COLSUM) else None
            dp = hip.layernorm_bwd(None if dy_map is not None else dy3.view(B * n, -1), ps[-1].view(B * n, -1),
                                   params[gi].detach(), ctx.stats, dgam, dbet, G.acc[gi] and G.acc[bi], colsum_dx=cs,
                                   acc_colsum=bool(cs is not None and G.acc[bi_last]), dy_map=dy_map).view(B, n, -1)
            cs_done = cs is not None
        else:
            dp = dy3
        si = 2 * L
        dsl = G.dst[si] if params[si] is not None else None
        slope_t, akind = ctx.slope_t, ctx.akind
        dx = None
        Fo = params[2 * (L - 1)].shape[0]
        Fp = Fo
        if pad is not None:
            # The last conv ran Fp = roundup(Fout, 4) wide (zero padding columns).  Its gradient is widened to that
            # layout (extra rows / columns 0) in one pass; the dense backward then reads it Fout wide with row stride Fp
            # - the padding columns only ever meet zero weights - so dW / dX need no padded weight copy.
            Fp = pad[0]
            dp = hip.pad_rows(dy_rows if not ctx.has_ln else dp, n, Fp)
        if G.dst[bi_last] is not None and not cs_done:  # bias of the last conv: its dp comes from outside this stack
            src = dy_rows if not ctx.has_ln else dp[..., :Fo]
            hip.colsum(src.reshape(-1, Fo), G.dst[bi_last], G.acc[bi_last])
        for k in range(L - 1, -1, -1):
            W = params[2 * k].detach()
            wi = 2 * k
            inp = (ctx.x3 if k == 0 else ps[k - 1]).view(-1, (ctx.x3 if k == 0 else ps[k - 1]).shape[-1])
            dh2 = hip.aggregate(graph, dp, None, transpose=True).view(B * n, -1)
            if k == L - 1 and Fp != Fo:
                dh2 = dh2[:, :Fo]  # [rows, Fout] view with row stride Fp
            dW = G.dst[wi] if G.dst[wi] is not None else torch.zeros_like(params[wi])
            if k > 0:
                # one fused launch: dp_{k-1} (with PReLU'), dW_k, d(slope) and the bias gradient of
                # conv k-1 (= column sums of dp_{k-1}), which accumulates by ITS parameter's state
                dp = hip.linear_bwd_all(dh2, W, inp, slope_t, dsl, dW, None, G.dst[2 * k - 1], G.acc[wi],
                                        acc_colsum=G.acc[2 * k - 1], act=akind).view(B, n, -1)
            elif ctx.lat is not None:
                dx = _lat_first_layer_bwd(ctx.lat, ctx.x3, dh2.reshape(B, n, -1), W, dW, G.acc[wi], ctx.needs_input_grad[0])
            elif ctx.needs_input_grad[0]:
                dx = hip.linear_bwd_all(dh2, W, inp, None, None, dW, None, None, G.acc[wi],
                                        act=hip.ACT_NONE).view(B, n, -1)
            else:
                hip.linear_bwd_dw(dh2, inp, None, dW, None, G.acc[wi])
        if dx is not None and ctx.squeeze:
            dx = dx[0]
        return (dx, None, None, None, None, None, None) + G.out()


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
        act = getattr(owner, "_in_act", None)  # None: PReLU when a slope is given
        lat = getattr(owner, "_lat_src", None)
        ctx.lat = ctx.tab = None
        if lat is not None:
            # x is the encoder output [B, ne, D] (LatSource): only its compact mesh rows are transformed, the attention
            # kernels read the transformed rows through the table - no [B, M, D] latents, no [B, M, H*C] transform
            owner._lat_src = None
            assert sl is None, "a LatSource feeds the first layer of a stack (no input activation)"
            if not x3.is_contiguous():
                x3 = x3.contiguous()
            ne, D = x3.shape[1], x3.shape[2]
            nc = lat.Md + lat.r
            Pc = hip.copy_rows(x3[:, lat.G:, :], torch.empty(B, nc, D, dtype=torch.float32, device=x3.device))
            h = hip.linear_fwd(Pc.view(B * nc, D), W.detach(), None, None, act=act).view(B, nc, H * Cc)
            ctx.lat, ctx.tab, ctx.enc_shape = lat, lat.compact_tab(ne), x3.shape
            x3 = Pc
            if lat.landing is not None:
                lat.landing.mesh_pending = True
        else:
            h = hip.linear_fwd(x3.view(B * n, -1), W.detach(), None, sl, act=act).view(B, n, H * Cc)
        ctx.act = act
        y, a_s, a_d, alpha = hip.gat_fwd(graph, h, att_src.detach().reshape(-1), att_dst.detach().reshape(-1),
                                         bias.detach(), H, Cc, tab=ctx.tab)
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
                             ctx.a_d, ctx.alpha, t_as.view(-1), t_ad.view(-1), t_b, False, H, Cc, tab=ctx.tab)
            for dst, t, a in ((G.dst[2], t_as, G.acc[2]), (G.dst[3], t_ad, G.acc[3]), (G.dst[4], t_b, G.acc[4])):
                if dst is not None:
                    dst.add_(t) if a else dst.copy_(t)
        else:
            dh = hip.gat_bwd(graph, dy3, ctx.h, att_src.detach().reshape(-1), att_dst.detach().reshape(-1), ctx.a_s,
                             ctx.a_d, ctx.alpha, d_as.view(-1), d_ad.view(-1), G.dst[4], True, H, Cc, tab=ctx.tab)
        if ctx.lat is not None:
            # fold dh [B, M, H*C] back onto the compact rows and run the dense backward there (LatSource)
            dW = G.dst[1] if G.dst[1] is not None else torch.zeros_like(W)
            dx = _lat_first_layer_bwd(ctx.lat, None, dh, W.detach(), dW, G.acc[1], ctx.needs_input_grad[0], Pc=ctx.x3,
                                      enc_shape=tuple(ctx.enc_shape))
            return (dx, None, None, None, None) + G.out()
        dh2 = dh.view(B * n, -1)
        inp = ctx.x3.view(B * n, -1)
        sl = slope.detach() if slope is not None else None
        dx = None
        if G.dst[1] is not None and ctx.needs_input_grad[0]:
            # one launch for dx (with the activation's derivative), dW and the slope gradient
            dx = hip.linear_bwd_all(dh2, W.detach(), inp, sl, G.dst[0] if sl is not None else None, G.dst[1], None, None,
                                    G.acc[1], act=ctx.act)
        else:
            if G.dst[1] is not None:
                hip.linear_bwd_dw(dh2, inp, sl, G.dst[1], None, G.acc[1], act=ctx.act)
            if ctx.needs_input_grad[0]:
                dx = hip.linear_bwd_dx(dh2, W.detach(), inp, sl, G.dst[0] if sl is not None else None, act=ctx.act)
        if dx is not None:
            dx = dx.view(B, n, -1)
            if ctx.squeeze:
                dx = dx[0]
        return (dx, None, None, None, None) + G.out()


class LayerNormFn(torch.autograd.Function):
    """Node-mode LayerNorm.  When the owner carries a GradLanding (`_grad_src`: the final LayerNorm of a GAT processor
    inside WeatherPrediction.forward) it takes part in the same two channels as GCNStackFn's fused LayerNorm: the
    output rows the decoder reads go straight into the decoder's input (dec_buf), and the backward reads its gradient
    through the decoder-input gather's row map instead of a zero-filled dense tensor."""

    @staticmethod
    def forward(ctx, x, owner, eps, gamma, beta):
        x2 = hip.rows2d(x.detach())
        land = getattr(owner, "_grad_src", None)
        ctx.land = None
        if land is not None:
            owner._grad_src = None
            if x.dim() == 3:
                ctx.land = land
                land.proc_ready = True
        ctx.params = (gamma, beta)
        if ctx.land is not None and land.dec_buf is not None and land.dec_buf.shape[2] == x.shape[-1]:
            B, n, F = x.shape
            stats = hip.layernorm_fwd_map(x2, gamma.detach(), beta.detach(), eps, land.dec_buf, land.dec_map)
            land.dec_filled = True
            ctx.x2, ctx.stats = x2, stats
            return x2.new_zeros(()).expand(B, n, F)
        y, stats = hip.layernorm_fwd(x2, gamma.detach(), beta.detach(), eps)
        ctx.x2, ctx.stats = x2, stats
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        gamma, beta = ctx.params
        G = _Grads([gamma, beta], list(ctx.needs_input_grad[3:]))
        dg = G.dst[0] if G.dst[0] is not None else torch.zeros_like(gamma)
        db = G.dst[1] if G.dst[1] is not None else torch.zeros_like(beta)
        land, dy_map = ctx.land, None
        if land is not None and land.proc_src is not None:
            if all(st == 0 for st in dy.stride()):
                dy_map = (land.proc_src, land.proc_map)  # the token: the gradient is the gather's, through its row map
            else:  # someone else also sent gradient here: materialise the gather's part and add (see GCNStackFn.backward)
                src, pmap = land.proc_src, land.proc_map
                dy = _flat3(dy) + hip.gather2_rows(src, pmap, None, None, dy.shape[-2], src.shape[0])
            land.proc_src = land.proc_map = None
            land.proc_ready = False
        dx = hip.layernorm_bwd(None if dy_map is not None else hip.rows2d(dy), ctx.x2, gamma.detach(), ctx.stats, dg, db,
                               G.acc[0] and G.acc[1], dy_map=dy_map)
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
    def forward(ctx, x, grid_static, mesh_static, tail3=None):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        ctx.squeeze, ctx.G, ctx.Cdyn = squeeze, x3.shape[1], x3.shape[2]
        out = hip.assemble_input(x3, grid_static, mesh_static, tail3)
        return out[0] if squeeze else out

    @staticmethod
    def backward(ctx, dy):
        dy3 = dy if dy.dim() == 3 else dy.unsqueeze(0)
        dx = dy3[:, : ctx.G, : ctx.Cdyn].contiguous()
        return (dx[0] if ctx.squeeze else dx), None, None, None


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
        d = ctx.dd * g
        # pred = x_last + delta: in an autoregressive rollout x_last is the previous step's prediction and
        # carries the same gradient as delta (src/train.py:202-204)
        return d, (d if ctx.needs_input_grad[1] else None), None, None, None, None


class ARStepLossFn(torch.autograd.Function):
    """One autoregressive step of the training loop (src/train.py:203-228) as ONE differentiable unit:
    loss_out = loss_prev + weighted MSE(residual ? x_last + delta : delta, y_step) and, when `advance`, the window
    rolled forward (static channels carried, forcing channels from y).  Forward = gcl_wmse_fwd_bwd + gcl_ar_advance,
    backward = ONE gcl_ar_step_bwd pass producing d_delta and d_state (the upstream loss gradient is read on the
    device): no torch arithmetic, clone, per-channel writes or cat on the differentiated path."""

    @staticmethod
    def forward(ctx, state4, delta3, y3, loss_prev, node_w, chan_w, inv_wsum: float, kinds, residual: bool,
                advance: bool):
        st = state4.detach()
        if not st.is_contiguous():
            st = st.contiguous()
        B, G, obs, Cc = st.shape
        d3 = delta3.detach()
        if d3.stride(2) != 1:
            d3 = d3.contiguous()
        x_last = st[:, :, obs - 1, :] if residual else None
        loss, dd, _ = hip.wmse_fwd_bwd(d3, x_last, y3, node_w, chan_w, inv_wsum, 1.0, want_grad=True,
                                       loss_prev=loss_prev.detach() if loss_prev is not None else None)
        new_state = hip.ar_advance(st, d3, y3, kinds, None, 0, residual) if advance else st.new_empty(0)
        ctx.dd, ctx.kinds, ctx.residual, ctx.advance, ctx.obs, ctx.has_y = dd, kinds, residual, advance, obs, y3 is not None
        if not advance:
            ctx.mark_non_differentiable(new_state)
        return loss, new_state

    @staticmethod
    def backward(ctx, g_loss, g_new):
        g_new = g_new if (ctx.advance and g_new is not None and g_new.numel()) else None
        d_delta, d_state = hip.ar_step_bwd(ctx.dd, g_loss, g_new, ctx.kinds, ctx.has_y, ctx.residual, ctx.obs,
                                           want_state=ctx.needs_input_grad[0])
        return d_state, d_delta, None, (g_loss if ctx.needs_input_grad[3] else None), None, None, None, None, None, None


class Gather2Fn(torch.autograd.Function):
    """Row gather from two sources with precomputed index maps (see hip.gather2_rows).
    maps = (map_a, map_b, inv_a, inv_b): forward maps have length nd; inv_x[j] = destination row
    fed by row j of source x, or -1.  A source with leading dimension 1 is broadcast over the
    batch and receives the batch-summed gradient."""

    @staticmethod
    def forward(ctx, a, b, maps, nd: int, B: int, landing=None):
        map_a, map_b, inv_a, inv_b = maps
        a3 = a.detach()
        if not a3.is_contiguous():
            a3 = a3.contiguous()
        ctx.maps, ctx.B, ctx.landing = maps, B, landing
        ctx.sa, ctx.sb = a3.shape, (b.shape if b is not None else None)
        if landing is not None and landing.dec_filled:
            # the rows of source b are already in place (written by the producer's LayerNorm through the row map, b itself
            # is a stride-0 token): only the head rows of source a are copied
            landing.dec_filled = False
            buf, landing.dec_buf = landing.dec_buf, None
            hip.copy_rows(a3[:, : landing.head, :], buf[:, : landing.head, :])
            return buf
        if landing is not None:
            landing.dec_buf = None
        b3 = b.detach() if b is not None else None
        if b3 is not None and not b3.is_contiguous():
            b3 = b3.contiguous()
        return hip.gather2_rows(a3, map_a, b3, map_b, nd, B)

    @staticmethod
    def backward(ctx, g):
        map_a, map_b, inv_a, inv_b = ctx.maps
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            bc = ctx.sa[0] == 1 and ctx.B > 1
            land = ctx.landing
            if land is not None and land.mesh_pending and not bc:
                # GradLanding: this call fills the head rows of the shared gradient buffer and hands the WHOLE buffer to
                # autograd; MeshLatFn.backward (always later: it needs the processor's backward, which needs db below)
                # fills the tail rows in place and returns None - no zero-filled halves, no add of two full tensors
                land.buf = torch.empty(ctx.sa, dtype=torch.float32, device=g.device)
                hip.gather2_rows(g, inv_a, None, None, land.head, ctx.B, out=land.buf[:, : land.head])
                da = land.buf
                land.handed_over = True  # MeshLatFn.backward must run and fill the tail rows (checked there / by callers)
            else:
                da = hip.gather2_rows(g, inv_a, None, None, ctx.sa[1], ctx.B, sum_batch=bc)
        if ctx.sb is not None and ctx.needs_input_grad[1]:
            bc = ctx.sb[0] == 1 and ctx.B > 1
            land = ctx.landing
            if land is not None and land.proc_ready and not bc:
                land.proc_src, land.proc_map = g, inv_b
                db = g.new_zeros(()).expand(ctx.sb)  # a stride-0 token: the consumer reads land.proc_src instead
            else:
                db = hip.gather2_rows(g, inv_b, None, None, ctx.sb[1], ctx.B, sum_batch=bc)
        return da, db, None, None, None, None


class GradLanding:
    """One gradient buffer shared by the two consumers of the compact encoder output [B, ne, D]: the decoder-input
    gather only ever sends gradient to the first `head` (grid) rows, the mesh-latent gather only to the rest.  See
    Gather2Fn.backward / MeshLatFn.backward; used only when nothing else consumes the encoder output with a gradient
    (WeatherPrediction.forward, not forward_with_latents)."""

    def __init__(self, head: int):
        self.head, self.buf, self.mesh_pending = head, None, False
        self.handed_over = False  # True between Gather2Fn.backward handing the buffer out and MeshLatFn.backward filling its tail
        # second channel: the processor's output is only consumed by the decoder-input gather, so its gradient is the
        # gather's incoming gradient seen through a row map - the processor's LayerNorm backward reads it that way
        # (gcl_layernorm_bwd_map) instead of a zero-filled dense [B, M, D] tensor
        self.proc_ready, self.proc_src, self.proc_map = False, None, None
        # third channel (forward): the decoder-input buffer [B, head + U, D], allocated by the model before the processor
        # runs, and the map mesh row -> row of that buffer (or -1).  The processor's final LayerNorm writes the rows
        # the decoder reads straight into it (gcl_layernorm_fwd_map) and the gather then only copies the head rows.
        self.dec_buf, self.dec_map, self.dec_filled = None, None, False


class MeshLatFn(torch.autograd.Function):
    """Mesh latents [B, M, D] of the compact pipeline from ONE encoder output [B, n_e, D] whose rows per sample are
    [G grid | Md batch-dependent mesh | r folded batch-invariant mesh rows]: the Mi batch-invariant mesh rows are
    dealt r = ceil(Mi / B) to a sample as isolated nodes, so they ride through the encoder's launches instead of a
    second (B = 1) pass of small launches.  Mesh row i reads enc[b, map_a[i]] (dependent rows) or the flat row
    map_b[i] of enc viewed as [1, B * n_e, D] (folded rows, shared by all samples).
    Backward: rows < G + Md gather g[b, inv_a[j]] (grid rows: 0), folded row j of the flat list gets the batch sum
    of g[:, inv_fold[j]]."""

    @staticmethod
    def forward(ctx, enc, maps, M: int, gmd: int, r: int, landing=None):
        map_a, map_b, inv_a, inv_fold = maps
        e3 = enc.detach()
        if not e3.is_contiguous():
            e3 = e3.contiguous()
        B, ne, D = e3.shape
        ctx.maps, ctx.shape, ctx.gmd, ctx.r, ctx.landing = maps, (B, ne, D), gmd, r, landing
        if landing is not None:
            landing.mesh_pending = True
        return hip.gather2_rows(e3, map_a, e3.view(1, B * ne, D), map_b, M, B)

    @staticmethod
    def backward(ctx, g):
        _, _, inv_a, inv_fold = ctx.maps
        B, ne, D = ctx.shape
        g = g.contiguous()
        land = ctx.landing
        shared = land is not None and land.buf is not None
        out = land.buf if shared else torch.empty(B, ne, D, dtype=torch.float32, device=g.device)
        h = land.head if shared else 0  # rows below `head` were written by the decoder-input gather's backward
        hip.gather2_rows(g, inv_a[h:], None, None, ctx.gmd - h, B, out=out[:, h: ctx.gmd])
        if ctx.r > 1:  # the batch sums land r to a sample, straight in the tail rows
            hip.gather2_rows(g, inv_fold, None, None, B * ctx.r, B, sum_batch=True, out=out[:, ctx.gmd:], deal=ctx.r)
        elif ctx.r == 1:
            tmp = hip.gather2_rows(g, inv_fold, None, None, B * ctx.r, B, sum_batch=True)
            hip.copy_rows(tmp.view(B, ctx.r, D), out[:, ctx.gmd:])
        if shared:
            land.buf, land.mesh_pending, land.handed_over = None, False, False
            return None, None, None, None, None, None
        return out, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# InteractionNet processor (src/models.py:166-285): edge encoder + N unshared message-passing steps.
# The first edge-MLP layer on cat([x_s, x_r, e]) is split by operand,
#     hidden = e We^T + (x Ws^T)[senders] + (x Wr^T)[receivers] + b,     [Ws | Wr | We] = edge_mlp[0].weight
# so the per-edge contraction is D wide instead of 3D and the node projections are done once per
# node; same for cat([x, agg]) of the node MLP.  Edges are kept receiver-sorted internally
# (EdgeLayout): the scatter-mean is then a contiguous segment mean and nothing needs atomics.
# params layout: [enc_W, enc_b, slope_enc] + per step [We1, be1, We2, be2, Wn1, bn1, Wn2, bn2, slope,
#                 gamma_e, beta_e, gamma_n, beta_n]   (slope* only for PReLU, gamma/beta only with LN; None otherwise)
# ------------------------------------------------------------------------------------------------
class EdgeLayout:
    """Receiver-sorted view of a reference-layout edge_index [2, E] (+ its sender-sorted transpose)."""

    def __init__(self, edge_index: torch.Tensor, n: int, device):
        ei = edge_index.detach().to("cpu", torch.int64)
        snd, rcv = ei[0], ei[1]
        if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= n):
            raise ValueError("edge_index refers to nodes outside [0, n)")
        order = torch.sort(rcv, stable=True).indices            # sorted position -> reference edge id
        self.order = order.to(device)
        snd_s, rcv_s = snd[order], rcv[order]
        cnt = torch.bincount(rcv_s, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(cnt, 0)
        torder = torch.sort(snd_s, stable=True).indices        # sender-grouped list of sorted positions
        tcnt = torch.bincount(snd_s, minlength=n)
        trowptr = torch.zeros(n + 1, dtype=torch.int64)
        trowptr[1:] = torch.cumsum(tcnt, 0)
        i32 = lambda t: t.to(torch.int32).to(device)
        self.n, self.E = n, int(ei.shape[1])
        self.snd, self.rcv, self.rowptr = i32(snd_s), i32(rcv_s), i32(rowptr)
        self.tperm, self.trowptr = i32(torder), i32(trowptr)
        self.invdeg = (1.0 / cnt.clamp(min=1).to(torch.float32)).to(device)


class InteractionNetFn(torch.autograd.Function):
    PER_STEP = 13

    @staticmethod
    def forward(ctx, x, owner, lay: EdgeLayout, raw_edges, n_steps: int, act: int, use_ln: bool, eps: float, *params):
        squeeze = x.dim() == 2
        x3 = _flat3(x.detach())
        B, n, D = x3.shape
        E = lay.E
        P = [p.detach() if p is not None else None for p in params]
        # ReLU (src/models.py:154-163) runs as a PReLU with the owner's constant zero slope (no slope gradient)
        cs = getattr(owner, "const_slope", None)
        slope_enc = P[2] if P[2] is not None else cs
        e0pre = hip.dense_fwd(raw_edges, P[0], P[1])                       # [E, D]  batch-invariant
        e0 = hip.act_fwd(e0pre, act, slope_enc)
        e = e0.unsqueeze(0) if B == 1 else e0.unsqueeze(0).expand(B, E, D).contiguous()
        saved = []
        xc = x3
        for k in range(n_steps):
            We1, be1, We2, be2, Wn1, bn1, Wn2, bn2, slope, ge, bte, gn, btn = P[3 + 13 * k: 16 + 13 * k]
            slope = slope if slope is not None else cs
            last = k == n_steps - 1
            x2, e2 = xc.view(B * n, D), e.view(B * E, D)
            # edge update
            PS = torch.empty(B, n, 2 * D, dtype=torch.float32, device=x3.device)
            PS2 = PS.view(B * n, 2 * D)
            hip.dense_fwd(x2, We1[:, :D], None, out=PS2[:, :D])
            hip.dense_fwd(x2, We1[:, D:2 * D], None, out=PS2[:, D:])
            H = hip.dense_fwd(e2, We1[:, 2 * D:], be1).view(B, E, D)
            hip.edge_combine(H, None, PS[:, :, :D], lay.snd, None, PS[:, :, D:], lay.rcv, out3=H)
            U = hip.dense_fwd(H.view(B * E, D), We2, be2, act, slope).view(B, E, D)
            agg = hip.segment_reduce(U, None, lay.rowptr, True)
            epre = estats = None
            if not last:  # the edge state after the last step is never read (src/models.py:282-285)
                epre = hip.edge_combine(e, U, None, None, None, None, None, out3=U)  # e + U, in place of U
                if use_ln:
                    e_next, estats = hip.graphnorm_fwd(epre, ge, bte, eps)
                else:
                    e_next = epre
            # node update
            Hn = hip.dense_fwd(x2, Wn1[:, :D], bn1)
            hip.dense_fwd(agg.view(B * n, D), Wn1[:, D:], None, addend=Hn, out=Hn)
            xpre = hip.dense_fwd(Hn, Wn2, bn2, act, slope, addend=x2)
            xstats = None
            if use_ln:
                xn, xstats = hip.layernorm_fwd(xpre, gn, btn, eps)
            else:
                xn = xpre
            saved.append((xc, e, H, agg, epre, estats, Hn, xpre, xstats))
            xc = xn.view(B, n, D)
            if not last:
                e = e_next
        ctx.owner, ctx.lay, ctx.raw, ctx.n_steps, ctx.act, ctx.use_ln, ctx.eps = owner, lay, raw_edges, n_steps, act, use_ln, eps
        ctx.params, ctx.saved, ctx.e0pre, ctx.squeeze, ctx.dims, ctx.cs = params, saved, e0pre, squeeze, (B, n, D, E), cs
        return xc[0] if squeeze else xc

    @staticmethod
    def backward(ctx, dy):
        params, lay, act, use_ln, eps = ctx.params, ctx.lay, ctx.act, ctx.use_ln, ctx.eps
        B, n, D, E = ctx.dims
        needs = list(ctx.needs_input_grad[8:])
        G = _Grads(list(params), needs)
        P = [p.detach() if p is not None else None for p in params]

        def dst(i):  # gradient destination of parameter i (a scratch tensor when not wanted)
            return G.dst[i] if G.dst[i] is not None else (torch.zeros_like(P[i]) if P[i] is not None else None)

        dx = _flat3(dy).view(B * n, D)
        de = None  # gradient wrt the edge state flowing into the step above
        for k in range(ctx.n_steps - 1, -1, -1):
            o = 3 + 13 * k
            We1, be1, We2, be2, Wn1, bn1, Wn2, bn2, slope, ge, bte, gn, btn = P[o: o + 13]
            slope = slope if slope is not None else ctx.cs
            xc, e, H, agg, epre, estats, Hn, xpre, xstats = ctx.saved[k]
            ctx.saved[k] = None
            x2, e2, H2 = xc.view(B * n, D), e.view(B * E, D), H.view(B * E, D)
            dsl = dst(o + 8)
            # node side
            if use_ln:
                dxpre = hip.layernorm_bwd(dx, xpre, gn, xstats, dst(o + 11), dst(o + 12), G.acc[o + 11] and G.acc[o + 12])
            else:
                dxpre = dx
            dHn = hip.dense_bwd_dx(dxpre, Wn2, Hn, act, slope, dsl)
            hip.dense_bwd_dw(dxpre, Hn, dst(o + 6), dst(o + 7), G.acc[o + 6], act, slope)
            dxa = hip.dense_bwd_dx(dHn, Wn1[:, :D], addend=dxpre)
            dagg = hip.dense_bwd_dx(dHn, Wn1[:, D:])
            dWn1 = dst(o + 4)
            hip.dense_bwd_dw(dHn, x2, dWn1[:, :D], dst(o + 5), G.acc[o + 4])
            hip.dense_bwd_dw(dHn, agg.view(B * n, D), dWn1[:, D:], None, G.acc[o + 4])
            # edge side
            depre = None
            if de is not None:
                depre = hip.graphnorm_bwd(de, epre, ge, estats, dst(o + 9), dst(o + 10), G.acc[o + 9] and G.acc[o + 10],
                                          eps) if use_ln else de
            dU = hip.edge_combine(depre, None, dagg.view(B, n, D), lay.rcv, lay.invdeg, None, None)
            dU2 = dU.view(B * E, D)
            dH = hip.dense_bwd_dx(dU2, We2, H2, act, slope, dsl)
            hip.dense_bwd_dw(dU2, H2, dst(o + 2), dst(o + 3), G.acc[o + 2], act, slope)
            dWe1 = dst(o)
            de_in = hip.dense_bwd_dx(dH, We1[:, 2 * D:], addend=depre.view(B * E, D) if depre is not None else None)
            hip.dense_bwd_dw(dH, e2, dWe1[:, 2 * D:], dst(o + 1), G.acc[o])
            dPS = torch.empty(B, n, 2 * D, dtype=torch.float32, device=dH.device)
            dH3 = dH.view(B, E, D)
            hip.segment_reduce(dH3, lay.tperm, lay.trowptr, False, out3=dPS[:, :, :D])
            hip.segment_reduce(dH3, None, lay.rowptr, False, out3=dPS[:, :, D:])
            dPS2 = dPS.view(B * n, 2 * D)
            t = hip.dense_bwd_dx(dPS2[:, :D], We1[:, :D], addend=dxa)
            dx = hip.dense_bwd_dx(dPS2[:, D:], We1[:, D:2 * D], addend=t, out=t)
            hip.dense_bwd_dw(dPS2[:, :D], x2, dWe1[:, :D], None, G.acc[o])
            hip.dense_bwd_dw(dPS2[:, D:], x2, dWe1[:, D:2 * D], None, G.acc[o])
            de = de_in.view(B, E, D)
        # edge encoder (batch-invariant: its gradient is the sum over samples)
        de0 = de[0] if B == 1 else de.sum(dim=0)
        de0pre = hip.act_bwd(ctx.e0pre, de0.contiguous(), act, P[2] if P[2] is not None else ctx.cs, dst(2))
        hip.dense_bwd_dw(de0pre, ctx.raw, dst(0), dst(1), G.acc[0])
        gx = None
        if ctx.needs_input_grad[0]:
            gx = dx.view(B, n, D)
            gx = gx[0] if ctx.squeeze else gx
        return (gx, None, None, None, None, None, None, None) + G.out()
