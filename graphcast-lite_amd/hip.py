"""ctypes binding of `libgcl_hip.so` (C ABI: `include/gcl.h`).

PyTorch tensors are containers only: every wrapper hands raw device pointers, leading dimensions
and the current HIP stream to the library.  There is NO CPU fallback - if the shared library is
missing or a tensor is not on a GPU the call raises.
"""
import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCL_LIB") or os.path.join(_HERE, "libgcl_hip.so")  # GCL_LIB: A/B builds while tuning

GRAPH_GCN, GRAPH_GAT, GRAPH_MEAN = 0, 1, 2

_lib = None

_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/gcl.h
_SIGNATURES = {
    "gcl_version": (C.c_int, []),
    "gcl_last_error": (C.c_char_p, []),
    "gcl_graph_count_edges": (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(_i64)]),
    "gcl_graph_build_host": (C.c_int, [_vp, _i64, _i32, _i32] + [_vp] * 8),
    "gcl_graph_create": (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(_vp)]),
    "gcl_graph_destroy": (None, [_vp]),
    "gcl_graph_num_nodes": (_i32, [_vp]),
    "gcl_graph_num_edges": (_i64, [_vp]),
    "gcl_graph_max_in_degree": (_i32, [_vp]),
    "gcl_graph_export_edges": (C.c_int, [_vp, _vp]),
    "gcl_graph_eperm_device": (_vp, [_vp]),
    "gcl_graph_halo_info": (C.c_int, [_vp, _i32, _i32, _vp]),
    "gcl_linear_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "gcl_linear_bwd_dx": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _sz, _vp]),
    "gcl_linear_bwd_dw": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_linear_bwd_ws_bytes": (_sz, [_i64, _i32, _i32]),
    "gcl_linear_bwd_all": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32,
                                     _i32, _vp, _sz, _vp]),
    "gcl_linear_bwd_all_deferred": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32,
                                              _i32, _i32, _vp, _sz, _vp, _vp]),
    "gcl_reduce_jobs": (C.c_int, [_vp, _i32, _vp]),
    "gcl_linear_bwd_all_ws_bytes": (_sz, [_i64, _i32, _i32]),
    "gcl_aggregate": (C.c_int, [_vp, _i32, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "gcl_gat_fwd": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "gcl_gat_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                              _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_gat_bwd_ws_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "gcl_gat_fwd_tab": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "gcl_gat_bwd_tab": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                  _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_gat_tab_ok": (C.c_int, [_vp, _i64, _i32, _i32]),
    "gcl_gat_alpha_to_edge_order": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "gcl_gat_prune": (C.c_int, [_vp, _vp, _f32, _vp, C.POINTER(_i64), _vp, _sz, _vp]),
    "gcl_gat_prune_ws_bytes": (_sz, [_i64]),
    "gcl_layernorm_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _vp, _i64, _vp, _i64, _i32, _vp]),
    "gcl_layernorm_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i64, _i32, _vp, _sz, _vp]),
    "gcl_layernorm_bwd_cs": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _sz, _vp]),
    "gcl_layernorm_bwd_map": (C.c_int, [_vp, _i64, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i64,
                                        _i32, _vp, _sz, _vp]),
    "gcl_layernorm_bwd_ws_bytes": (_sz, [_i64, _i32]),
    "gcl_graphnorm_fwd": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _f32, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_graphnorm_bwd": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _f32, _vp, _i64, _i64, _vp, _vp,
                                    _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_graphnorm_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "gcl_colsum": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "gcl_colsum_ws_bytes": (_sz, [_i64, _i32]),
    "gcl_assemble_input": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gcl_assemble_input_tail": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gcl_wmse_fwd_bwd": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _f32, _f32, _vp,
                                   _vp, _vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "gcl_ar_step_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gcl_pad_rows": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "gcl_zero": (C.c_int, [_vp, _sz, _vp]),
    "gcl_wmse_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "gcl_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "gcl_adam_step_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp]),
    "gcl_copy_rows": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "gcl_dense_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "gcl_dense_bwd_dx": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32,
                                   _vp, C.c_size_t, _vp]),
    "gcl_dense_bwd_dw": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp,
                                   C.c_size_t, _vp]),
    "gcl_gcn_layer_fwd": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "gcl_gcn_layer_fwd_rows": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gcl_layernorm_fwd_map": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _i32, _vp]),
    "gcl_gcn_layer_fwd_tab": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "gcl_gcn_layer_fwd_tab_ok": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _i32, _i32]),
    "gcl_segment_reduce": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "gcl_edge_combine": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i32, _i64, _i32, _vp]),
    "gcl_act_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "gcl_act_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "gcl_act_bwd_ws_bytes": (C.c_size_t, []),
    "gcl_window_pack": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "gcl_ar_advance": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32,
                                 _i32, _i32, _vp]),
    "gcl_gather2_rows": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """The loaded shared library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C graphcast-lite_amd/csrc`). The HIP path has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libgcl_hip error {rc}: {lib().gcl_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libgcl_hip needs GPU tensors (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"libgcl_hip needs float32 tensors, got {t.dtype}")
    return t.data_ptr()


_ws = {}
_ws_retired = []  # outgrown buffers stay allocated: kernels inside a captured hipGraph may still point at them


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-device scratch buffer (uint8).  Grows geometrically and never frees the buffers
    it outgrows (a replayed hipGraph keeps using the one that was current at capture time)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    cur = _ws.get(key)
    if cur is None or cur.numel() < nbytes:
        size = max(int(nbytes), 1 << 20, 2 * cur.numel() if cur is not None else 0)
        if cur is not None:
            _ws_retired.append(cur)
        cur = torch.empty(size, dtype=torch.uint8, device=f"cuda:{key}")
        _ws[key] = cur
    return cur


def rows2d(t: torch.Tensor) -> torch.Tensor:
    """View `[..., F]` as `[rows, F]` with a single row stride; copies only if it must."""
    if t.dim() == 2 and t.stride(1) == 1:
        return t
    if not t.is_contiguous():
        t = t.contiguous()
    return t.view(-1, t.shape[-1])


class Graph:
    """Device CSR (+transpose) of a reference-layout `edge_index` (int64 `[2,E]`, CPU or GPU)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, kind: int):
        ei = edge_index.detach().to("cpu", torch.int64).contiguous()
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(ei.shape)}")
        self.kind = kind
        self._h = _vp()
        _check(lib().gcl_graph_create(ei.data_ptr(), ei.shape[1], int(num_nodes), kind, C.byref(self._h)))
        self.n = int(num_nodes)
        self.e = int(lib().gcl_graph_num_edges(self._h))
        self.max_in_degree = int(lib().gcl_graph_max_in_degree(self._h))
        # PyG-order edge list with loops, as a tensor with a STABLE identity: when the input already
        # is that list (SparseGATConv feeds its own output back, src/models.py:846) the input tensor
        # itself is returned, so the CSR cache keeps hitting instead of rebuilding every step.
        self._exported = {}
        if ei.shape[1] == self.e and torch.equal(self.export_edges(), ei):
            self._exported[str(edge_index.device)] = edge_index

    @property
    def handle(self):
        return self._h

    def export_edges(self) -> torch.Tensor:
        out = torch.empty(2, self.e, dtype=torch.int64)
        _check(lib().gcl_graph_export_edges(self._h, out.data_ptr()))
        return out

    def halo_info(self, transpose: bool = False, T: int = 64):
        """(T, tiles, staged-source stride) of the source-tile layout, or None when it was not built."""
        out = (C.c_int32 * 4)()
        _check(lib().gcl_graph_halo_info(self._h, 1 if transpose else 0, T, out))
        return (out[0], out[1], out[2]) if out[0] else None

    def edges_with_loops(self, device) -> torch.Tensor:
        key = str(torch.device(device))
        t = self._exported.get(key)
        if t is None:
            t = self.export_edges().to(device)
            self._exported[key] = t
        return t

    def __del__(self):
        try:
            if self._h:
                lib().gcl_graph_destroy(self._h)
                self._h = _vp()
        except Exception:
            pass


def build_csr_host(edge_index: torch.Tensor, num_nodes: int, kind: int):
    """CPU-only CSR construction (no GPU needed) - returns a dict of torch CPU tensors."""
    ei = edge_index.detach().to("cpu", torch.int64).contiguous()
    e_out = _i64(0)
    _check(lib().gcl_graph_count_edges(ei.data_ptr(), ei.shape[1], int(num_nodes), kind, C.byref(e_out)))
    Ep, n = e_out.value, int(num_nodes)
    i32 = lambda k: torch.zeros(max(k, 1), dtype=torch.int32)
    f32 = lambda k: torch.zeros(max(k, 1), dtype=torch.float32)
    out = dict(rowptr=i32(n + 1), col=i32(Ep), w=f32(Ep), eperm=i32(Ep), trowptr=i32(n + 1), tcol=i32(Ep),
               tw=f32(Ep), tslot=i32(Ep))
    _check(lib().gcl_graph_build_host(
        ei.data_ptr(), ei.shape[1], n, kind, out["rowptr"].data_ptr(), out["col"].data_ptr(), out["w"].data_ptr(),
        out["eperm"].data_ptr(), out["trowptr"].data_ptr(), out["tcol"].data_ptr(), out["tw"].data_ptr(),
        out["tslot"].data_ptr()))
    for k in ("col", "w", "eperm", "tcol", "tw", "tslot"):
        out[k] = out[k][:Ep]
    out["num_edges"] = Ep
    return out


# ------------------------------------------------------------------------------------------------
# raw wrappers (no autograd).  2-D tensors are [rows, F] with stride (ld, 1).
# ------------------------------------------------------------------------------------------------
def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a [rows, F] view with unit channel stride"
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _act_of(act, slope):
    return (ACT_PRELU if slope is not None else ACT_NONE) if act is None else int(act)


def linear_fwd(x, W, bias, in_slope, out=None, ld_out=None, act=None):
    """y = act(x) W^T + bias; act None: PReLU when in_slope is given, else identity."""
    rows, Fin = x.shape
    Fout = W.shape[0]
    if out is None:
        ld = ld_out or Fout
        out = torch.empty(rows, ld, dtype=torch.float32, device=x.device)[:, :Fout]
    if rows == 0:
        return out
    _check(lib().gcl_dense_fwd(_p(x), _ld(x), _act_of(act, in_slope), _p(in_slope), _p(W), W.stride(0), _p(bias), None, 0,
                               _p(out), _ld(out), rows, Fin, Fout, _stream()))
    return out


def linear_bwd_dx(dy, W, x, in_slope, d_in_slope, act=None):
    a = _act_of(act, in_slope)
    return dense_bwd_dx(dy, W, x if a != ACT_NONE else None, a, in_slope, d_in_slope)


def linear_bwd_dw(dy, x, in_slope, dW, db, accumulate: bool, act=None):
    dense_bwd_dw(dy, x, dW, db, accumulate, _act_of(act, in_slope), in_slope)


ACT_NONE, ACT_PRELU, ACT_SILU = 0, 1, 2


# Optional launch probe (None in the product): measurement code (bench.py) may install an object with
# `begin(kind, **info) -> token | None` and `end(token)`; the wrappers of the roofline kernels call it around
# their launch so that HIP events can be recorded on the launch stream without any profiling state here.
PROBE = None


def _probe_begin(kind, **info):
    return PROBE.begin(kind, **info) if PROBE is not None else None


def _probe_end(tok):
    if tok is not None:
        PROBE.end(tok)


def dense_fwd(x, W, bias, act=ACT_NONE, slope=None, addend=None, out=None):
    """y = act(x) W^T + bias + addend.  x / W / addend / out may be column blocks (unit channel
    stride, any row stride) of wider tensors."""
    rows, Fin = x.shape
    Fout = W.shape[0]
    assert W.shape[1] == Fin and W.stride(1) == 1 and x.stride(1) == 1
    if out is None:
        out = torch.empty(rows, Fout, dtype=torch.float32, device=x.device)
    if rows == 0:  # an empty batch of rows: nothing to launch (empty tensors have no device pointer)
        return out
    tok = _probe_begin("dense_fwd", rows=rows, Fin=Fin, Fout=Fout)
    _check(lib().gcl_dense_fwd(_p(x), _ld(x), int(act), _p(slope), _p(W), W.stride(0), _p(bias), _p(addend),
                               _ld(addend) if addend is not None else 0, _p(out), _ld(out), rows, Fin, Fout, _stream()))
    _probe_end(tok)
    return out


def dense_bwd_dx(dy, W, z=None, act=ACT_NONE, slope=None, d_slope=None, addend=None, out=None):
    """dx = (dy W) * act'(z) + addend."""
    rows, Fout = dy.shape
    Fin = W.shape[1]
    assert W.shape[0] == Fout and W.stride(1) == 1
    if out is None:
        out = torch.empty(rows, Fin, dtype=torch.float32, device=dy.device)
    if rows == 0:
        return out
    nb = lib().gcl_linear_bwd_ws_bytes(rows, Fin, Fout)
    ws = workspace(nb, dy.device)
    _check(lib().gcl_dense_bwd_dx(_p(dy), _ld(dy), _p(W), W.stride(0), _p(z), _ld(z) if z is not None else 0, int(act),
                                  _p(slope), _p(d_slope), _p(addend), _ld(addend) if addend is not None else 0, _p(out),
                                  _ld(out), rows, Fin, Fout, ws.data_ptr(), ws.numel(), _stream()))
    return out


def dense_bwd_dw(dy, x, dW, db, accumulate: bool, act=ACT_NONE, slope=None):
    """dW (+)= dy^T act(x) (dW may be a column block of a wider gradient), db (+)= colsum(dy)."""
    rows, Fout = dy.shape
    Fin = x.shape[1]
    assert tuple(dW.shape) == (Fout, Fin) and dW.stride(1) == 1
    if rows == 0:  # the sum over no rows: zero unless accumulating
        if not accumulate:
            dW.zero_()
            if db is not None:
                db.zero_()
        return
    nb = lib().gcl_linear_bwd_ws_bytes(rows, Fin, Fout)
    ws = workspace(nb, dy.device)
    _check(lib().gcl_dense_bwd_dw(_p(dy), _ld(dy), _p(x), _ld(x), int(act), _p(slope), _p(dW), dW.stride(0), _p(db), rows,
                                  Fin, Fout, 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()))


ACC_DW, ACC_DB, ACC_COLSUM = 1, 2, 4  # GCL_ACC_* bits of gcl_linear_bwd_all


class _RedSeg(C.Structure):
    _fields_ = [("out", C.c_void_p), ("poff", C.c_int32), ("count", C.c_int32), ("pld", C.c_int32), ("cols", C.c_int32),
                ("ldo", C.c_int32), ("acc", C.c_int32)]


class ReduceJob(C.Structure):
    """Mirror of gcl_reduce_job (include/gcl.h)."""
    _fields_ = [("part", C.c_void_p), ("pstride", C.c_int64), ("seg", _RedSeg * 3), ("spart", C.c_void_p),
                ("sout", C.c_void_p), ("nparts", C.c_int32), ("ns", C.c_int32)]


class _Deferred:
    """Pending final passes of the fused dense backward (gcl_linear_bwd_all_deferred): a training step opens the
    queue before its backward and flushes it once afterwards - one launch per 16 layers instead of one per layer.
    Every pending call owns a workspace from a pool that persists across steps (a replayed hipGraph keeps using it)."""

    def __init__(self):
        self.active = False
        self.jobs, self.dests, self.pool, self.used = [], set(), [], 0
        # the queued job only carries raw pointers: the destination (and slope-gradient) tensors are held here until
        # the flush, so a temporary one (the gradient slot of a frozen parameter) cannot be freed and its block handed
        # to another tensor before the final pass writes it
        self.keep = []

    def ws(self, nbytes, device):
        if self.used == len(self.pool):
            self.pool.append(None)
        cur = self.pool[self.used]
        if cur is None or cur.numel() < nbytes or cur.device != torch.device(device):
            if cur is not None:
                _ws_retired.append(cur)
            cur = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.pool[self.used] = cur
        self.used += 1
        return cur


_deferred = _Deferred()


def defer_begin():
    """Open the deferred-reduction queue (idempotent); pair with defer_flush()."""
    _deferred.active = True


def defer_flush(close: bool = True, drop: bool = False):
    """Run the pending final passes (gcl_reduce_jobs) on the current stream.  `drop` discards them instead (the
    backward that queued them failed).  The queue is emptied whatever happens, so a failed launch cannot leave
    later, unrelated backward calls queueing into a list nobody flushes."""
    d = _deferred
    try:
        if d.jobs and not drop:
            arr = (ReduceJob * len(d.jobs))(*d.jobs)
            _check(lib().gcl_reduce_jobs(C.cast(arr, C.c_void_p), len(d.jobs), _stream()))
    finally:
        d.jobs, d.dests, d.used, d.keep = [], set(), 0, []
        if close:
            d.active = False


def linear_bwd_all(dy, W, x, in_slope, d_in_slope, dW, db, colsum_dx, acc_dW: bool, acc_db=None, acc_colsum=None,
                   act=None):
    """dx (pre-activation gradient) + dW (+ db, slope gradient, column sums of dx) in one call.
    Each destination has its own accumulate flag (acc_db / acc_colsum default to acc_dW): dW, db and
    colsum_dx are gradients of different parameters."""
    acc_db = acc_dW if acc_db is None else acc_db
    acc_colsum = acc_dW if acc_colsum is None else acc_colsum
    a = _act_of(act, in_slope)
    if a == ACT_SILU:  # the fused kernel knows PReLU only
        if db is not None and bool(acc_db) != bool(acc_dW):
            dense_bwd_dw(dy, x, dW, None, acc_dW, a, None)
            colsum(dy, db, acc_db)
        else:
            dense_bwd_dw(dy, x, dW, db, acc_dW, a, None)
        dx = dense_bwd_dx(dy, W, x, a, None, None)
        if colsum_dx is not None:
            colsum(dx, colsum_dx, acc_colsum)
        return dx
    rows, Fout = dy.shape
    Fin = W.shape[1]
    dx = torch.empty(rows, Fin, dtype=torch.float32, device=dy.device)
    nb = lib().gcl_linear_bwd_all_ws_bytes(rows, Fin, Fout)
    acc = (ACC_DW if acc_dW else 0) | (ACC_DB if acc_db else 0) | (ACC_COLSUM if acc_colsum else 0)
    d = _deferred
    if d.active:
        # two pending passes must not write the same destination (a layer that runs twice in one backward, e.g. an
        # autoregressive rollout): flush what is queued first
        dests = {t.data_ptr() for t in (dW, db, colsum_dx) if t is not None}
        if dests & d.dests:
            defer_flush(close=False)
        ws = d.ws(nb, dy.device)
        job = ReduceJob()
        _check(lib().gcl_linear_bwd_all_deferred(
            _p(dy), _ld(dy), _p(W), _p(x), _ld(x), _p(in_slope), _p(d_in_slope), _p(dx), Fin, _p(dW), _p(db),
            _p(colsum_dx), rows, Fin, Fout, acc, ws.data_ptr(), ws.numel(), _stream(), C.byref(job)))
        if job.nparts > 0:
            d.jobs.append(job)
            d.dests |= dests
            d.keep.append((dW, db, colsum_dx, d_in_slope, ws))
        else:
            d.used -= 1  # reduced on the spot: the workspace slot is free again
        return dx
    ws = workspace(nb, dy.device)
    _check(lib().gcl_linear_bwd_all(
        _p(dy), _ld(dy), _p(W), _p(x), _ld(x), _p(in_slope), _p(d_in_slope), _p(dx), Fin, _p(dW), _p(db),
        _p(colsum_dx), rows, Fin, Fout, acc, ws.data_ptr(), ws.numel(), _stream()))
    return dx


def aggregate(graph: Graph, h3, bias, transpose=False, out=None):
    """h3: [B, n, F] with unit channel stride; returns [B, n, F]."""
    B, n, F = h3.shape
    assert n == graph.n, f"graph has {graph.n} nodes, features have {n} rows"
    assert h3.stride(2) == 1
    if out is None:
        out = torch.empty(B, n, F, dtype=torch.float32, device=h3.device)
    tok = _probe_begin("aggregate", graph=graph, transpose=bool(transpose), B=B, F=F)
    _check(lib().gcl_aggregate(graph.handle, 1 if transpose else 0, _p(h3), h3.stride(1), h3.stride(0), _p(bias),
                               _p(out), out.stride(1), out.stride(0), B, F, _stream()))
    _probe_end(tok)
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    rows, F = x.shape
    y = torch.empty(rows, F, dtype=torch.float32, device=x.device)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    _check(lib().gcl_layernorm_fwd(_p(x), _ld(x), _p(gamma), _p(beta), float(eps), _p(y), F, _p(stats), rows, F, _stream()))
    return y, stats


def layernorm_fwd_map(x, gamma, beta, eps, out3, pos):
    """LayerNorm of x [B * n, F] with row (b, i) written to out3[b, pos[i], :F] (pos[i] >= 0) and dropped otherwise;
    returns the (mean, rstd) statistics of every row."""
    rows, F = x.shape
    n_per = pos.numel()
    assert out3.stride(2) == 1 and rows % n_per == 0 and out3.shape[0] == rows // n_per
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    _check(lib().gcl_layernorm_fwd_map(_p(x), _ld(x), _p(gamma), _p(beta), float(eps), _p(out3), out3.stride(1), out3.stride(0),
                                       _pi(pos), n_per, _p(stats), rows, F, _stream()))
    return stats


def layernorm_bwd(dy, x, gamma, stats, dgamma, dbeta, accumulate: bool, colsum_dx=None, acc_colsum: bool = False,
                  dy_map=None):
    """dx of the node LayerNorm (+ dgamma, dbeta); `colsum_dx` also receives the column sums of dx (the bias
    gradient of the layer below) from the same pass.  dy_map = (src3 [B, m, F'], pos int32 [n]): dy is not dense -
    row (b, i) reads src3[b, pos[i], :F] (zero where pos[i] < 0); x then holds B * n rows."""
    rows, F = x.shape
    dx = torch.empty(rows, F, dtype=torch.float32, device=x.device)
    nb = lib().gcl_layernorm_bwd_ws_bytes(rows, F)
    ws = workspace(nb, x.device)
    acc = (ACC_DW if accumulate else 0) | (ACC_COLSUM if acc_colsum else 0)
    if dy_map is not None:
        src3, pos = dy_map
        assert src3.stride(2) == 1 and src3.shape[2] >= F and rows % pos.numel() == 0 and src3.shape[0] == rows // pos.numel()
        _check(lib().gcl_layernorm_bwd_map(_p(src3), src3.stride(1), src3.stride(0), _pi(pos), pos.numel(), _p(x), _ld(x),
                                           _p(gamma), _p(stats), _p(dx), F, _p(dgamma), _p(dbeta), _p(colsum_dx), acc, rows, F,
                                           ws.data_ptr(), ws.numel(), _stream()))
        return dx
    _check(lib().gcl_layernorm_bwd_cs(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(stats), _p(dx), F, _p(dgamma),
                                      _p(dbeta), _p(colsum_dx), acc, rows, F, ws.data_ptr(), ws.numel(), _stream()))
    return dx


def graphnorm_fwd(x3, gamma, beta, eps=1e-5):
    B, n, F = x3.shape
    y = torch.empty(B, n, F, dtype=torch.float32, device=x3.device)
    stats = torch.empty(B, 2, dtype=torch.float32, device=x3.device)
    nb = lib().gcl_graphnorm_ws_bytes(B, n, F)
    ws = workspace(nb, x3.device)
    _check(lib().gcl_graphnorm_fwd(_p(x3), x3.stride(1), x3.stride(0), _p(gamma), _p(beta), float(eps), _p(y), F, n * F,
                                   _p(stats), B, n, F, ws.data_ptr(), ws.numel(), _stream()))
    return y, stats


def graphnorm_bwd(dy3, x3, gamma, stats, dgamma, dbeta, accumulate: bool, eps=1e-5):
    B, n, F = x3.shape
    dx = torch.empty(B, n, F, dtype=torch.float32, device=x3.device)
    nb = lib().gcl_graphnorm_ws_bytes(B, n, F)
    ws = workspace(nb, x3.device)
    _check(lib().gcl_graphnorm_bwd(_p(dy3), dy3.stride(1), dy3.stride(0), _p(x3), x3.stride(1), x3.stride(0), _p(gamma),
                                   _p(stats), float(eps), _p(dx), F, n * F, _p(dgamma), _p(dbeta),
                                   1 if accumulate else 0, B, n, F, ws.data_ptr(), ws.numel(), _stream()))
    return dx


def colsum(x, out, accumulate: bool):
    rows, F = x.shape
    nb = lib().gcl_colsum_ws_bytes(rows, F)
    ws = workspace(nb, x.device)
    _check(lib().gcl_colsum(_p(x), _ld(x), rows, F, _p(out), 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()))
    return out


def gat_tab_ok(graph: Graph, H: int, Cc: int) -> bool:
    """True when gat_fwd / gat_bwd can read the rows of h through a row table (one head, source-tile graph)."""
    return bool(lib().gcl_gat_tab_ok(graph.handle, H * Cc, int(H), int(Cc)))


def gat_fwd(graph: Graph, h3, att_src, att_dst, bias, H, Cc, need_alpha=True, tab=None):
    """tab (int32 [graph.n]): h3 is [B, rows, H*C] and mesh row i of sample b is h3[b, tab[i]] or the flat row ~tab[i]."""
    B, n, HC = h3.shape
    if tab is not None:
        n = graph.n
        assert h3.is_contiguous()
    dev = h3.device
    a_s = torch.empty(B, n, H, dtype=torch.float32, device=dev)
    a_d = torch.empty(B, n, H, dtype=torch.float32, device=dev)
    alpha = torch.empty(B, graph.e, H, dtype=torch.float32, device=dev) if need_alpha else None
    y = torch.empty(B, n, Cc, dtype=torch.float32, device=dev)
    tok = _probe_begin("gat_fwd", graph=graph, B=B, H=H, C=Cc, alpha=need_alpha)
    if tab is not None:
        _check(lib().gcl_gat_fwd_tab(graph.handle, _p(h3), h3.stride(1), h3.stride(0), _pi(tab), _p(att_src), _p(att_dst),
                                     _p(bias), _p(a_s), _p(a_d), _p(alpha), _p(y), Cc, n * Cc, B, H, Cc, _stream()))
    else:
        _check(lib().gcl_gat_fwd(graph.handle, _p(h3), h3.stride(1), h3.stride(0), _p(att_src), _p(att_dst), _p(bias),
                                 _p(a_s), _p(a_d), _p(alpha), _p(y), Cc, n * Cc, B, H, Cc, _stream()))
    _probe_end(tok)
    return y, a_s, a_d, alpha


def gat_bwd(graph: Graph, dy3, h3, att_src, att_dst, a_s, a_d, alpha, d_att_src, d_att_dst, d_bias, accumulate, H, Cc, tab=None):
    B, n, HC = h3.shape
    if tab is not None:
        n = graph.n  # dh is dense [B, n, HC]: the gradient of the table-read rows
    dy3 = dy3.contiguous()
    dh = torch.empty(B, n, HC, dtype=torch.float32, device=h3.device)
    nb = lib().gcl_gat_bwd_ws_bytes(graph.e, n, B, H, Cc)
    ws = workspace(nb, h3.device)
    tok = _probe_begin("gat_bwd", graph=graph, B=B, H=H, C=Cc)
    if tab is not None:
        _check(lib().gcl_gat_bwd_tab(graph.handle, _p(dy3), Cc, n * Cc, _p(h3), h3.stride(1), h3.stride(0), _pi(tab), _p(att_src),
                                     _p(att_dst), _p(a_s), _p(a_d), _p(alpha), _p(dh), HC, n * HC, _p(d_att_src),
                                     _p(d_att_dst), _p(d_bias), 1 if accumulate else 0, B, H, Cc, ws.data_ptr(), ws.numel(),
                                     _stream()))
    else:
        _check(lib().gcl_gat_bwd(graph.handle, _p(dy3), Cc, n * Cc, _p(h3), h3.stride(1), h3.stride(0), _p(att_src),
                                 _p(att_dst), _p(a_s), _p(a_d), _p(alpha), _p(dh), HC, n * HC, _p(d_att_src), _p(d_att_dst),
                                 _p(d_bias), 1 if accumulate else 0, B, H, Cc, ws.data_ptr(), ws.numel(), _stream()))
    _probe_end(tok)
    return dh


def gat_alpha_edge_order(graph: Graph, alpha_slots_1sample, H):
    out = torch.empty(graph.e, H, dtype=torch.float32, device=alpha_slots_1sample.device)
    _check(lib().gcl_gat_alpha_to_edge_order(graph.handle, _p(alpha_slots_1sample.contiguous()), _p(out), H, _stream()))
    return out


def gat_prune(graph: Graph, alpha_edges, threshold: float) -> torch.Tensor:
    """Surviving PyG-order edge list (CPU int64 `[2, kept]`)."""
    buf = torch.empty(2 * graph.e, dtype=torch.int64)
    kept = _i64(0)
    nb = lib().gcl_gat_prune_ws_bytes(graph.e)
    ws = workspace(nb, alpha_edges.device)
    _check(lib().gcl_gat_prune(graph.handle, _p(alpha_edges.contiguous()), float(threshold), buf.data_ptr(),
                               C.byref(kept), ws.data_ptr(), ws.numel(), _stream()))
    k = kept.value
    return buf[: 2 * k].view(2, k).clone()


def assemble_input(x3, grid_static, mesh_static, tail3=None):
    """[B, G + M, Cdyn + Cs] = grid rows [x | grid_static], mesh rows [0 | mesh_static]; `tail3` [B, r, C] (optional)
    overwrites the last r mesh rows of every sample (per-sample rows: the folded batch-invariant mesh rows)."""
    B, G, Cdyn = x3.shape
    M, Cs = mesh_static.shape
    x3 = x3.contiguous()
    out = torch.empty(B, G + M, Cdyn + Cs, dtype=torch.float32, device=x3.device)
    if tail3 is not None and tail3.is_contiguous() and tail3.shape[2] == Cdyn + Cs:
        _check(lib().gcl_assemble_input_tail(_p(x3), _p(grid_static), _p(mesh_static), _p(tail3), tail3.shape[1], _p(out),
                                             Cdyn + Cs, B, G, M, Cdyn, Cs, _stream()))
        return out
    _check(lib().gcl_assemble_input(_p(x3), _p(grid_static), _p(mesh_static), _p(out), Cdyn + Cs, B, G, M, Cdyn, Cs, _stream()))
    if tail3 is not None:
        copy_rows(tail3, out[:, G + M - tail3.shape[1]:, :])
    return out


def wmse_fwd_bwd(delta3, x_last3, y3, node_w, chan_w, inv_wsum, grad_scale, want_grad=True, want_state=False,
                 loss_prev=None):
    """delta3 [B,G,C] contiguous; x_last3 / y3 may be strided views (unit channel stride)."""
    B, G, Cc = delta3.shape
    dev = delta3.device
    if delta3.stride(2) != 1:
        delta3 = delta3.contiguous()  # (a row-strided view - e.g. the grid rows of a padded output - is read in place)
    dd = torch.empty(B, G, Cc, dtype=torch.float32, device=dev) if want_grad else None
    st = torch.empty(B, G, Cc, dtype=torch.float32, device=dev) if want_state else None
    loss = torch.empty((), dtype=torch.float32, device=dev)
    nb = lib().gcl_wmse_ws_bytes(B, G, Cc)
    ws = workspace(nb, dev)
    xl = x_last3
    _check(lib().gcl_wmse_fwd_bwd(
        _p(delta3), delta3.stride(1), delta3.stride(0), _p(xl), xl.stride(1) if xl is not None else 0, xl.stride(0) if xl is not None else 0,
        _p(y3), y3.stride(1), y3.stride(0), _p(node_w), _p(chan_w), float(inv_wsum), float(grad_scale), _p(dd), _p(st),
        _p(loss_prev), _p(loss), B, G, Cc, ws.data_ptr(), ws.numel(), _stream()))
    return loss, dd, st


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _check(lib().gcl_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, int(step),
                               float(grad_scale), _stream()))


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_dev, bc_dev, grad_scale=1.0):
    assert step_dev.dtype == torch.int32 and step_dev.is_cuda
    _check(lib().gcl_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                   step_dev.data_ptr(), _p(bc_dev), float(grad_scale), _stream()))


def copy_rows(src3, dst3):
    B, rows, F = src3.shape
    _check(lib().gcl_copy_rows(_p(src3), src3.stride(1), src3.stride(0), _p(dst3), dst3.stride(1), dst3.stride(0), B, rows, F, _stream()))
    return dst3


def _pi(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    return t.data_ptr()


def gather2_rows(a3, map_a, b3, map_b, nd: int, B: int, sum_batch: bool = False, out=None, deal: int = 0):
    """dst[b,i] = a3[b, map_a[i]] | b3[b or 0, map_b[i]] | 0   (see gcl_gather2_rows).
    a3 / b3: [Ba, na, F] with unit channel stride; Ba == 1 broadcasts over B.  `out`: a (possibly row-strided)
    [B | 1, nd, F] destination, e.g. a row range of a larger buffer.  sum_batch with deal = R > 1: the nd batch-summed
    rows are stored R to a destination sample, out [nd / R, R, F] (possibly strided)."""
    F = a3.shape[-1]
    if sum_batch and deal > 1:
        assert out is not None and nd % deal == 0 and out.shape == (nd // deal, deal, F) and out.stride(2) == 1
        flag = deal
    else:
        if out is None:
            out = torch.empty(1 if sum_batch else B, nd, F, dtype=torch.float32, device=a3.device)
        assert out.shape == (1 if sum_batch else B, nd, F) and out.stride(2) == 1
        flag = 1 if sum_batch else 0
    bsa = a3.stride(0) if (a3.shape[0] > 1 or sum_batch) else 0
    bsb = 0 if b3 is None else (b3.stride(0) if b3.shape[0] > 1 else 0)
    _check(lib().gcl_gather2_rows(_p(a3), a3.stride(1), bsa, _pi(map_a), _p(b3), 0 if b3 is None else b3.stride(1),
                                  bsb, _pi(map_b), _p(out), out.stride(1), out.stride(0), B, nd, F, flag, _stream()))
    return out


def ar_advance(state4, delta3, y_step3, chan_kind, out3, out_off: int, residual: bool):
    """state4 [B,G,obs,C] contiguous -> new state (same shape); appends the step to out3 [B,G,steps*C]."""
    B, G, obs, Cc = state4.shape
    assert state4.is_contiguous() and delta3.stride(2) == 1
    new_state = torch.empty_like(state4)
    _check(lib().gcl_ar_advance(
        _p(state4), _p(delta3), delta3.stride(1), delta3.stride(0), _p(y_step3), y_step3.stride(1) if y_step3 is not None else 0,
        y_step3.stride(0) if y_step3 is not None else 0, _pi(chan_kind), _p(new_state), _p(out3),
        out3.stride(1) if out3 is not None else 0, out3.stride(0) if out3 is not None else 0, int(out_off), B, G, obs, Cc,
        1 if residual else 0, _stream()))
    return new_state


def ar_step_bwd(dd3, g_loss, g_new4, chan_kind, has_y: bool, residual: bool, obs: int, want_state: bool):
    """Backward of one autoregressive training step (see gcl_ar_step_bwd): (d_delta [B,G,C], d_state [B,G,obs,C] | None)."""
    B, G, Cc = dd3.shape
    d_delta = torch.empty_like(dd3)
    d_state = torch.empty(B, G, obs, Cc, dtype=torch.float32, device=dd3.device) if want_state else None
    if g_new4 is not None and not g_new4.is_contiguous():
        g_new4 = g_new4.contiguous()
    _check(lib().gcl_ar_step_bwd(_p(dd3), _p(g_loss), _p(g_new4), _pi(chan_kind), 1 if has_y else 0, 1 if residual else 0,
                                 _p(d_delta), _p(d_state), B, G, obs, Cc, _stream()))
    return d_delta, d_state


def pad_rows(src3, rows_dst: int, F_dst: int):
    """[B, r, F] (unit channel stride) -> zero-padded contiguous [B, rows_dst, F_dst] in one pass."""
    B, r, F = src3.shape
    assert src3.stride(2) == 1
    dst = torch.empty(B, rows_dst, F_dst, dtype=torch.float32, device=src3.device)
    _check(lib().gcl_pad_rows(_p(src3), src3.stride(1), src3.stride(0), r, F, _p(dst), F_dst, rows_dst * F_dst, rows_dst, F_dst,
                              B, _stream()))
    return dst


def zero_(t: torch.Tensor):
    """t.zero_() as a stream memset through the C ABI (no torch fill kernel on the step)."""
    assert t.is_contiguous() and t.is_cuda
    _check(lib().gcl_zero(t.data_ptr(), t.numel() * t.element_size(), _stream()))
    return t


def segment_reduce(src3, perm, rowptr, mean: bool, out3=None):
    """src3 [B, E, D] (unit channel stride) -> out3 [B, n, D]: per-segment sum / mean of rows
    perm[rowptr[i]:rowptr[i+1]] (perm None: the rows themselves)."""
    B, _, D = src3.shape
    n = rowptr.numel() - 1
    if out3 is None:
        out3 = torch.empty(B, n, D, dtype=torch.float32, device=src3.device)
    assert src3.stride(2) == 1 and out3.stride(2) == 1 and tuple(out3.shape) == (B, n, D)
    _check(lib().gcl_segment_reduce(_p(src3), src3.stride(1), src3.stride(0), _pi(perm), _pi(rowptr), 1 if mean else 0,
                                    _p(out3), out3.stride(1), out3.stride(0), B, n, D, _stream()))
    return out3


def edge_combine(base3, extra3, A3, ia, sa, C3, ic, out3=None):
    """out[b,e] = base[b,e] + extra[b,e] + A[b, ia[e]] * sa[ia[e]] + C[b, ic[e]]  (operands optional)."""
    ref = base3 if base3 is not None else extra3
    if ref is not None:
        B, E, D = ref.shape
    else:
        B, E, D = (A3 if A3 is not None else C3).shape[0], ia.numel() if ia is not None else ic.numel(), \
            (A3 if A3 is not None else C3).shape[2]
    for t in (base3, extra3):
        assert t is None or t.is_contiguous()
    if out3 is None:
        out3 = torch.empty(B, E, D, dtype=torch.float32, device=(A3 if ref is None else ref).device)
    _check(lib().gcl_edge_combine(
        _p(base3), _p(extra3), _p(A3), A3.stride(1) if A3 is not None else 0, A3.stride(0) if A3 is not None else 0,
        _pi(ia), _p(sa), _p(C3), C3.stride(1) if C3 is not None else 0, C3.stride(0) if C3 is not None else 0, _pi(ic),
        _p(out3), B, E, D, _stream()))
    return out3


def act_fwd(x, act, slope=None):
    y = torch.empty_like(x)
    assert x.is_contiguous()
    _check(lib().gcl_act_fwd(_p(x), _p(y), x.numel(), int(act), _p(slope), _stream()))
    return y


def act_bwd(x, dy, act, slope=None, d_slope=None):
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    ws = workspace(lib().gcl_act_bwd_ws_bytes(), x.device)
    _check(lib().gcl_act_bwd(_p(x), _p(dy), _p(dx), x.numel(), int(act), _p(slope), _p(d_slope), ws.data_ptr(), ws.numel(),
                             _stream()))
    return dx


def window_pack(series, t0, mean, std, C: int, obs: int, pred: int, out=None):
    """series: fp16 [T, n_lon, n_lat, Ct] (or flat [T, N, Ct]) on the GPU; t0: int64 [B] window starts on
    the GPU.  Returns X [B, G, obs*C] and Y [B, G, pred*C] (None when pred == 0).  out = (X, Y): contiguous fp32
    buffers of those shapes to fill in place (e.g. TrainStep.input_buffers())."""
    assert series.is_cuda and series.dtype == torch.float16 and series.is_contiguous()
    assert t0.is_cuda and t0.dtype == torch.int64 and t0.is_contiguous()
    if series.dim() == 3:
        T, n_lon, Ct = series.shape
        n_lat = 1
    else:
        T, n_lon, n_lat, Ct = series.shape
    B, G = t0.numel(), n_lon * n_lat
    if out is not None:
        X, Y = out
        assert X.is_contiguous() and X.shape == (B, G, obs * C) and X.dtype == torch.float32 and X.device == series.device
        assert pred == 0 or (Y.is_contiguous() and Y.shape == (B, G, pred * C) and Y.dtype == torch.float32)
    else:
        X = torch.empty(B, G, obs * C, dtype=torch.float32, device=series.device)
        Y = torch.empty(B, G, pred * C, dtype=torch.float32, device=series.device) if pred > 0 else None
    _check(lib().gcl_window_pack(series.data_ptr(), T, n_lon, n_lat, Ct, t0.data_ptr(), _p(mean), _p(std), C, obs, pred,
                                 _p(X), _p(Y), B, _stream()))
    return X, Y


def gcn_layer_fusable(graph: Graph, x3, Fin: int, Fout: int) -> bool:
    """Does this GCNConv layer fit the one-kernel path (csrc/gcn_layer.hip)?  GCL_FUSED_GCN=0 forces the
    two-kernel path (linear + aggregate) for A/B measurements."""
    import os
    if os.environ.get("GCL_FUSED_GCN", "1") in ("0",):
        return False
    return (Fin % 4 == 0 and 4 <= Fin <= 64 and 1 <= Fout <= 64 and graph.max_in_degree <= 64 and graph.kind in (GRAPH_GCN, GRAPH_MEAN)
            and x3.stride(2) == 1 and x3.stride(1) % 4 == 0 and x3.stride(0) % 4 == 0 and x3.data_ptr() % 16 == 0)


def gcn_layer_fwd(graph: Graph, x3, act, slope, W, bias, out=None, rows_out=None):
    """y = (A_hat act(x)) W^T + bias for x3 [B, n, Fin] -> [B, n, Fout] (one kernel).  The result is a view of
    a [B, n, roundup(Fout, 4)] buffer whose padding columns are zero.  rows_out: only the first rows_out rows of every
    sample are computed (the others stay unwritten)."""
    B, n, Fin = x3.shape
    assert n == graph.n
    Fout = W.shape[0]
    Fst = (Fout + 3) // 4 * 4
    if out is None:
        out = torch.empty(B, n, Fst, dtype=torch.float32, device=x3.device)
    assert out.shape[2] >= Fst or out.stride(1) >= Fst
    tok = _probe_begin("gcn_layer_fwd", graph=graph, B=B, Fin=Fin, Fout=Fout)
    _check(lib().gcl_gcn_layer_fwd_rows(graph.handle, _p(x3), x3.stride(1), x3.stride(0), int(act), _p(slope),
                                        _p(W.contiguous()), _p(bias), _p(out), out.stride(1), out.stride(0), B, Fin, Fout,
                                        Fst, int(rows_out) if rows_out else n, _stream()))
    _probe_end(tok)
    return out[..., :Fout]


def gcn_layer_tab_ok(graph: Graph, x3, Fout: int) -> bool:
    """True when gcn_layer_fwd_tab would run on x3 [B, rows, Fin] (source-tile graph, 48 / 64-wide rows, 32-bit offsets)."""
    B, nx, Fin = x3.shape
    return bool(lib().gcl_gcn_layer_fwd_tab_ok(graph.handle, x3.stride(1), x3.stride(0), B * nx, B, Fin, int(Fout)))


def gcn_layer_fwd_tab(graph: Graph, x3, tab, act, slope, W, bias):
    """The one-kernel GCNConv layer whose input row i of sample b is x3[b, tab[i]] (tab[i] >= 0) or the batch-invariant
    row ~tab[i] of x3 viewed as [B * rows, Fin] (tab[i] < 0): the mesh latents of the compact pipeline are never
    materialised (see gcl_gcn_layer_fwd_tab).  x3 [B, rows, Fin] contiguous, tab int32 [graph.n]."""
    B, nx, Fin = x3.shape
    assert x3.is_contiguous() and tab.dtype == torch.int32 and tab.numel() == graph.n
    Fout = W.shape[0]
    Fst = (Fout + 3) // 4 * 4
    out = torch.empty(B, graph.n, Fst, dtype=torch.float32, device=x3.device)
    tok = _probe_begin("gcn_layer_fwd", graph=graph, B=B, Fin=Fin, Fout=Fout)
    _check(lib().gcl_gcn_layer_fwd_tab(graph.handle, _p(x3), x3.stride(1), x3.stride(0), B * nx, _pi(tab), int(act), _p(slope),
                                       _p(W.contiguous()), _p(bias), _p(out), out.stride(1), out.stride(0), B, Fin, Fout, Fst,
                                       _stream()))
    _probe_end(tok)
    return out[..., :Fout]
