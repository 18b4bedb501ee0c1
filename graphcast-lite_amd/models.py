"""Module API of the reference's `src/models.py`, executed by hand-written gfx950 kernels.

Same class names, constructor arguments, attribute names and state-dict keys as the reference
 so that `src/train.py`-style callers and reference checkpoints work unchanged:

  MLP               src/models.py:54-109     `MLP.{i}.weight|bias`
  SparseGATConv     src/models.py:112-151
  InteractionNet*   src/models.py:166-285    `layers.edge_encoder.0.*`, `layers.steps.{k}.edge_mlp|node_mlp.{0,2}.*`,
                                             `layers.steps.{k}.edge_norm|node_norm.*`
  GraphLayer        src/models.py:289-440    `activation.weight`, `layers.{i}.lin.weight|bias|att_*`
  Model             src/models.py:443-473    `mlp.*`, `graph_layer.*`
  WeatherPrediction src/models.py:476-927    `encoder|processor|decoder.*`, `_processing_edge_features`

What is new relative to the reference (documented deviations):
  * a batch dimension: `[B, G, C]` is B independent samples (the reference only runs B = 1 because
    of its `X.squeeze()`); `[1, G, C]` / `[G, C]` still return `[G, C_out]`;
  * graph normalisation (`gcn_norm`, self-loop handling, degree counts) is done once per graph
    when the CSR handle is built, not on every forward;
  * the three `summary()` forward passes of the constructor are skipped.
Compute runs only on a GPU through `libgcl_hip.so`; there is no CPU path in this package.
"""
import math
import os
import sys
from collections import OrderedDict
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .config import (
    DataConfig,
    GraphBlock,
    GraphBuildingConfig,
    GraphLayerType,
    MLPBlock,
    ModelConfig,
    PipelineConfig,
)
from .create_graphs import (create_decoding_graph, create_encoding_graph, create_processing_graph,
                            create_product_graph)
from .functional import (AssembleFn, EdgeLayout, GATLayerFn, Gather2Fn, GCNStackFn, GradLanding, GraphNormFn, InteractionNetFn,
                         LatSource, LayerNormFn, MeanAggFn, MeshLatFn, MLPFn)
from .mesh import get_hierarchy_of_triangular_meshes_for_sphere, get_mesh_lat_long, prune_mesh_to_region, tile_order


# ------------------------------------------------------------------------------------------------
# CSR handle cache: modules receive reference-layout `edge_index` tensors (the reference API) and
# look the device CSR up by tensor identity + version, so normalisation is paid once per graph.
# ------------------------------------------------------------------------------------------------
class _GraphCache:
    def __init__(self, capacity: int = 32):
        self._d = OrderedDict()
        self._cap = capacity
        # While a hipGraph is being captured the handles it uses are appended here, so that the owner
        # of the captured graph keeps them alive even after the LRU evicts them (a replay would
        # otherwise read freed CSR arrays).
        self.pin = None

    def get(self, edge_index: torch.Tensor, n: int, kind: int) -> hip.Graph:
        key = (id(edge_index), edge_index._version, tuple(edge_index.shape), int(n), kind)
        hit = self._d.get(key)
        if hit is not None:
            self._d.move_to_end(key)
            if self.pin is not None:
                self.pin.append(hit)
            return hit[1]
        g = hip.Graph(edge_index, n, kind)
        self._d[key] = (edge_index, g)  # keep the tensor alive so its id cannot be reused
        if self.pin is not None:
            self.pin.append(self._d[key])
        if len(self._d) > self._cap:
            self._d.popitem(last=False)
        return g


_graphs = _GraphCache()


def _glorot_(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


def _num_nodes(X: torch.Tensor) -> int:
    return X.shape[-2]


class LayerNorm(nn.Module):
    """PyG `LayerNorm(in_channels, eps=1e-5, affine=True, mode)` parameter holder + node-mode kernel."""

    def __init__(self, in_channels: int, eps: float = 1e-5, affine: bool = True, mode: str = "graph"):
        super().__init__()
        self.in_channels, self.eps, self.mode = in_channels, eps, mode or "graph"
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))

    def forward(self, x):
        if self.mode == "node":
            return LayerNormFn.apply(x, self, self.eps, self.weight, self.bias)
        if self.mode == "graph":
            return GraphNormFn.apply(x, self, self.eps, self.weight, self.bias)
        raise ValueError(f"Unknown normalization mode: {self.mode}")


class GCNConv(nn.Module):
    """Parameter holder with PyG GCNConv's keys (`lin.weight` [out,in], `bias` [out])."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot_(self.lin.weight)

    def forward(self, x, edge_index):
        g = _graphs.get(edge_index, _num_nodes(x), hip.GRAPH_GCN)
        return GCNStackFn.apply(x, self, g, 1, False, 1e-5, 0, self.lin.weight, self.bias, None)


class GATConv(nn.Module):
    """Parameter holder with PyG 2.5 GATConv's keys (`lin.weight` [H*C,in], `att_src`, `att_dst`
    [1,H,C], `bias` [C]); `concat=False` only, as the reference uses it (src/models.py:336,364)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = False,
                 dropout: float = 0.0, bias: bool = True, **kwargs):
        super().__init__()
        if concat or dropout != 0.0 or not bias:
            raise NotImplementedError("only GATConv(concat=False, dropout=0, bias=True) is on the HIP path")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot_(self.lin.weight)
        _glorot_(self.att_src)
        _glorot_(self.att_dst)

    def _run(self, x, edge_index, slope, want_alpha, act=None, lat_src=None):
        g = _graphs.get(edge_index, lat_src.M if lat_src is not None else _num_nodes(x), hip.GRAPH_GAT)
        self._in_act = act
        self._lat_src = lat_src  # functional.LatSource (picked up by GATLayerFn): x is the encoder output
        y, alpha = GATLayerFn.apply(x, self, g, self.heads, want_alpha, slope, self.lin.weight, self.att_src,
                                    self.att_dst, self.bias)
        return y, alpha, g

    def forward(self, x, edge_index, return_attention_weights: bool = False):
        y, alpha, g = self._run(x, edge_index, None, return_attention_weights)
        if return_attention_weights:
            return y, (g.edges_with_loops(x.device), alpha)
        return y

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # older PyG stored two aliases lin_src / lin_dst instead of one lin
        src, dst, lin = prefix + "lin_src.weight", prefix + "lin_dst.weight", prefix + "lin.weight"
        if lin not in state_dict and src in state_dict:
            state_dict[lin] = state_dict.pop(src)
            state_dict.pop(dst, None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class SparseGATConv(GATConv):
    """`src/models.py:112-151`: GAT that also returns the (optionally pruned) edge list."""

    def forward(self, x, edge_index, attention_threshold=0.0, **kwargs):
        batch_num = kwargs.get("batch_num", 1)
        slope = kwargs.get("_in_slope", None)
        out, alpha_edges, g = self._run(x, edge_index, slope, True)
        alpha = alpha_edges.squeeze()  # [E'] for heads == 1, as in the reference
        if batch_num == 0:
            print("edge_index", torch.Size([2, g.e]), file=sys.stderr)  # the reference prints it too (src/models.py:139): stderr here
            new_ei = hip.gat_prune(g, alpha, float(attention_threshold)).to(x.device)
            new_ei = _broadcast_edges_from_rank0(new_ei)  # C2: every rank keeps rank 0's pruned graph
            mask = alpha >= attention_threshold
            return out, (new_ei, alpha[mask])
        return out, (g.edges_with_loops(x.device), alpha)


def _broadcast_edges_from_rank0(edge_index: torch.Tensor) -> torch.Tensor:
    """Collective C2: under data parallelism each rank sees different samples, so the
    attention-based prune decision (taken from sample 0 of batch 0, src/models.py:138-149) would
    differ per rank; rank 0's surviving edge list is broadcast (count first, then the int64 list)
    so that all ranks rebuild the same CSR.  No-op in single-process runs."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return edge_index
    dev = edge_index.device
    count = torch.tensor([edge_index.shape[1]], dtype=torch.int64, device=dev)
    dist.broadcast(count, src=0)
    k = int(count.item())
    buf = edge_index.contiguous() if dist.get_rank() == 0 else torch.empty(2, k, dtype=torch.int64, device=dev)
    dist.broadcast(buf, src=0)
    return buf


class SimpleConv(nn.Module):
    """PyG `SimpleConv(aggr="mean")` (no parameters)."""

    def __init__(self, aggr: str = "mean"):
        super().__init__()
        if aggr != "mean":
            raise NotImplementedError("only SimpleConv(aggr='mean') is on the HIP path")
        self.aggr = aggr

    def forward(self, x, edge_index):
        return MeanAggFn.apply(x, _graphs.get(edge_index, _num_nodes(x), hip.GRAPH_MEAN))


def _get_activation(name: str = "prelu"):
    """`src/models.py:154-163`.  The modules only carry the parameters / state-dict keys; the kernels
    apply the activation while loading the next layer's input (`_act_spec`)."""
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "prelu":
        return nn.PReLU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(f"Unknown activation: {name}")


def _act_spec(owner: nn.Module, act: nn.Module):
    """Sets `owner.act_kind` / `owner.const_slope` for the HIP Functions and returns the learnable
    slope (or None).  ReLU runs as a PReLU with a constant zero slope."""
    owner.act_kind = hip.ACT_SILU if isinstance(act, nn.SiLU) else hip.ACT_PRELU
    if isinstance(act, nn.ReLU):
        owner.register_buffer("const_slope", torch.zeros(1), persistent=False)
    else:
        owner.const_slope = None
    return act.weight if isinstance(act, nn.PReLU) else None


class InteractionNetLayer(nn.Module):
    """Parameter holder of one message-passing step with the reference's keys (`src/models.py:166-204`):
    `edge_mlp.{0,2}`, `node_mlp.{0,2}`, `edge_norm` (graph mode), `node_norm` (node mode).  The
    compute is `functional.InteractionNetFn`, run by the processor over all steps."""

    def __init__(self, node_dim: int, edge_dim: int, hidden_dim: int, activation: str = "swish",
                 use_layer_norm: bool = True):
        super().__init__()
        if not (node_dim == edge_dim == hidden_dim):
            raise NotImplementedError("InteractionNet on the HIP path needs node = edge = hidden width "
                                      "(what GraphLayer builds, src/models.py:390-398)")
        act = _get_activation(activation)  # one instance inside both MLPs (src/models.py:184-198)
        self.edge_mlp = nn.Sequential(nn.Linear(node_dim * 2 + edge_dim, hidden_dim), act, nn.Linear(hidden_dim, edge_dim))
        self.node_mlp = nn.Sequential(nn.Linear(node_dim + edge_dim, hidden_dim), act, nn.Linear(hidden_dim, node_dim))
        self.use_layer_norm = use_layer_norm
        if use_layer_norm:
            self.edge_norm = LayerNorm(edge_dim, mode="graph")
            self.node_norm = LayerNorm(node_dim, mode="node")

    def step_params(self):
        act = self.edge_mlp[1]
        ln = self.use_layer_norm
        return [self.edge_mlp[0].weight, self.edge_mlp[0].bias, self.edge_mlp[2].weight, self.edge_mlp[2].bias,
                self.node_mlp[0].weight, self.node_mlp[0].bias, self.node_mlp[2].weight, self.node_mlp[2].bias,
                act.weight if isinstance(act, nn.PReLU) else None,
                self.edge_norm.weight if ln else None, self.edge_norm.bias if ln else None,
                self.node_norm.weight if ln else None, self.node_norm.bias if ln else None]


class InteractionNetProcessor(nn.Module):
    """`src/models.py:239-285`: edge encoder (raw 4-D edge features -> latent) + N unshared steps."""

    def __init__(self, node_dim: int, raw_edge_dim: int, edge_latent_dim: int, hidden_dim: int, num_steps: int,
                 activation: str = "swish", use_layer_norm: bool = True):
        super().__init__()
        if node_dim % 4 != 0:
            raise NotImplementedError("InteractionNet on the HIP path needs a latent width that is a multiple of 4")
        self.edge_encoder = nn.Sequential(nn.Linear(raw_edge_dim, edge_latent_dim), _get_activation(activation))
        self.steps = nn.ModuleList([
            InteractionNetLayer(node_dim, edge_latent_dim, hidden_dim, activation, use_layer_norm)
            for _ in range(num_steps)])
        self.use_layer_norm = use_layer_norm
        _act_spec(self, self.edge_encoder[1])
        self._layout = None  # (edge_index, raw features) -> receiver-sorted layout, built once

    def _edge_layout(self, edge_index, edge_attr_raw, n, device):
        c = self._layout
        if c is None or c[0] is not edge_index or c[1] is not edge_attr_raw or c[2].n != n:
            lay = EdgeLayout(edge_index, n, device)
            raw = edge_attr_raw.detach().to(device=device, dtype=torch.float32)[lay.order].contiguous()
            pad = (-raw.shape[1]) % 4  # the dense kernels read 16-byte rows
            if pad:
                raw = torch.nn.functional.pad(raw, (0, pad))
            self._layout = c = (edge_index, edge_attr_raw, lay, raw, pad)
        return c[2], c[3], c[4]

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr_raw: torch.Tensor):
        lay, raw, pad = self._edge_layout(edge_index, edge_attr_raw, _num_nodes(x), x.device)
        enc_W = self.edge_encoder[0].weight
        if pad:
            # raw edge features are padded to 16-byte rows (zeros); the tiny [D, raw_dim] encoder weight gets matching
            # zero columns (a differentiable pad of a few hundred floats, outside the per-edge work)
            enc_W = torch.nn.functional.pad(enc_W, (0, pad))
        enc_act = self.edge_encoder[1]
        params = [enc_W, self.edge_encoder[0].bias, enc_act.weight if isinstance(enc_act, nn.PReLU) else None]
        for st in self.steps:
            params += st.step_params()
        eps = self.steps[0].node_norm.eps if self.use_layer_norm else 1e-5
        return InteractionNetFn.apply(x, self, lay, raw, len(self.steps), self.act_kind, self.use_layer_norm, eps, *params)


class MLP(nn.Module):
    """`src/models.py:54-109`."""

    def __init__(self, mlp_config: MLPBlock, input_dim):
        super().__init__()
        hidden = list(mlp_config.mlp_hidden_dims or [])
        self.MLP = nn.ModuleList()
        d = input_dim
        for hdim in hidden:
            self.MLP.extend([nn.Linear(in_features=d, out_features=hdim), nn.PReLU()])
            d = hdim
        self.MLP.append(nn.Linear(in_features=d, out_features=mlp_config.output_dim))
        self._has_ln = bool(mlp_config.use_layer_norm)
        if self._has_ln:
            self.MLP.append(LayerNorm(in_channels=mlp_config.output_dim, mode=mlp_config.layer_norm_mode))

    def forward(self, X: torch.Tensor):
        params = []
        ln = None
        for layer in self.MLP:
            if isinstance(layer, nn.Linear):
                params += [layer.weight, layer.bias]
            elif isinstance(layer, nn.PReLU):
                params.append(layer.weight)
            else:
                ln = layer
        if ln is not None and ln.mode != "node":
            y = MLPFn.apply(X, self, False, 1e-5, *params)
            return ln(y)
        if ln is not None:
            params += [ln.weight, ln.bias]
        return MLPFn.apply(X, self, ln is not None, ln.eps if ln is not None else 1e-5, *params)


class GraphLayer(nn.Module):
    """`src/models.py:289-440`."""

    def __init__(self, graph_config: GraphBlock, input_dim):
        super().__init__()
        self.layer_type: GraphLayerType = graph_config.layer_type
        self.output_dim = None
        lt = graph_config.layer_type
        if lt == GraphLayerType.SimpleConv:
            self.output_dim = input_dim
            self.layers = SimpleConv(aggr="mean")
        elif lt in (GraphLayerType.ConvGCN, GraphLayerType.GATConv, GraphLayerType.SparseGATConv):
            self.activation = _get_activation(graph_config.activation or "prelu")
            _act_spec(self, self.activation)
            self.output_dim = graph_config.output_dim
            self.layers = nn.ModuleList()
            hidden = list(graph_config.hidden_dims or [])
            if lt == GraphLayerType.SparseGATConv:
                self.num_heads = graph_config.gat_props.num_heads
                print(graph_config.layer_type, file=sys.stderr)  # src/models.py:347 prints it; stdout stays clean for callers
                self.layers.append(SparseGATConv(input_dim, graph_config.output_dim, heads=self.num_heads, concat=False))
            else:
                if lt == GraphLayerType.GATConv:
                    self.num_heads = graph_config.gat_props.num_heads
                    mk = lambda i, o: GATConv(i, o, heads=self.num_heads, concat=False)
                else:
                    mk = GCNConv
                dims = [input_dim] + hidden + [graph_config.output_dim]
                for k in range(len(dims) - 1):
                    self.layers.append(mk(dims[k], dims[k + 1]))
                    if k < len(dims) - 2:
                        self.layers.append(self.activation)  # one shared PReLU instance (src/models.py:316-328)
            if graph_config.use_layer_norm:
                self.layers.append(LayerNorm(in_channels=graph_config.output_dim, mode=graph_config.layer_norm_mode))
        elif lt == GraphLayerType.InteractionNet:  # src/models.py:376-398
            self.output_dim = graph_config.output_dim
            assert graph_config.output_dim == input_dim, (
                f"InteractionNet requires output_dim ({graph_config.output_dim}) == input_dim ({input_dim}) "
                f"(residual connections)")
            use_ln = graph_config.use_layer_norm if graph_config.use_layer_norm is not None else True
            self.layers = InteractionNetProcessor(
                node_dim=input_dim, raw_edge_dim=graph_config.edge_feature_dim or 4, edge_latent_dim=input_dim,
                hidden_dim=input_dim, num_steps=graph_config.num_message_passing_steps or 4,
                activation=graph_config.activation or "swish", use_layer_norm=use_ln)
        else:
            print(graph_config.layer_type, file=sys.stderr)
            raise NotImplementedError(f"Layer type {graph_config.layer_type} not supported.")

    @property
    def _slope(self):
        """Learnable slope of the shared activation (PReLU only)."""
        return self.activation.weight if isinstance(self.activation, nn.PReLU) else None

    def _final_ln(self) -> Optional[LayerNorm]:
        last = self.layers[len(self.layers) - 1]
        return last if isinstance(last, LayerNorm) else None

    def forward(self, X: torch.Tensor, edge_index: torch.Tensor, attention_threshold=0.0, **kwargs):
        if self.layer_type == GraphLayerType.SimpleConv:
            return self.layers(x=X, edge_index=edge_index)
        if self.layer_type == GraphLayerType.InteractionNet:
            edge_attr = kwargs.get("edge_attr", None)
            if edge_attr is None:
                raise ValueError("InteractionNet requires edge_attr (edge features)")
            return self.layers(x=X, edge_index=edge_index, edge_attr_raw=edge_attr)
        lat_src = kwargs.get("_lat_src") if self.layer_type in (GraphLayerType.ConvGCN, GraphLayerType.GATConv) else None
        n = lat_src.M if lat_src is not None else _num_nodes(X)  # with a LatSource X is the encoder output, the graph the mesh
        ln = self._final_ln()
        fuse_ln = ln is not None and ln.mode == "node"

        if self.layer_type == GraphLayerType.ConvGCN:
            convs = [m for m in self.layers if isinstance(m, GCNConv)]
            params = []
            for c in convs:
                params += [c.lin.weight, c.bias]
            params.append(self._slope if len(convs) > 1 else None)
            if fuse_ln:
                params += [ln.weight, ln.bias]
            g = _graphs.get(edge_index, n, hip.GRAPH_GCN)
            out_rows = int(kwargs.get("_out_rows") or 0)  # the caller keeps only the first rows (decoder: grid rows)
            self._grad_src = kwargs.get("_grad_src") if fuse_ln else None  # functional.GradLanding (picked up by GCNStackFn)
            self._lat_src = lat_src  # functional.LatSource (picked up by GCNStackFn): X is read through a row table
            if ln is not None and not fuse_ln:
                X = GCNStackFn.apply(X, self, g, len(convs), False, 1e-5, 0, *params)
                X = ln(X)
                return X[..., :out_rows, :] if out_rows else X
            return GCNStackFn.apply(X, self, g, len(convs), fuse_ln, ln.eps if fuse_ln else 1e-5, out_rows, *params)

        if self.layer_type == GraphLayerType.GATConv:
            slope, act = None, hip.ACT_NONE
            for layer in self.layers:
                if isinstance(layer, GATConv):
                    X, _, _ = layer._run(X, edge_index, slope, False, act, lat_src=lat_src)
                    lat_src = None  # only the first conv reads the encoder output
                    # the next conv applies the shared activation while loading its input
                    slope = self._slope if self._slope is not None else self.const_slope
                    act = self.act_kind
                elif isinstance(layer, LayerNorm):
                    if layer is ln and fuse_ln:
                        layer._grad_src = kwargs.get("_grad_src")  # functional.GradLanding (picked up by LayerNormFn)
                    X = layer(X)
            return X

        if self.layer_type == GraphLayerType.SparseGATConv:
            for layer in self.layers:
                if isinstance(layer, SparseGATConv):
                    X, (edge_index, _) = layer.forward(X, edge_index, attention_threshold, **kwargs)
                elif isinstance(layer, LayerNorm):
                    if layer is ln and fuse_ln:
                        layer._grad_src = kwargs.get("_grad_src")  # functional.GradLanding (picked up by LayerNormFn)
                    X = layer(X)
            return X, edge_index
        raise NotImplementedError(f"Layer type {self.layer_type} not supported.")


class Model(nn.Module):
    """`src/models.py:443-473`: optional MLP then GraphLayer."""

    def __init__(self, model_config: ModelConfig, input_dim: int):
        super().__init__()
        self.mlp = None
        self.output_dim = None
        graph_input_dim = input_dim
        if model_config.mlp:
            self.mlp = MLP(mlp_config=model_config.mlp, input_dim=input_dim)
            graph_input_dim = model_config.mlp.output_dim
        self.graph_layer = GraphLayer(graph_config=model_config.gcn, input_dim=graph_input_dim)
        self.output_dim = self.graph_layer.output_dim

    def forward(self, X: torch.Tensor, edge_index: torch.Tensor, attention_threshold=0.0, **kwargs):
        if self.mlp:
            X = self.mlp(X=X)
        return self.graph_layer(X=X, edge_index=edge_index, attention_threshold=attention_threshold, **kwargs)


class WeatherPrediction(nn.Module):
    """`src/models.py:476-927`: encode (grid->mesh) / process (mesh) / decode (mesh->grid)."""

    def __init__(
        self,
        cordinates: Tuple[np.ndarray, np.ndarray],
        graph_config: GraphBuildingConfig,
        pipeline_config: PipelineConfig,
        data_config: DataConfig,
        device,
        region_bounds=None,
        mesh_buffer: float = 15.0,
        flat_grid: bool = False,
    ):
        super().__init__()
        self.device = device
        self.flat_grid = flat_grid
        self.obs_window = data_config.obs_window_used
        self.num_features = data_config.num_features_used
        self.total_feature_size = self.num_features * self.obs_window
        self.use_product_graph = pipeline_config.product_graph is not None
        # dynamic channels per grid node that reach the encoder (src/models.py:575-579,787-789)
        self._dyn_size = self.num_features if self.use_product_graph else self.total_feature_size

        self._init_grid_properties(cordinates[0], cordinates[1], flat_grid)
        self._init_mesh_properties(graph_config, region_bounds, mesh_buffer)
        if self.use_product_graph:  # time x space pre-encoder (src/models.py:517-524,707-774)
            if flat_grid:
                raise NotImplementedError("the product graph is defined on the regular lat x lon grid only")
            pg = pipeline_config.product_graph
            self.product_graph = create_product_graph(self._grid_lat, self._grid_lon, self.obs_window, pg.num_k,
                                                      pg.type).to(device)
            self.product_graph_model = Model(model_config=pg.model, input_dim=self.num_features).to(device)
        ptype = pipeline_config.processor.gcn.layer_type
        self.using_sparse_gat = ptype == GraphLayerType.SparseGATConv
        self.using_interaction_net = ptype == GraphLayerType.InteractionNet
        self._total_nodes = self._num_grid_nodes + self._num_mesh_nodes

        self.encoding_graph, self.init_grid_features, self.init_mesh_features = create_encoding_graph(
            grid_node_lats=self._grid_lat, grid_node_longs=self._grid_lon, mesh_node_lats=self._mesh_nodes_lat,
            mesh_node_longs=self._mesh_nodes_lon, mesh=self._finest_mesh, graph_building_config=graph_config,
            num_grid_nodes=self._num_grid_nodes, flat_grid=self.flat_grid,
        )
        self.init_grid_features = self.init_grid_features.to(device)
        self.init_mesh_features = self.init_mesh_features.to(device)
        self._init_feature_size = self.init_grid_features.shape[1]

        self.processing_graph, proc_edge_features = create_processing_graph(
            meshes=self._meshes, mesh_levels=graph_config.mesh_levels, mesh_node_lats=self._mesh_nodes_lat,
            mesh_node_longs=self._mesh_nodes_lon,
        )
        self.register_buffer("_processing_edge_features", proc_edge_features)
        self.decoding_graph = create_decoding_graph(
            cordinates=cordinates, mesh=self._finest_mesh, graph_building_config=graph_config,
            num_grid_nodes=self._num_grid_nodes, flat_grid=self.flat_grid,
        )

        encoder_input_dim = self._dyn_size + self._init_feature_size
        self.encoder = Model(model_config=pipeline_config.encoder, input_dim=encoder_input_dim).to(device)
        self.processor = Model(model_config=pipeline_config.processor, input_dim=self.encoder.output_dim).to(device)
        self.decoder = Model(model_config=pipeline_config.decoder, input_dim=self.processor.output_dim).to(device)
        self.encoding_graph = self.encoding_graph.to(device)
        self.decoding_graph = self.decoding_graph.to(device)
        self.processing_graph = self.processing_graph.to(device)
        self._processing_edge_features = self._processing_edge_features.to(device)

    def _init_grid_properties(self, grid_lat, grid_lon, flat_grid=False):
        self._grid_lat = np.asarray(grid_lat).astype(np.float32)
        self._grid_lon = np.asarray(grid_lon).astype(np.float32)
        self._num_grid_nodes = len(grid_lat) if flat_grid else self._grid_lat.shape[0] * self._grid_lon.shape[0]

    def _init_mesh_properties(self, graph_config, region_bounds=None, mesh_buffer: float = 15.0):
        self._meshes = get_hierarchy_of_triangular_meshes_for_sphere(splits=max(graph_config.mesh_levels))
        if region_bounds is not None:
            lat_min, lat_max, lon_min, lon_max = region_bounds
            self._meshes = prune_mesh_to_region(self._meshes, lat_min, lat_max, lon_min, lon_max, buffer_deg=mesh_buffer)
        self._finest_mesh = self._meshes[-1]
        self._num_mesh_nodes = len(self._finest_mesh.vertices)
        lat, lon = get_mesh_lat_long(finest_mesh=self._finest_mesh)
        self._mesh_nodes_lat, self._mesh_nodes_lon = lat.astype(np.float32), lon.astype(np.float32)

    def _product_stage(self, X: torch.Tensor) -> torch.Tensor:
        """`src/models.py:823-828`: the [G, T*F] window is VIEWED as [T*G, F] (as the reference does - a plain
        view, no transpose), run through the GCN stack on the product graph, and the last G rows go on."""
        G, F = self._num_grid_nodes, self.num_features
        Xp = X.reshape(X.shape[:-2] + (G * self.obs_window, F))
        Xp = self.product_graph_model(X=Xp, edge_index=self.product_graph)
        return Xp[..., -G:, :]

    def _preprocess_input(self, grid_node_features: torch.Tensor):
        """`src/models.py:776-806` in one kernel: [grid dyn | grid static ; 0 | mesh static]."""
        return AssembleFn.apply(grid_node_features, self.init_grid_features, self.init_mesh_features)

    # --------------------------------------------------------------------------------------------
    # Compact pipeline: the same arithmetic per row, on fewer rows.
    #  * Encoder: a mesh row without grid in-edges only ever sees its own self-loop and its input
    #    is [0 | static] (src/models.py:792-801), so its whole encoder output is batch-invariant:
    #    those rows (8302 of 10242 at 64x32) are computed ONCE per forward instead of per sample.
    #  * Decoder: only grid rows are returned (src/models.py:870-872) and mesh rows only carry
    #    self-loops there, so mesh rows that send to no grid node are dead and are dropped.
    # Sub-graphs keep every in-edge (and its order) of every kept row and the in-degree of every
    # sender, so gcn_norm weights and summation order per row are unchanged.
    # --------------------------------------------------------------------------------------------
    def _compact_eligible(self) -> bool:
        if os.environ.get("GCL_NO_COMPACT", "0") not in ("0", ""):
            return False
        def ok(m):
            # graph-mode LayerNorm statistics couple every row of a sample, so no row of that stage
            # may be dropped or shared across the batch
            node_ln = all(ln.mode == "node" for ln in m.modules() if isinstance(ln, LayerNorm))
            return m.graph_layer.layer_type == GraphLayerType.ConvGCN and node_ln

        return ok(self.encoder) and ok(self.decoder) and getattr(self, "compact", True)

    # --------------------------------------------------------------------------------------------
    # Mesh rows in tile order (compact pipeline, GCN processor).  Mesh nodes never leave the model, so the processor
    # may number its rows freely: rows are listed patch by patch (mesh.tile_order), the permutation rides in the row
    # maps of the two stage gathers that exist anyway, and the processing graph is the reference's edge list with
    # renamed end points - every row keeps the order of its in-edges, so each sum runs in the reference's order.
    # What it buys: a 64-row tile of the [3, 5] mesh reads 108 distinct source rows instead of 211, which lets the
    # aggregation stage a tile's sources once in LDS (gcl_graph_halo_info, csrc/aggregate.hip).
    # --------------------------------------------------------------------------------------------
    _renumber_mesh = os.environ.get("GCL_NO_RENUMBER", "0") in ("0", "")

    def _mesh_order(self):
        """(order, pos) int64 CPU tensors: order[new] = old, pos[old] = new; None when the processor keeps the
        reference numbering (any processor but a GCN or GAT stack; GCL_NO_RENUMBER=1)."""
        if not self._renumber_mesh or self.processor.graph_layer.layer_type not in (
                GraphLayerType.ConvGCN, GraphLayerType.GATConv, GraphLayerType.SparseGATConv):
            return None
        cached = getattr(self, "_mesh_perm", None)
        if cached is None:
            M = self._num_mesh_nodes
            ei = self.processing_graph.detach().cpu()
            deg = torch.bincount(ei[1], minlength=M).numpy()
            order = torch.from_numpy(np.ascontiguousarray(tile_order(self._finest_mesh.vertices, 64, degree=deg))).to(torch.int64)
            pos = torch.empty(M, dtype=torch.int64)
            pos[order] = torch.arange(M)
            cached = self._mesh_perm = (order, pos)
        return cached

    def processor_graph(self) -> torch.Tensor:
        """The edge list the processor is run on: `processing_graph` itself, or (compact pipeline with a GCN
        processor) the same edges with mesh nodes renamed to tile order.  For callers that look the CSR handle up
        (bench.py's roofline probe); the model's public attributes keep the reference numbering."""
        if self._compact_eligible() and self._mesh_order() is not None and not self.using_interaction_net:
            return self._processing_graph_tiled()
        return self.processing_graph

    def _processing_graph_tiled(self):
        """The processing graph with mesh nodes renamed to tile order (same edge order); rebuilt only when
        `processing_graph` is replaced or modified."""
        g = self.processing_graph
        key = (id(g), g._version)
        c = getattr(self, "_proc_tiled", None)
        if c is None or c[0] != key:
            _, pos = self._mesh_order()
            c = self._proc_tiled = (key, pos.to(g.device)[g], g)  # keep g alive: its id must not be reused
        return c[1]

    def _processing_graph_from_tiled(self, tiled: torch.Tensor) -> torch.Tensor:
        """SparseGATConv hands back the (possibly pruned) edge list it ran on, self-loops included
        (src/models.py:846): the public `processing_graph` keeps the reference numbering, so the tile-order list is
        renamed back - once per distinct list; the pair is remembered in both directions, so the next forward finds
        the SAME tile-order tensor again and the CSR handle cache keeps hitting."""
        c = getattr(self, "_proc_tiled", None)
        if c is not None and c[1] is tiled:
            return c[2]
        order, _ = self._mesh_order()
        ref = order.to(tiled.device)[tiled]
        self._proc_tiled = ((id(ref), ref._version), tiled, ref)
        return ref

    def _compact_setup(self, device):
        G, M = self._num_grid_nodes, self._num_mesh_nodes
        enc, dec = self.encoding_graph.cpu(), self.decoding_graph.cpu()
        i32 = lambda t: t.to(torch.int32).contiguous().to(device)
        md = torch.unique(enc[1]) - G                       # mesh rows with grid in-edges (ascending)
        is_dep = torch.zeros(M, dtype=torch.bool)
        is_dep[md] = True
        mi = torch.nonzero(~is_dep).flatten()               # batch-invariant mesh rows
        used = torch.unique(dec[0]) - G                     # mesh rows some grid node reads in the decoder
        Md, Mi, U = md.numel(), mi.numel(), used.numel()
        remap_e = torch.full((G + M,), -1, dtype=torch.int64)
        remap_e[:G] = torch.arange(G)
        remap_e[G + md] = G + torch.arange(Md)
        remap_d = torch.full((G + M,), -1, dtype=torch.int64)
        remap_d[:G] = torch.arange(G)
        remap_d[G + used] = G + torch.arange(U)
        c = type("Compact", (), {})()
        c.Md, c.Mi, c.U = Md, Mi, U
        c.enc_graph = remap_e[enc].contiguous().to(device)
        c.dec_graph = remap_d[dec].contiguous().to(device)
        c.empty_graph = torch.zeros(2, 0, dtype=torch.int64, device=device)
        c.mstat_dep = self.init_mesh_features[md.to(self.init_mesh_features.device)].contiguous()
        Cdyn = self._dyn_size
        inv_stat = self.init_mesh_features[mi.to(self.init_mesh_features.device)]
        c.x_inv = torch.cat([torch.zeros(Mi, Cdyn, device=device), inv_stat.to(device)], dim=1).unsqueeze(0).contiguous()
        # mesh latents [M] <- encoder compact output [G+Md] (a) | invariant rows [Mi] (b)
        map_a = torch.full((M,), -1, dtype=torch.int64)
        map_a[md] = G + torch.arange(Md)
        map_b = torch.full((M,), -1, dtype=torch.int64)
        map_b[mi] = torch.arange(Mi)
        inv_a = torch.full((G + Md,), -1, dtype=torch.int64)
        inv_a[G:] = md
        perm = self._mesh_order()
        c.perm = perm
        if perm is not None:  # mesh rows of the processor are in tile order: forward maps are indexed by the NEW row
            order, pos = perm
            map_a, map_b = map_a[order], map_b[order]
            inv_a[G:] = pos[md]
        c.maps_mesh = (i32(map_a), i32(map_b), i32(inv_a), i32(mi if perm is None else perm[1][mi]))
        # decoder input [G+U] <- encoder compact output (a: grid rows) | processed mesh [M] (b)
        dmap_a = torch.full((G + U,), -1, dtype=torch.int64)
        dmap_a[:G] = torch.arange(G)
        dmap_b = torch.full((G + U,), -1, dtype=torch.int64)
        dmap_b[G:] = used
        dinv_a = torch.full((G + Md,), -1, dtype=torch.int64)
        dinv_a[:G] = torch.arange(G)
        dinv_b = torch.full((M,), -1, dtype=torch.int64)
        dinv_b[used] = G + torch.arange(U)
        if perm is not None:
            dmap_b[G:] = perm[1][used]
            dinv_b = dinv_b[perm[0]]
        c.maps_dec = (i32(dmap_a), i32(dmap_b), i32(dinv_a), i32(dinv_b))
        c.fold = {}
        c.mi = mi
        self._compact = c
        return c

    _fold_invariant_rows = os.environ.get("GCL_NO_FOLD", "0") in ("0", "")

    def _fold_setup(self, c, B: int, device):
        """Per batch size: r = ceil(Mi / B) batch-invariant mesh rows are appended to every sample of the compact
        encoder input as isolated nodes (GCNConv gives them their self-loop with weight 1, so each still sees only
        its own `[0 | static]` input - src/models.py:792-801), flat row j of that list sits in sample j // r."""
        G, M, Md, Mi = self._num_grid_nodes, self._num_mesh_nodes, c.Md, c.Mi
        r = -(-Mi // B)
        ne = G + Md + r
        Cs, Cdyn = self.init_mesh_features.shape[1], self._dyn_size
        f = type("Fold", (), {})()
        f.r, f.ne = r, ne
        stat = self.init_mesh_features.to(device)
        f.mstat = torch.cat([c.mstat_dep.to(device), torch.zeros(r, Cs, device=device)], dim=0).contiguous()
        x_fold = torch.zeros(B * r, Cdyn + Cs, device=device)
        x_fold[:Mi, Cdyn:] = stat[c.mi.to(device)]
        f.x_fold = x_fold.view(B, r, Cdyn + Cs)
        i32 = lambda t: t.to(torch.int32).contiguous().to(device)
        j = torch.arange(Mi)
        map_b = torch.full((M,), -1, dtype=torch.int64)
        map_b[c.mi] = (j // r) * ne + G + Md + (j % r)
        inv_fold = torch.full((B * r,), -1, dtype=torch.int64)
        inv_fold[:Mi] = c.mi
        if c.perm is not None:
            map_b = map_b[c.perm[0]]
            inv_fold[:Mi] = c.perm[1][c.mi]
        f.maps = (c.maps_mesh[0], i32(map_b), c.maps_mesh[2], i32(inv_fold))
        # the same two maps as ONE row table for consumers that read the mesh latents through it (gcl_gcn_layer_fwd_tab):
        # entry >= 0: row of the sample's own encoder output, entry < 0: ~(flat row of the batch-invariant list)
        f.tab = torch.where(f.maps[0] >= 0, f.maps[0], -f.maps[1] - 1).to(torch.int32).contiguous()
        # The encoder's MLP is row-wise and EVERY mesh row enters it as [0 | static] (src/models.py:792-801): the Md
        # batch-dependent mesh rows differ between samples only after the encoder's first GCNConv.  So the MLP runs on
        # [G grid | r folded invariant | rd = ceil(Md / B) folded dependent] rows per sample (2209 instead of 4118 at
        # 64x32, B = 64) and the conv stack's input [G | Md | r] is put together from its output: grid and invariant rows
        # from the sample's own rows, a dependent row from the flat list (shared by all samples; its gradient is the batch
        # sum) - the op MeshLatFn already is, with these maps.
        rd = -(-Md // B) if Md > 0 else 0
        f.rd, f.nm = rd, G + r + rd
        x_dep = torch.zeros(B * rd, Cdyn + Cs, device=device)
        x_dep[:Md, Cdyn:] = c.mstat_dep.to(device)
        f.x_tail2 = torch.cat([f.x_fold, x_dep.view(B, rd, Cdyn + Cs)], dim=1).contiguous()
        f.mstat2 = torch.zeros(r + rd, Cs, device=device)
        ea = torch.full((ne,), -1, dtype=torch.int64)
        ea[:G] = torch.arange(G)
        ea[G + Md:] = G + torch.arange(r)
        eb = torch.full((ne,), -1, dtype=torch.int64)
        d = torch.arange(Md)
        if rd > 0:
            eb[G: G + Md] = (d // rd) * f.nm + G + r + (d % rd)
        einv_a = torch.full((G + r,), -1, dtype=torch.int64)
        einv_a[:G] = torch.arange(G)
        einv_a[G:] = G + Md + torch.arange(r)
        einv_f = torch.full((B * rd,), -1, dtype=torch.int64)
        einv_f[:Md] = G + d
        f.maps_e = (i32(ea), i32(eb), i32(einv_a), i32(einv_f))
        # decoder-input maps: the encoder output now has ne rows per sample, the folded ones feed nothing there
        dinv_a = torch.cat([c.maps_dec[2], torch.full((r,), -1, dtype=torch.int32, device=device)]).contiguous()
        f.maps_dec = (c.maps_dec[0], c.maps_dec[1], dinv_a, c.maps_dec[3])
        c.fold[B] = f
        return f

    def _forward_compact(self, X: torch.Tensor, attention_threshold=0.0, **kwargs):
        G, M = self._num_grid_nodes, self._num_mesh_nodes
        squeeze = X.dim() == 2 or (X.dim() == 3 and X.shape[0] == 1)
        X3 = X if X.dim() == 3 else X.unsqueeze(0)
        B = X3.shape[0]
        if self.use_product_graph:
            X3 = self._product_stage(X3)
        c = getattr(self, "_compact", None) or self._compact_setup(X3.device)
        if self._fold_invariant_rows:
            # the Mi batch-invariant mesh rows ride through the SAME launches as r isolated nodes per sample; with Mi = 0
            # (every mesh node has grid senders, e.g. 512x256) the same path runs with r = 0, so the two gradient
            # consumers of the encoder output still share one landing buffer instead of autograd adding two full tensors
            f = c.fold.get(B) or self._fold_setup(c, B, X3.device)
            if self._mlp_on_folded_rows and self.encoder.mlp is not None and f.rd > 0:
                x_m = AssembleFn.apply(X3, self.init_grid_features, f.mstat2, f.x_tail2)   # [B, G+r+rd, C]
                h_m = self.encoder.mlp(X=x_m)                                              # [B, G+r+rd, D']
                x_g = MeshLatFn.apply(h_m, f.maps_e, f.ne, G + f.r, f.rd, None)            # [B, G+Md+r, D']
                enc_c = self.encoder.graph_layer(X=x_g, edge_index=c.enc_graph)            # [B, G+Md+r, D]
            else:
                x_c = AssembleFn.apply(X3, self.init_grid_features, f.mstat, f.x_fold if f.r > 0 else None)  # [B, G+Md+r, C]
                enc_c = self.encoder.forward(X=x_c, edge_index=c.enc_graph)         # [B, G+Md+r, D]
            land = GradLanding(G) if kwargs.pop("_landing", False) and self._grad_landing else None
            lat_src = self._lat_source(c, f, enc_c, land)
            if lat_src is None:
                mesh_lat = MeshLatFn.apply(enc_c, f.maps, M, G + c.Md, f.r, land)   # [B, M, D]
            maps_dec = f.maps_dec
        else:
            kwargs.pop("_landing", None)
            land = None
            lat_src = None
            maps_dec = c.maps_dec
            x_c = AssembleFn.apply(X3, self.init_grid_features, c.mstat_dep)           # [B, G+Md, C]
            enc_c = self.encoder.forward(X=x_c, edge_index=c.enc_graph)                 # [B, G+Md, D]
            inv = self.encoder.forward(X=c.x_inv, edge_index=c.empty_graph) if c.Mi > 0 else None  # [1, Mi, D]
            mesh_lat = Gather2Fn.apply(enc_c, inv, c.maps_mesh, M, B)                   # [B, M, D]
        if self.using_sparse_gat:
            pg = self._processing_graph_tiled() if c.perm is not None else self.processing_graph
            if land is not None and self._ln_into_decoder_input and self._latents_discarded and not squeeze:
                land.dec_buf = torch.empty(B, G + c.U, enc_c.shape[-1], dtype=torch.float32, device=enc_c.device)
                land.dec_map = maps_dec[3]
            processed, new_edge_index = self.processor.forward(
                X=mesh_lat, edge_index=pg, attention_threshold=attention_threshold,
                **({"_grad_src": land} if land is not None else {}), **kwargs)
            self.processing_graph = self._processing_graph_from_tiled(new_edge_index) if c.perm is not None else new_edge_index
        elif self.using_interaction_net:
            processed = self.processor.forward(X=mesh_lat, edge_index=self.processing_graph,
                                               attention_threshold=attention_threshold,
                                               edge_attr=self._processing_edge_features)
        else:
            pg = self._processing_graph_tiled() if c.perm is not None else self.processing_graph
            if (land is not None and self._ln_into_decoder_input and self._latents_discarded
                    and self.processor.graph_layer.layer_type in (GraphLayerType.ConvGCN, GraphLayerType.GATConv)
                    and not squeeze):
                # the processor's output is only consumed by the decoder-input gather: its final LayerNorm writes the
                # rows the decoder reads straight into the decoder's input (functional.GradLanding.dec_buf)
                land.dec_buf = torch.empty(B, G + c.U, enc_c.shape[-1], dtype=torch.float32, device=enc_c.device)
                land.dec_map = maps_dec[3]
            if lat_src is not None:
                # the first GCNConv reads the mesh latents THROUGH the row table from the encoder output: they are never
                # materialised, and its backward works on the compact rows (functional.GCNStackFn, LatSource)
                processed = self.processor.forward(X=enc_c, edge_index=pg, attention_threshold=attention_threshold,
                                                   _lat_src=lat_src, **({"_grad_src": land} if land is not None else {}))
            else:
                processed = self.processor.forward(X=mesh_lat, edge_index=pg, attention_threshold=attention_threshold,
                                                   **({"_grad_src": land} if land is not None else {}))
        dec_in = Gather2Fn.apply(enc_c, processed, maps_dec, G + c.U, B, land)      # [B, G+U, D]
        gcn_dec = self.decoder.graph_layer.layer_type == GraphLayerType.ConvGCN
        decoded = self.decoder.forward(X=dec_in, edge_index=c.dec_graph, **({"_out_rows": G} if gcn_dec else {}))
        out, grid_lat = (decoded if gcn_dec else decoded[:, :G, :]), enc_c[:, :G, :]
        if c.perm is not None and not self._want_prediction_only:
            processed = processed.index_select(1, c.perm[1].to(processed.device))  # callers see reference row order
        if squeeze:
            return out[0], grid_lat[0], processed[0]
        return out, grid_lat, processed

    _lat_through_table = os.environ.get("GCL_NO_LAT_TABLE", "0") in ("0", "")
    _mlp_on_folded_rows = os.environ.get("GCL_NO_MLP_FOLD", "0") in ("0", "")
    _ln_into_decoder_input = os.environ.get("GCL_NO_LN_MAP", "0") in ("0", "")
    _latents_discarded = False  # True only inside forward(): nobody sees the processor's output

    def _lat_source(self, c, f, enc_c, land):
        """functional.LatSource when the processor's first layer can read the mesh latents through the row table
        (a GCNConv stack on a source-tile mesh graph, gcl_gcn_layer_fwd_tab), else None."""
        if not self._lat_through_table or c.perm is None or self.processor.mlp is not None:
            return None
        gl = self.processor.graph_layer
        if gl.layer_type == GraphLayerType.GATConv:
            first = gl.layers[0]
            if not isinstance(first, GATConv) or first.heads != 1 or enc_c.dim() != 3 or first.lin.weight.shape[1] != enc_c.shape[-1]:
                return None
            g = _graphs.get(self._processing_graph_tiled(), self._num_mesh_nodes, hip.GRAPH_GAT)
            if not hip.gat_tab_ok(g, 1, first.lin.weight.shape[0]):
                return None
            return LatSource(f.tab, f.maps, self._num_mesh_nodes, self._num_grid_nodes, c.Md, f.r, land)
        if gl.layer_type != GraphLayerType.ConvGCN:
            return None
        convs = [m for m in gl.layers if isinstance(m, GCNConv)]
        if not convs or enc_c.dim() != 3 or convs[0].lin.weight.shape[1] != enc_c.shape[-1]:
            return None
        g = _graphs.get(self._processing_graph_tiled(), self._num_mesh_nodes, hip.GRAPH_GCN)
        if not hip.gcn_layer_tab_ok(g, enc_c, convs[0].lin.weight.shape[0]):
            return None
        return LatSource(f.tab, f.maps, self._num_mesh_nodes, self._num_grid_nodes, c.Md, f.r, land)

    _want_prediction_only = False

    def forward_with_latents(self, X: torch.Tensor, attention_threshold=0.0, **kwargs):
        landing = kwargs.pop("_landing", False)
        self._want_prediction_only = landing  # forward(): the mesh latents are dropped, so they stay in tile order
        if self._compact_eligible():
            return self._forward_compact(X, attention_threshold, _landing=landing, **kwargs)
        G = self._num_grid_nodes
        if X.dim() == 3 and X.shape[0] == 1:
            X = X.squeeze(0)  # reference: X.squeeze() with batch 1 (src/models.py:822)
        if self.use_product_graph:
            X = self._product_stage(X)
        X = self._preprocess_input(grid_node_features=X)
        encoded = self.encoder.forward(X=X, edge_index=self.encoding_graph)
        grid_node_features = encoded[..., :G, :]
        mesh_node_features = encoded[..., G:, :]
        if self.using_sparse_gat:
            processed, new_edge_index = self.processor.forward(
                X=mesh_node_features, edge_index=self.processing_graph, attention_threshold=attention_threshold,
                **kwargs)
            self.processing_graph = new_edge_index
        elif self.using_interaction_net:
            processed = self.processor.forward(
                X=mesh_node_features, edge_index=self.processing_graph, attention_threshold=attention_threshold,
                edge_attr=self._processing_edge_features)
        else:
            processed = self.processor.forward(
                X=mesh_node_features, edge_index=self.processing_graph, attention_threshold=attention_threshold)
        processed_features = torch.cat((grid_node_features, processed), dim=-2)
        if self.decoder.graph_layer.layer_type == GraphLayerType.ConvGCN:  # the row slice happens inside the stack
            decoded = self.decoder.forward(X=processed_features, edge_index=self.decoding_graph, _out_rows=G)
            return decoded, grid_node_features, processed
        decoded = self.decoder.forward(X=processed_features, edge_index=self.decoding_graph)
        return decoded[..., :G, :], grid_node_features, processed

    _grad_landing = os.environ.get("GCL_NO_LANDING", "0") in ("0", "")

    def forward(self, X: torch.Tensor, attention_threshold=0.0, **kwargs):
        # only the prediction leaves this call, so the compact encoder output has exactly two gradient consumers and
        # they may share one gradient buffer (functional.GradLanding)
        # ... and the processor's output itself is never seen by the caller, so its LayerNorm may write only the rows the
        # decoder reads, straight into the decoder's input (the `processed` the inner call returns is then a zero token)
        self._latents_discarded = True
        try:
            return self.forward_with_latents(X, attention_threshold, _landing=True, **kwargs)[0]
        finally:
            self._latents_discarded = False
