"""CPU restatement of the reference's module tree (`src/models.py`) on top of `oracle.pyg_ops`.

TEST INFRASTRUCTURE ONLY (see `oracle/pyg_ops.py` header; parity of the PyG-backed layers is
UNPINNED - no PyG in the image, no reference tests).  Class wiring, attribute names and
state-dict keys follow the reference so that weights can be exchanged with the product modules
by `state_dict()` / `load_state_dict()`:

  MLP            `src/models.py:54-109`   keys `MLP.{i}.weight|bias`
  GraphLayer     `src/models.py:289-440`  keys `activation.weight`, `layers.{i}.lin.weight`,
                                          `layers.{i}.bias`, `layers.{i}.att_src|att_dst`,
                                          `layers.{odd}.weight` (alias of the shared PReLU),
                                          LayerNorm at `layers.{last}.weight|bias`
  InteractionNet `src/models.py:166-285`  `layers.edge_encoder.0.*`, `layers.steps.{k}.edge_mlp.{0,2}.*`,
                                          `layers.steps.{k}.node_mlp.{0,2}.*`, `...edge_norm.*`, `...node_norm.*`
  Model          `src/models.py:443-473`  `mlp.*`, `graph_layer.*`
  WeatherPrediction `src/models.py:476-927` `encoder.*`, `processor.*`, `decoder.*`,
                                          buffer `_processing_edge_features`

Differences from the reference, all deliberate and documented in SURVEY.md Appendix B:
  * graphs and static features are passed in (built by the caller) instead of being built in the
    constructor, and the three `summary()` forward passes are skipped;
  * a batch dimension is supported: `[B, G, C]` is B independent samples; `[1, G, C]` / `[G, C]`
    follow the reference's squeeze semantics and return `[G, C_out]`.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from . import pyg_ops as P


def _glorot(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class OLayerNorm(nn.Module):
    def __init__(self, in_channels: int, mode: Optional[str] = "graph", eps: float = 1e-5):
        super().__init__()
        self.mode = mode or "graph"
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(in_channels))
        self.bias = nn.Parameter(torch.zeros(in_channels))

    def forward(self, x):
        return P.pyg_layer_norm(x, self.weight, self.bias, self.mode, self.eps)


class OGCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.lin.weight)

    def forward(self, x, edge_index):
        return P.gcn_conv(x, edge_index, self.lin.weight, self.bias)


class OGATConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: int = 1):
        super().__init__()
        self.heads = heads
        self.out_channels = out_channels
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.lin.weight)
        _glorot(self.att_src)
        _glorot(self.att_dst)

    def forward(self, x, edge_index, return_attention_weights: bool = False):
        y, ei, alpha = P.gat_conv(x, edge_index, self.lin.weight, self.att_src, self.att_dst, self.bias, self.heads)
        if return_attention_weights:
            return y, (ei, alpha)
        return y


class OSparseGATConv(OGATConv):
    """`src/models.py:112-151`."""

    def forward(self, x, edge_index, attention_threshold=0.0, **kwargs):
        batch_num = kwargs.get("batch_num", 1)
        out, (ei, alpha) = super().forward(x, edge_index, return_attention_weights=True)
        if alpha.dim() == 3:   # batched: the prune decision comes from sample 0
            alpha = alpha[0]
        alpha = alpha.squeeze()
        if batch_num == 0:
            ei, alpha = P.sparse_gat_prune(ei, alpha, attention_threshold)
        return out, (ei, alpha)


class OSimpleConv(nn.Module):
    def forward(self, x, edge_index):
        return P.simple_conv_mean(x, edge_index)


def get_activation(name: str = "prelu") -> nn.Module:
    """`src/models.py:154-163`."""
    if name in ("swish", "silu"):
        return nn.SiLU()
    if name == "prelu":
        return nn.PReLU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(f"Unknown activation: {name}")


class OInteractionNetLayer(nn.Module):
    """One message-passing step, `src/models.py:166-236`: edge update from [sender | receiver | edge],
    mean over incoming edges, node update from [node | aggregate], residuals, then graph-mode LN on
    the edges and node-mode LN on the nodes."""

    def __init__(self, node_dim: int, edge_dim: int, hidden_dim: int, activation: str = "swish",
                 use_layer_norm: bool = True):
        super().__init__()
        act = get_activation(activation)  # one instance inside both MLPs (src/models.py:184-198)
        self.edge_mlp = nn.Sequential(nn.Linear(2 * node_dim + edge_dim, hidden_dim), act, nn.Linear(hidden_dim, edge_dim))
        self.node_mlp = nn.Sequential(nn.Linear(node_dim + edge_dim, hidden_dim), act, nn.Linear(hidden_dim, node_dim))
        self.use_layer_norm = use_layer_norm
        if use_layer_norm:
            self.edge_norm = OLayerNorm(edge_dim, mode="graph")
            self.node_norm = OLayerNorm(node_dim, mode="node")

    def forward(self, x, edge_index, edge_attr):
        snd, rcv = edge_index[0], edge_index[1]
        n = x.shape[-2]
        upd_e = self.edge_mlp(torch.cat([x[..., snd, :], x[..., rcv, :], edge_attr], dim=-1))
        agg = P.scatter_mean_rows(upd_e, rcv, n)
        upd_x = self.node_mlp(torch.cat([x, agg], dim=-1))
        new_e, new_x = edge_attr + upd_e, x + upd_x
        if self.use_layer_norm:
            new_e, new_x = self.edge_norm(new_e), self.node_norm(new_x)
        return new_x, new_e


class OInteractionNetProcessor(nn.Module):
    """`src/models.py:239-285`: raw edge features -> latent (Linear + activation), then N unshared steps."""

    def __init__(self, node_dim, raw_edge_dim, edge_latent_dim, hidden_dim, num_steps, activation="swish",
                 use_layer_norm=True):
        super().__init__()
        self.edge_encoder = nn.Sequential(nn.Linear(raw_edge_dim, edge_latent_dim), get_activation(activation))
        self.steps = nn.ModuleList([
            OInteractionNetLayer(node_dim, edge_latent_dim, hidden_dim, activation, use_layer_norm)
            for _ in range(num_steps)])

    def forward(self, x, edge_index, edge_attr_raw):
        e = self.edge_encoder(edge_attr_raw)
        if x.dim() == 3:  # batched samples share the raw edge features
            e = e.unsqueeze(0).expand(x.shape[0], -1, -1)
        for step in self.steps:
            x, e = step(x, edge_index, e)
        return x


class MLP(nn.Module):
    def __init__(self, mlp_config, input_dim: int):
        super().__init__()
        hidden = list(mlp_config.mlp_hidden_dims or [])
        self.MLP = nn.ModuleList()
        d = input_dim
        for hdim in hidden:
            self.MLP.extend([nn.Linear(d, hdim), nn.PReLU()])
            d = hdim
        self.MLP.append(nn.Linear(d, mlp_config.output_dim))
        if mlp_config.use_layer_norm:
            self.MLP.append(OLayerNorm(mlp_config.output_dim, mode=mlp_config.layer_norm_mode))

    def forward(self, X):
        for layer in self.MLP:
            X = layer(X)
        return X


class GraphLayer(nn.Module):
    def __init__(self, graph_config, input_dim: int):
        super().__init__()
        lt = getattr(graph_config.layer_type, "value", graph_config.layer_type)
        self.layer_type = lt
        if lt == "simple_conv":
            self.output_dim = input_dim
            self.layers = OSimpleConv()
            return
        if lt == "interaction_net":  # src/models.py:376-398
            self.output_dim = graph_config.output_dim
            assert graph_config.output_dim == input_dim, "InteractionNet requires output_dim == input_dim (residuals)"
            use_ln = graph_config.use_layer_norm if graph_config.use_layer_norm is not None else True
            self.layers = OInteractionNetProcessor(
                node_dim=input_dim, raw_edge_dim=graph_config.edge_feature_dim or 4, edge_latent_dim=input_dim,
                hidden_dim=input_dim, num_steps=graph_config.num_message_passing_steps or 4,
                activation=graph_config.activation or "swish", use_layer_norm=use_ln)
            return
        if lt not in ("conv_gcn", "conv_gat", "sparse_gat"):
            raise NotImplementedError(f"Layer type {graph_config.layer_type} not supported.")
        self.activation = get_activation(graph_config.activation or "prelu")
        self.output_dim = graph_config.output_dim
        self.layers = nn.ModuleList()
        hidden = list(graph_config.hidden_dims or [])
        if lt == "sparse_gat":
            self.num_heads = graph_config.gat_props.num_heads
            self.layers.append(OSparseGATConv(input_dim, graph_config.output_dim, heads=self.num_heads))
        else:
            if lt == "conv_gat":
                self.num_heads = graph_config.gat_props.num_heads
                mk = lambda i, o: OGATConv(i, o, heads=self.num_heads)
            else:
                mk = OGCNConv
            dims = [input_dim] + hidden + [graph_config.output_dim]
            for k in range(len(dims) - 1):
                self.layers.append(mk(dims[k], dims[k + 1]))
                if k < len(dims) - 2:
                    self.layers.append(self.activation)  # ONE shared PReLU instance
        if graph_config.use_layer_norm:
            self.layers.append(OLayerNorm(graph_config.output_dim, mode=graph_config.layer_norm_mode))

    def forward(self, X, edge_index, attention_threshold=0.0, **kwargs):
        if self.layer_type == "simple_conv":
            return self.layers(X, edge_index)
        if self.layer_type == "interaction_net":
            edge_attr = kwargs.get("edge_attr", None)
            if edge_attr is None:
                raise ValueError("InteractionNet requires edge_attr (edge features)")
            return self.layers(X, edge_index, edge_attr)
        if self.layer_type == "sparse_gat":
            for layer in self.layers:
                if isinstance(layer, OSparseGATConv):
                    X, (edge_index, _) = layer(X, edge_index, attention_threshold, **kwargs)
                else:
                    X = layer(X)
            return X, edge_index
        for layer in self.layers:
            if isinstance(layer, (OGCNConv, OGATConv)):
                X = layer(X, edge_index)
            else:
                X = layer(X)
        return X


class Model(nn.Module):
    def __init__(self, model_config, input_dim: int):
        super().__init__()
        self.mlp = None
        gin = input_dim
        if model_config.mlp:
            self.mlp = MLP(model_config.mlp, input_dim)
            gin = model_config.mlp.output_dim
        self.graph_layer = GraphLayer(model_config.gcn, gin)
        self.output_dim = self.graph_layer.output_dim

    def forward(self, X, edge_index, attention_threshold=0.0, **kwargs):
        if self.mlp:
            X = self.mlp(X)
        return self.graph_layer(X, edge_index, attention_threshold=attention_threshold, **kwargs)


class WeatherPrediction(nn.Module):
    """Restatement of `src/models.py:476-927` minus graph construction and product graph."""

    def __init__(self, pipeline_config, data_config, *, num_grid_nodes, num_mesh_nodes, encoding_graph,
                 processing_graph, decoding_graph, init_grid_features, init_mesh_features,
                 processing_edge_features=None, product_graph=None):
        super().__init__()
        self.obs_window = data_config.obs_window_used
        self.num_features = data_config.num_features_used
        self.total_feature_size = self.num_features * self.obs_window
        self._num_grid_nodes, self._num_mesh_nodes = num_grid_nodes, num_mesh_nodes
        self.encoding_graph, self.processing_graph, self.decoding_graph = (
            encoding_graph, processing_graph, decoding_graph)
        self.init_grid_features, self.init_mesh_features = init_grid_features, init_mesh_features
        if processing_edge_features is not None:
            self.register_buffer("_processing_edge_features", processing_edge_features)
        else:
            self._processing_edge_features = None
        lt = pipeline_config.processor.gcn.layer_type
        self.using_sparse_gat = getattr(lt, "value", lt) == "sparse_gat"
        self.using_interaction_net = getattr(lt, "value", lt) == "interaction_net"
        self.use_product_graph = product_graph is not None  # src/models.py:505,517-524
        if self.use_product_graph:
            self.product_graph = product_graph
            self.product_graph_model = Model(pipeline_config.product_graph.model, self.num_features)
        dyn = self.num_features if self.use_product_graph else self.total_feature_size
        self._dyn_size = dyn
        enc_in = dyn + init_grid_features.shape[1]
        self.encoder = Model(pipeline_config.encoder, enc_in)
        self.processor = Model(pipeline_config.processor, self.encoder.output_dim)
        self.decoder = Model(pipeline_config.decoder, self.processor.output_dim)

    def _preprocess_input(self, grid_node_features):
        """`src/models.py:776-806`: [grid dyn | grid static ; 0 | mesh static]."""
        lead = grid_node_features.shape[:-2]
        gs = self.init_grid_features.to(grid_node_features.dtype).expand(lead + self.init_grid_features.shape)
        ms = self.init_mesh_features.to(grid_node_features.dtype).expand(lead + self.init_mesh_features.shape)
        g = torch.cat((grid_node_features, gs), dim=-1)
        zeros = torch.zeros(lead + (self._num_mesh_nodes, self._dyn_size), dtype=g.dtype)
        m = torch.cat((zeros, ms), dim=-1)
        return torch.cat((g, m), dim=-2)

    def forward_with_latents(self, X, attention_threshold=0.0, **kwargs):
        if X.dim() == 3 and X.shape[0] == 1:
            X = X.squeeze(0)
        G = self._num_grid_nodes
        if self.use_product_graph:  # src/models.py:823-828: a plain view to [T*G, F], GCN, last G rows
            Xp = X.reshape(X.shape[:-2] + (G * self.obs_window, self.num_features))
            X = self.product_graph_model(Xp, self.product_graph)[..., -G:, :]
        X = self._preprocess_input(X)
        enc = self.encoder(X, self.encoding_graph)
        grid_lat, mesh_lat = enc[..., :G, :], enc[..., G:, :]
        if self.using_sparse_gat:
            mesh_out, new_graph = self.processor(
                mesh_lat, self.processing_graph, attention_threshold=attention_threshold, **kwargs)
            self.processing_graph = new_graph
        elif self.using_interaction_net:
            mesh_out = self.processor(mesh_lat, self.processing_graph, attention_threshold=attention_threshold,
                                      edge_attr=self._processing_edge_features)
        else:
            mesh_out = self.processor(mesh_lat, self.processing_graph, attention_threshold=attention_threshold)
        dec = self.decoder(torch.cat((grid_lat, mesh_out), dim=-2), self.decoding_graph)
        return dec[..., :G, :], grid_lat, mesh_out

    def forward(self, X, attention_threshold=0.0, **kwargs):
        return self.forward_with_latents(X, attention_threshold, **kwargs)[0]
