"""Experiment configuration schema.

Restates the schema of the reference (`src/config.py:8-252`) so that the reference's
`experiments/*/config.json` files parse unchanged (including the `"True"`/`"False"` strings,
which pydantic's lax mode coerces).  Field names, enum values and defaults follow the reference;
the reference's hard-coded wandb key (`src/config.py:240`) is deliberately not reproduced.
"""
from enum import Enum
from typing import List, Optional

from pydantic import BaseModel


class Grid2MeshEdgeCreation(str, Enum):
    K_NEAREST = "k_nearest"
    RADIUS = "radius"


class Mesh2GridEdgeCreation(str, Enum):
    CONTAINED = "contained"


class GraphLayerType(str, Enum):
    ConvGCN = "conv_gcn"
    SimpleConv = "simple_conv"
    GATConv = "conv_gat"
    SparseGATConv = "sparse_gat"
    InteractionNet = "interaction_net"


class ProductGraphType(str, Enum):
    KRONECKER = "kronecker"
    CARTESIAN = "cartesian"
    STRONG = "strong"


class GraphBuildingConfig(BaseModel):
    """How the three graphs are built (mesh refinement levels, grid->mesh radius / k, mesh->grid rule)."""
    grid2mesh_edge_creation: Grid2MeshEdgeCreation
    grid2mesh_k: Optional[int] = None
    grid2mesh_radius_query: Optional[float] = None
    mesh2grid_edge_creation: Mesh2GridEdgeCreation
    mesh_levels: List[int]


class MLPBlock(BaseModel):
    """Per-node MLP in front of a graph layer."""
    layer_norm_mode: Optional[str] = None
    mlp_hidden_dims: Optional[List[int]] = None
    output_dim: int
    use_layer_norm: bool


class GATProps(BaseModel):
    """Attention heads and the (unused) sparsity schedule."""
    num_heads: int
    sparsity_thresholds: List[float]


class GraphBlock(BaseModel):
    """One graph layer stack: type, widths, normalisation, activation, InteractionNet options."""
    activation: Optional[str] = "prelu"
    edge_feature_dim: Optional[int] = None
    gat_props: Optional[GATProps] = None
    hidden_dims: Optional[List[int]] = None
    layer_norm_mode: Optional[str] = None
    layer_type: GraphLayerType
    num_message_passing_steps: Optional[int] = None
    output_dim: Optional[int] = None
    use_layer_norm: Optional[bool] = None


class ModelConfig(BaseModel):
    """Optional MLP followed by a graph layer."""
    gcn: GraphBlock
    mlp: Optional[MLPBlock] = None


class ProductGraphConfig(BaseModel):
    """Time x space pre-encoder."""
    model: ModelConfig
    num_k: int
    self_loop: bool
    type: ProductGraphType


class PipelineConfig(BaseModel):
    """Encoder / processor / decoder (+ optional product-graph stage)."""
    decoder: ModelConfig
    encoder: ModelConfig
    processor: ModelConfig
    product_graph: Optional[ProductGraphConfig] = None


class DataConfig(BaseModel):
    """Which slice of the dataset the model sees."""
    # the reference restricts dataset_name to an enum of its own dataset directories
    # (`src/config.py:50-65`); the hot path never reads it, so any string is accepted here.
    dataset_name: str
    num_features_used: int
    obs_window_used: int
    pred_window_used: int
    want_feats_flattened: bool


class ExperimentConfig(BaseModel):
    """Top level of an experiments/*/config.json."""
    batch_size: int = 1
    boundary_mask_width: int = 0
    data: DataConfig
    data_dir: Optional[str] = None
    early_stopping_delta: float = 1e-4
    early_stopping_patience: int = 10
    finetune_processor_lr_factor: float = 0.1
    forcing_channels: List[int] = []
    freeze_processor_epochs: int = 0
    graph: GraphBuildingConfig
    learning_rate: float = 1e-5
    max_ar_steps: int = 1
    num_epochs: int = 100
    pipeline: PipelineConfig
    random_seed: Optional[int] = 42
    roi_only_loss: bool = False
    static_channels: List[int] = []
    use_latitude_weighting: bool = True
    use_residual: bool = True
    wandb_log: bool = True
    wandb_name: Optional[str] = None
