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
    grid2mesh_edge_creation: Grid2MeshEdgeCreation
    grid2mesh_radius_query: Optional[float] = None
    grid2mesh_k: Optional[int] = None
    mesh_levels: List[int]
    mesh2grid_edge_creation: Mesh2GridEdgeCreation


class MLPBlock(BaseModel):
    mlp_hidden_dims: Optional[List[int]] = None
    output_dim: int
    use_layer_norm: bool
    layer_norm_mode: Optional[str] = None


class GATProps(BaseModel):
    num_heads: int
    sparsity_thresholds: List[float]


class GraphBlock(BaseModel):
    layer_type: GraphLayerType
    gat_props: Optional[GATProps] = None
    hidden_dims: Optional[List[int]] = None
    output_dim: Optional[int] = None
    use_layer_norm: Optional[bool] = None
    layer_norm_mode: Optional[str] = None
    activation: Optional[str] = "prelu"
    num_message_passing_steps: Optional[int] = None
    edge_feature_dim: Optional[int] = None


class ModelConfig(BaseModel):
    mlp: Optional[MLPBlock] = None
    gcn: GraphBlock


class ProductGraphConfig(BaseModel):
    model: ModelConfig
    num_k: int
    self_loop: bool
    type: ProductGraphType


class PipelineConfig(BaseModel):
    product_graph: Optional[ProductGraphConfig] = None
    encoder: ModelConfig
    processor: ModelConfig
    decoder: ModelConfig


class DataConfig(BaseModel):
    # the reference restricts dataset_name to an enum of its own dataset directories
    # (`src/config.py:50-65`); the hot path never reads it, so any string is accepted here.
    dataset_name: str
    num_features_used: int
    obs_window_used: int
    pred_window_used: int
    want_feats_flattened: bool


class ExperimentConfig(BaseModel):
    batch_size: int = 1
    learning_rate: float = 1e-5
    early_stopping_patience: int = 10
    early_stopping_delta: float = 1e-4
    num_epochs: int = 100
    random_seed: Optional[int] = 42
    graph: GraphBuildingConfig
    pipeline: PipelineConfig
    data: DataConfig
    wandb_log: bool = True
    wandb_name: Optional[str] = None
    use_latitude_weighting: bool = True
    max_ar_steps: int = 1
    data_dir: Optional[str] = None
    static_channels: List[int] = []
    forcing_channels: List[int] = []
    roi_only_loss: bool = False
    boundary_mask_width: int = 0
    freeze_processor_epochs: int = 0
    finetune_processor_lr_factor: float = 0.1
    use_residual: bool = True
