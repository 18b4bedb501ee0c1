"""The experiment configurations BASELINE.json names, restated as `ExperimentConfig` objects.

Values are those of the reference's `experiments/{baseline,attention,sparse_attention,
wb2_512x256_19f_ar,wb2_512x256_19f_ar_v2,region_krsk_cds_19f}/config.json` (checked against `tests/golden/config_parse.json`, which was
produced by parsing the reference's files with the reference's own pydantic schema)."""
from .config import ExperimentConfig


def _pipeline(enc_hidden, F, proc_type, proc_hidden, dec_mlp_hidden, dec_mlp_out, dec_hidden, out_dim, heads=1,
              enc_dec_type="conv_gcn"):
    gat = {"num_heads": heads, "sparsity_thresholds": [0.0, 0.0]}
    return {
        "encoder": {"mlp": {"mlp_hidden_dims": enc_hidden, "output_dim": F, "use_layer_norm": True,
                            "layer_norm_mode": "node"},
                    "gcn": {"layer_type": enc_dec_type, "hidden_dims": [F, F], "output_dim": F}},
        "processor": {"gcn": {"layer_type": proc_type, "gat_props": gat, "hidden_dims": proc_hidden,
                              "output_dim": F, "use_layer_norm": True, "layer_norm_mode": "node"}},
        "decoder": {"mlp": {"mlp_hidden_dims": dec_mlp_hidden, "output_dim": dec_mlp_out, "use_layer_norm": False},
                    "gcn": {"layer_type": enc_dec_type, "hidden_dims": dec_hidden, "output_dim": out_dim}},
    }


def _interaction_pipeline(F, steps, enc_hidden, dec_mlp_hidden, dec_mlp_out, dec_hidden, out_dim, act="swish"):
    """The InteractionNet family (`experiments/wb2_512x256_19f_ar_v2`, `multires_*`, `region_*`)."""
    return {
        "encoder": {"mlp": {"mlp_hidden_dims": enc_hidden, "output_dim": F, "use_layer_norm": True,
                            "layer_norm_mode": "node"},
                    "gcn": {"layer_type": "conv_gcn", "hidden_dims": [F, F], "output_dim": F, "activation": act}},
        "processor": {"gcn": {"layer_type": "interaction_net", "output_dim": F, "activation": act,
                              "use_layer_norm": True, "num_message_passing_steps": steps, "edge_feature_dim": 4}},
        "decoder": {"mlp": {"mlp_hidden_dims": dec_mlp_hidden, "output_dim": dec_mlp_out, "use_layer_norm": False},
                    "gcn": {"layer_type": "conv_gcn", "hidden_dims": dec_hidden, "output_dim": out_dim,
                            "activation": act}},
    }


GRID = {"product_graph": (32, 64), "wb2_64x32_15f": (32, 64), "wb2_64x32_15f_gat": (32, 64), "wb2_64x32_ar_15f_4obs_4pred": (32, 64), "demo_low": (32, 64),
        "wb2_512x256_sparse_gat": (256, 512), "wb2_512x256_19f_ar_v2": (256, 512), "region_krsk_cds_19f": (32, 64), "baseline": (32, 64), "attention": (32, 64), "attention_h4": (32, 64), "sparse_attention": (32, 64),
        "wb2_512x256_19f_ar": (256, 512)}


def experiment(name: str, mesh_levels=None) -> ExperimentConfig:
    graph = {"grid2mesh_edge_creation": "radius", "mesh2grid_edge_creation": "contained",
             "grid2mesh_radius_query": 0.5, "mesh_levels": mesh_levels or [3, 5]}
    data = {"dataset_name": "synthetic", "num_features_used": 33, "obs_window_used": 2, "pred_window_used": 1,
            "want_feats_flattened": True}
    if name == "baseline":
        pipe = _pipeline([48, 48], 64, "conv_gcn", [64, 64], [64, 64], 64, [48, 48], 33)
    elif name == "attention":
        pipe = _pipeline([48, 48], 64, "conv_gat", [64, 64], [64, 64], 64, [48, 48], 33)
    elif name == "attention_h4":  # README.md:148-150 also reports 4 heads
        pipe = _pipeline([48, 48], 64, "conv_gat", [64, 64], [64, 64], 64, [48, 48], 33, heads=4)
    elif name == "sparse_attention":
        pipe = _pipeline([48, 48], 64, "sparse_gat", [], [64, 64], 12, [48, 48], 12, enc_dec_type="simple_conv")
        data.update(num_features_used=12)
    elif name == "wb2_512x256_19f_ar":
        pipe = _pipeline([128, 128], 128, "conv_gcn", [128] * 4, [128, 64], 64, [64, 64], 19)
        graph.update(grid2mesh_radius_query=0.6, mesh_levels=mesh_levels or [4, 6])
        data.update(num_features_used=19)
    elif name in ("wb2_64x32_15f", "wb2_64x32_ar_15f_4obs_4pred", "wb2_64x32_15f_gat"):  # 4 observed steps, 96-wide stacks
        # "_gat": the same widths with a GATConv processor (two heads of 96 channels: C / 4 = 24 is not a power of two)
        gat = name.endswith("_gat")
        pipe = _pipeline([64, 64], 64, "conv_gat" if gat else "conv_gcn", [96, 96, 96], [64, 64], 64, [48, 48], 15,
                         **({"heads": 2} if gat else {}))
        pipe["encoder"]["gcn"].update(hidden_dims=[96, 96], output_dim=96)
        pipe["processor"]["gcn"].update(output_dim=96)
        graph.update(grid2mesh_radius_query=0.65, mesh_levels=mesh_levels or [4, 6])
        data.update(num_features_used=15, obs_window_used=4,
                    pred_window_used=4 if name.endswith("4pred") else 1)
    elif name == "product_graph":  # experiments/product_graph: 5 observed steps through a time x space GCN first
        pipe = _pipeline([48, 48], 64, "conv_gcn", [64, 64], [64, 64], 64, [48, 48], 33)
        pipe["product_graph"] = {"model": {"gcn": {"layer_type": "conv_gcn", "hidden_dims": [33, 33], "output_dim": 33,
                                                   "use_layer_norm": True, "layer_norm_mode": "node"}},
                                 "num_k": 4, "self_loop": True, "type": "strong"}
        data.update(obs_window_used=5)
    elif name == "demo_low":  # the smallest config: widths 16 / 32, 3 variables, one mesh level
        pipe = {
            "encoder": {"mlp": {"mlp_hidden_dims": [16], "output_dim": 32, "use_layer_norm": True, "layer_norm_mode": "node"},
                        "gcn": {"layer_type": "conv_gcn", "hidden_dims": [32], "output_dim": 32}},
            "processor": {"gcn": {"layer_type": "conv_gcn", "hidden_dims": [32], "output_dim": 32}},
            "decoder": {"mlp": {"mlp_hidden_dims": [32], "output_dim": 32, "use_layer_norm": False},
                        "gcn": {"layer_type": "conv_gcn", "hidden_dims": [16], "output_dim": 3}},
        }
        graph.update(mesh_levels=mesh_levels or [3])
        data.update(num_features_used=3)
    elif name == "wb2_512x256_sparse_gat":
        # BASELINE.json configs[4]: the 512x256 encoder / decoder with ONE SparseGATConv(128 -> 128, H = 1) + LN as
        # processor.  The reference has no config file for this combination.
        pipe = _pipeline([128, 128], 128, "sparse_gat", [], [128, 64], 64, [64, 64], 19)
        graph.update(grid2mesh_radius_query=0.6, mesh_levels=mesh_levels or [4, 6])
        data.update(num_features_used=19)
    elif name == "wb2_512x256_19f_ar_v2":  # latent 256, 12 unshared InteractionNet steps, swish
        pipe = _interaction_pipeline(256, 12, [256, 256], [256, 128], 128, [128, 128], 19)
        graph.update(grid2mesh_radius_query=0.6, mesh_levels=mesh_levels or [4, 6])
        data.update(num_features_used=19)
    elif name == "region_krsk_cds_19f":  # latent 128, 8 steps (its regional grid is replaced by GRID's)
        pipe = _interaction_pipeline(128, 8, [128, 128], [128, 64], 64, [64, 64], 19)
        graph.update(grid2mesh_radius_query=0.6, mesh_levels=mesh_levels or [3, 5])
        data.update(num_features_used=19)
    else:
        raise KeyError(name)
    return ExperimentConfig(graph=graph, pipeline=pipe, data=data)
