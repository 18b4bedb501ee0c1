"""Pin the CPU oracle: fixtures produced by running the reference's own code (loss, schedule),
the reference authors' recorded known-answers (parameter counts, shapes) and the hand-derived
micro-cases of SURVEY.md Appendix A.7.  The PyG layers themselves stay "parity unpinned"."""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, build_graphs, experiment, oracle_model
from oracle import pyg_ops as P
from oracle import train_step as T


def test_loss_matches_reference_vectors():
    v = np.load(os.path.join(GOLDEN, "loss_vectors.npz"))
    pred, target = torch.from_numpy(v["pred"]), torch.from_numpy(v["target"])
    lat, chan, sm = torch.from_numpy(v["lat_w"]), torch.from_numpy(v["chan_mask"]), torch.from_numpy(v["spatial_mask"])
    np.testing.assert_array_equal(T.get_lat_weights(32, 64).numpy(), v["lat_w"])
    np.testing.assert_array_equal(T.build_boundary_mask(64, 32, 2).numpy(), v["spatial_mask"])
    assert T.weighted_mse_loss(pred, target).item() == pytest.approx(float(v["loss_plain"]), rel=1e-6)
    assert T.weighted_mse_loss(pred, target, lat).item() == pytest.approx(float(v["loss_lat"]), rel=1e-6)
    assert T.weighted_mse_loss(pred, target, lat, chan).item() == pytest.approx(float(v["loss_lat_chan"]), rel=1e-6)
    assert T.weighted_mse_loss(pred, target, lat, chan, sm).item() == pytest.approx(float(v["loss_all"]), rel=1e-6)
    # pole weights are ~-7e-8, max 1.6208, mean 1
    w = v["lat_w"].reshape(-1)
    assert -1e-7 < w.min() < 0 and abs(w.max() - 1.6208) < 1e-3 and abs(w.mean() - 1.0) < 1e-6


def test_threshold_schedule_matches_reference():
    v = np.load(os.path.join(GOLDEN, "loss_vectors.npz"))
    mine = np.array([T.update_attention_threshold(int(e)) for e in v["thr_epochs"]])
    np.testing.assert_allclose(mine, v["thr_values"], rtol=0, atol=0)
    assert mine[6] == pytest.approx(0.00542, abs=1e-5) and mine[30] == pytest.approx(0.1356)


def test_config_parse_matches_reference():
    with open(os.path.join(GOLDEN, "config_parse.json")) as fh:
        ref = json.load(fh)
    for name in ("baseline", "attention", "sparse_attention", "wb2_512x256_19f_ar"):
        cfg = experiment(name)
        r = ref[name]
        assert cfg.graph.mesh_levels == r["mesh_levels"] and cfg.graph.grid2mesh_radius_query == r["radius"]
        assert cfg.pipeline.processor.gcn.layer_type.value == r["proc_type"]
        assert cfg.data.num_features_used == r["features"] and cfg.data.obs_window_used == r["obs"]
        assert cfg.pipeline.encoder.mlp.use_layer_norm is r["enc_mlp_ln"] is True
    # the reference's "True"/"False" strings must coerce (experiments/baseline/config.json)
    from graphcast_lite_amd.config import MLPBlock

    assert MLPBlock(output_dim=4, use_layer_norm="True").use_layer_norm is True
    assert MLPBlock(output_dim=4, use_layer_norm="False").use_layer_norm is False


# ---- micro-cases, SURVEY.md Appendix A.7 -------------------------------------------------------
def test_gcn_path_graph():
    x = torch.randn(3, 4)
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    y = P.gcn_conv(x, ei, torch.eye(4), torch.zeros(4))
    torch.testing.assert_close(y[1], x[1] / 3 + (x[0] + x[2]) / math.sqrt(6))
    torch.testing.assert_close(y[0], x[0] / 2 + x[1] / math.sqrt(6))


def test_gcn_single_directed_edge():
    x = torch.randn(2, 3)
    y = P.gcn_conv(x, torch.tensor([[0], [1]]), torch.eye(3), torch.zeros(3))
    torch.testing.assert_close(y[1], x[0] / math.sqrt(2) + x[1] / 2)
    torch.testing.assert_close(y[0], x[0])


def test_gcn_bias_after_aggregation_and_existing_self_loop():
    x = torch.randn(2, 3)
    b = torch.tensor([1.0, 2.0, 3.0])
    # an explicit self-loop must not be double counted
    y = P.gcn_conv(x, torch.tensor([[0, 0], [1, 0]]), torch.eye(3), b)
    torch.testing.assert_close(y[0], x[0] + b)
    torch.testing.assert_close(y[1], x[0] / math.sqrt(2) + x[1] / 2 + b)


def test_gat_uniform_attention_and_edge_order():
    x = torch.randn(4, 5)
    ei = torch.tensor([[0, 1, 2, 2], [3, 3, 3, 2]])  # includes one self-loop that is re-appended last
    W = torch.randn(6, 5)
    z = torch.zeros(1, 1, 6)
    y, ei2, alpha = P.gat_conv(x, ei, W, z, z, torch.zeros(6), heads=1)
    assert ei2.tolist() == [[0, 1, 2, 0, 1, 2, 3], [3, 3, 3, 0, 1, 2, 3]]
    torch.testing.assert_close(alpha.squeeze(), torch.tensor([.25, .25, .25, 1, 1, 1, .25]))
    h = x @ W.t()
    torch.testing.assert_close(y[3], h.mean(0))


def test_gat_heads_mean_and_icosahedron_known_answer():
    g = build_graphs(experiment("baseline", mesh_levels=[0]))
    x = torch.randn(12, 8)
    y, ei2, alpha = P.gat_conv(x, g["proc"], torch.randn(12, 8), torch.randn(1, 2, 6), torch.randn(1, 2, 6),
                               torch.zeros(6), heads=2)
    assert tuple(ei2.shape) == (2, 72)  # notebooks/src/main.ipynb:196: [2,60] -> [2,72]
    assert tuple(alpha.shape) == (72, 2) and tuple(y.shape) == (12, 6)
    s = torch.zeros(12, 2).index_add_(0, ei2[1], alpha)
    torch.testing.assert_close(s, torch.ones(12, 2))


def test_simple_conv_mean_and_isolated_node():
    x = torch.randn(3, 2)
    y = P.simple_conv_mean(x, torch.tensor([[0, 1], [2, 2]]))
    torch.testing.assert_close(y[2], (x[0] + x[1]) / 2)
    assert y[0].abs().sum() == 0 and y[1].abs().sum() == 0


def test_layer_norm_modes():
    x = torch.randn(5, 8)
    w, b = torch.rand(8) + 0.5, torch.randn(8)
    yn = P.pyg_layer_norm(x, w, b, "node")
    ref = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5) * w + b
    torch.testing.assert_close(yn, ref)
    yg = P.pyg_layer_norm(x, w, b, "graph")
    torch.testing.assert_close(yg, (x - x.mean()) / (x.std(unbiased=False) + 1e-5) * w + b)
    const = P.pyg_layer_norm(torch.full((3, 8), 2.5), w, b, "graph")
    torch.testing.assert_close(const, b.expand(3, 8))  # 0/(0+eps)


def test_prune_threshold_zero_keeps_everything():
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    a = torch.tensor([0.0, 0.3, 1.0])
    e2, a2 = P.sparse_gat_prune(ei, a, 0.0)
    assert e2.shape[1] == 3 and a2.shape[0] == 3
    e3, _ = P.sparse_gat_prune(ei, a, 0.3)
    assert e3.tolist() == [[1, 2], [2, 0]]


def test_batched_ops_equal_per_sample():
    g = build_graphs(experiment("baseline", mesh_levels=[0]))
    x = torch.randn(3, 12, 8)
    W, b = torch.randn(5, 8), torch.randn(5)
    yb = P.gcn_conv(x, g["proc"], W, b)
    for i in range(3):
        torch.testing.assert_close(yb[i], P.gcn_conv(x[i], g["proc"], W, b))
    ab = P.gat_conv(x, g["proc"], torch.randn(6, 8), torch.randn(1, 1, 6), torch.randn(1, 1, 6), None, 1)[0]
    assert tuple(ab.shape) == (3, 12, 6)


# ---- known answers recorded by the reference authors -------------------------------------------
@pytest.mark.parametrize("name,count", [("baseline", 53784), ("attention", 54168), ("sparse_attention", 20625),
                                        ("wb2_512x256_19f_ar", 209882)])
def test_parameter_counts(name, count):
    cfg = experiment(name, mesh_levels=[0])
    m = oracle_model(cfg, build_graphs(cfg))
    assert sum(p.numel() for p in m.parameters()) == count  # README_RU.MD:239 "~210K" for the last


def test_sparse_gat_layer_has_4288_params():
    from oracle.model import OSparseGATConv

    assert sum(p.numel() for p in OSparseGATConv(64, 64, heads=1).parameters()) == 4288  # main.ipynb:178-208


def test_state_dict_keys_follow_reference_layout():
    cfg = experiment("baseline", mesh_levels=[0])
    keys = set(oracle_model(cfg, build_graphs(cfg)).state_dict().keys())
    for k in ("encoder.mlp.MLP.0.weight", "encoder.mlp.MLP.1.weight", "encoder.mlp.MLP.5.bias",
              "encoder.graph_layer.activation.weight", "encoder.graph_layer.layers.0.lin.weight",
              "encoder.graph_layer.layers.0.bias", "encoder.graph_layer.layers.1.weight",
              "processor.graph_layer.layers.5.weight", "decoder.graph_layer.layers.4.lin.weight",
              "_processing_edge_features"):
        assert k in keys, k


def test_oracle_gradcheck_fp64():
    torch.manual_seed(0)
    g = build_graphs(experiment("baseline", mesh_levels=[0]))
    ei = g["proc"]
    x = torch.randn(12, 4, dtype=torch.float64, requires_grad=True)
    W = torch.randn(3, 4, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x, W, b: P.gcn_conv(x, ei, W, b), (x, W, b))
    As = torch.randn(1, 1, 3, dtype=torch.float64, requires_grad=True)
    Ad = torch.randn(1, 1, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x, W, As, Ad: P.gat_conv(x, ei, W, As, Ad, None, 1)[0], (x, W, As, Ad))


def test_oracle_train_step_runs_all_configs():
    for name in ("baseline", "attention", "sparse_attention"):
        cfg = experiment(name, mesh_levels=[1, 2])
        g = build_graphs(cfg)
        m = oracle_model(cfg, g)
        F = cfg.data.num_features_used
        X = torch.randn(2, g["G"], 2 * F)
        y = X[..., F:] + 0.1 * torch.randn(2, g["G"], F)
        loss = T.train_step_loss(m, X, y, lat_weights=T.get_lat_weights(32, 64), batch_num=0, threshold=0.05)
        loss.backward()
        assert torch.isfinite(loss)
        assert all(p.grad is not None for n, p in m.named_parameters() if "activation" not in n or name != "sparse_attention")


def test_ar_rollout_glue_semantics():
    """oracle.train_step.ar_rollout against a hand-written model: window shift, residual, static and
    forcing overwrites of scripts/predict.py:499-538."""
    from oracle import train_step as T

    class Toy(torch.nn.Module):
        obs_window = 2

        def forward(self, X, attention_threshold=0.0):
            B, G, F2 = X.shape
            C = F2 // 2
            return X[..., :C] - X[..., C:]  # delta = older - newer

    X = torch.tensor([[[1.0, 10.0, 100.0, 2.0, 20.0, 200.0]]])  # B=1,G=1,obs=2,C=3
    y = torch.tensor([[[0.0, 0.0, 7.0]]])  # one known future step
    out = T.ar_rollout(Toy(), X, 2, y=y, static_channels=[1], forcing_channels=[2])
    # step 0: x_last=[2,20,200], delta=[-1,-10,-100] -> [1,10,100]; static ch1 <- 20; forcing ch2 <- 7
    # step 1: window=[[2,20,200],[1,20,7]], delta=[1,0,193] -> [2,20,200]; static ch1 <- 20; no y left
    assert out.tolist() == [[[1.0, 20.0, 7.0, 2.0, 20.0, 200.0]]]
    out = T.ar_rollout(Toy(), X, 1, use_residual=False)
    assert out.tolist() == [[[-1.0, -10.0, -100.0]]]


def test_interaction_net_layer_against_loops():
    """oracle.model.OInteractionNetLayer (restating src/models.py:206-236) against an independent
    per-edge / per-node loop evaluation in float64 on a tiny graph with an isolated receiver."""
    from oracle import model as OM

    torch.manual_seed(3)
    n, D = 5, 4
    ei = torch.tensor([[0, 1, 2, 2, 4, 0], [1, 0, 1, 4, 2, 4]])  # node 3 receives nothing
    layer = OM.OInteractionNetLayer(D, D, D, activation="swish", use_layer_norm=True).double()
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.1 * torch.randn_like(p))
    x, e = torch.randn(n, D, dtype=torch.float64), torch.randn(ei.shape[1], D, dtype=torch.float64)
    new_x, new_e = layer(x, ei, e)

    silu = lambda v: v / (1 + torch.exp(-v))
    W1, b1, W2, b2 = (layer.edge_mlp[0].weight, layer.edge_mlp[0].bias, layer.edge_mlp[2].weight, layer.edge_mlp[2].bias)
    V1, c1, V2, c2 = (layer.node_mlp[0].weight, layer.node_mlp[0].bias, layer.node_mlp[2].weight, layer.node_mlp[2].bias)
    upd = []
    for k in range(ei.shape[1]):
        s, r = int(ei[0, k]), int(ei[1, k])
        inp = torch.cat([x[s], x[r], e[k]])
        upd.append(W2 @ silu(W1 @ inp + b1) + b2)
    upd = torch.stack(upd)
    want_x = []
    for i in range(n):
        inc = [upd[k] for k in range(ei.shape[1]) if int(ei[1, k]) == i]
        agg = torch.stack(inc).mean(0) if inc else torch.zeros(D, dtype=torch.float64)
        xi = x[i] + V2 @ silu(V1 @ torch.cat([x[i], agg]) + c1) + c2
        xi = (xi - xi.mean()) / torch.sqrt(((xi - xi.mean()) ** 2).mean() + 1e-5)   # node mode: eps inside the sqrt
        want_x.append(xi * layer.node_norm.weight + layer.node_norm.bias)
    pre_e = e + upd
    std = torch.sqrt(((pre_e - pre_e.mean()) ** 2).mean())
    want_e = (pre_e - pre_e.mean()) / (std + 1e-5) * layer.edge_norm.weight + layer.edge_norm.bias  # graph mode
    assert torch.allclose(new_x, torch.stack(want_x), atol=1e-12)
    assert torch.allclose(new_e, want_e, atol=1e-12)


def test_interaction_net_processor_batched_equals_per_sample():
    from oracle import model as OM

    torch.manual_seed(5)
    n, D, E = 6, 8, 14
    ei = torch.randint(0, n, (2, E))
    proc = OM.OInteractionNetProcessor(D, 4, D, D, num_steps=3)
    raw, x = torch.randn(E, 4), torch.randn(2, n, D)
    yb = proc(x, ei, raw)
    for b in range(2):
        assert torch.allclose(yb[b], proc(x[b], ei, raw), atol=1e-6)


def test_window_sample_hand_case_and_split_indices():
    """oracle.data.window_sample (src/data/dataloader_chunked.py:179-223) on a 2x3 grid by hand, and the
    host-side sample index / split logic of the product (:132-172)."""
    from oracle.data import window_sample
    from graphcast_lite_amd.data import sample_indices

    T, n_lon, n_lat, Ct, C = 4, 2, 3, 3, 2
    chunk = np.arange(T * n_lon * n_lat * Ct, dtype=np.float16).reshape(T, n_lon, n_lat, Ct)
    mean, std = np.array([1.0, 2.0], np.float32), np.array([2.0, 4.0], np.float32)
    X, Y = window_sample(chunk, 1, 2, 1, C, mean, std, flat=False)
    assert X.shape == (6, 4) and Y.shape == (6, 2) and X.dtype == np.float32
    for lat in range(n_lat):
        for lon in range(n_lon):
            g = lat * n_lon + lon  # lat-major, lon fastest: the node order of np.meshgrid(lons, lats)
            for o in range(2):
                for c in range(C):
                    assert X[g, o * C + c] == (np.float32(chunk[1 + o, lon, lat, c]) - mean[c]) / std[c]
            assert Y[g, 1] == (np.float32(chunk[3, lon, lat, 1]) - mean[1]) / std[1]
    Xf, Yf = window_sample(chunk.reshape(T, 6, Ct), 0, 1, 2, C, mean, std, flat=True)
    assert Xf.shape == (6, 2) and Yf.shape == (6, 4) and Yf[5, 3] == (np.float32(chunk.reshape(T, 6, Ct)[2, 5, 1]) - 2) / 4

    # two chunks of 10 and 4 frames, window 3: 8 + 2 samples, never across the boundary
    allidx = sample_indices([10, 4], 2, 1, "all", 0.2)
    assert allidx == [(0, t) for t in range(8)] + [(1, 0), (1, 1)]
    assert sample_indices([10, 4], 2, 1, "train", 0.2) == allidx[:8]
    assert sample_indices([10, 4], 2, 1, "test", 0.2) == allidx[8:]
    assert sample_indices([10, 4], 2, 1, "val", 0.2) == allidx[8:9]
    assert sample_indices([10, 4], 2, 1, "test_only", 0.2) == allidx[9:]
    assert sample_indices([2], 2, 1, "all", 0.2) == []
    with pytest.raises(ValueError):
        sample_indices([10], 2, 1, "nope", 0.2)


def test_interaction_net_v2_parameter_count():
    """experiments/wb2_512x256_19f_ar_v2: the module tree built from the config has the sizes that
    src/models.py:166-285 implies - per step (768*256+256) + (256*256+256) edge MLP, (512*256+256) +
    (256*256+256) node MLP, 4*256 norms; edge encoder 4*256+256 - i.e. 5 530 880 in the processor and
    6 022 551 in all, the "~5.9M" (28x the ~210K of v1) that README_RU.MD:142,242 quotes."""
    from graphcast_lite_amd.models import Model
    from oracle import model as OM

    cfg = experiment("wb2_512x256_19f_ar_v2")
    count = lambda m: sum(p.numel() for p in m.parameters())
    per_step = (768 * 256 + 256) + (256 * 256 + 256) + (512 * 256 + 256) + (256 * 256 + 256) + 4 * 256
    for mk in (Model, OM.Model):
        enc = mk(cfg.pipeline.encoder, 2 * 19 + 6)
        proc = mk(cfg.pipeline.processor, enc.output_dim)
        dec = mk(cfg.pipeline.decoder, proc.output_dim)
        assert count(proc) == 12 * per_step + 4 * 256 + 256 == 5_530_880
        total = count(enc) + count(proc) + count(dec)
        assert total == 6_022_551 and abs(total / 209_882 - 28) < 1.0
