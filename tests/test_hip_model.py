"""Whole-path parity on the GPU: the product module tree (HIP kernels through the C ABI) against
the CPU oracle with identical weights, graphs and inputs - outputs, loss and every parameter
gradient within 1e-5 relative fp32 (north_star tolerance), plus optimiser-step equivalence."""
import numpy as np
import pytest
import torch

from conftest import experiment
from oracle import model as omodel
from oracle import train_step as T
from parity import check_grads, grads_of, oracle_fp64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    """max(Frobenius relative error, element-wise max|diff| / max|ref|): every `rel(..) < tol`
    below therefore bounds BOTH the norm-wise and the worst single element (one bad element in 10^6 fails)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    fro = ((a - b).norm() / (b.norm() + 1e-30)).item()
    mx = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item() if b.numel() else 0.0
    return max(fro, mx)


def arbitrated_grad_check(m, o, loss_fn64, tag):
    """fp64 arbitration (tests/parity.py): `loss_fn64(o64)` must build the same loss on the float64 copy of the oracle."""
    o64 = oracle_fp64(o)
    loss_fn64(o64).backward()
    return check_grads(grads_of(m), grads_of(o), grads_of(o64), tag=tag)


def make_pair(name, levels, nlat=32, nlon=64, seed=42, tweak=None):
    from graphcast_lite_amd.models import WeatherPrediction

    cfg = experiment(name, mesh_levels=levels)
    if tweak is not None:
        tweak(cfg)  # e.g. fewer message-passing steps, so a float64 oracle pass fits in host memory at full graph size
    torch.manual_seed(seed)
    lats = np.linspace(-90, 90, nlat, endpoint=True)
    lons = np.linspace(0, 360, nlon, endpoint=False)
    m = WeatherPrediction((lats, lons), cfg.graph, cfg.pipeline, cfg.data, torch.device(DEV))
    # perturb the parameters that PyG initialises to 0/1 so their gradients paths are exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if p.dim() == 1 and p.numel() > 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g).to(DEV))
    o = omodel.WeatherPrediction(
        cfg.pipeline, cfg.data, num_grid_nodes=m._num_grid_nodes, num_mesh_nodes=m._num_mesh_nodes,
        encoding_graph=m.encoding_graph.cpu(), processing_graph=m.processing_graph.cpu(),
        decoding_graph=m.decoding_graph.cpu(), init_grid_features=m.init_grid_features.cpu(),
        init_mesh_features=m.init_mesh_features.cpu(), processing_edge_features=m._processing_edge_features.cpu(),
        product_graph=m.product_graph.cpu() if m.use_product_graph else None)
    missing = o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return cfg, m, o


def data(cfg, G, B, seed=1234):
    g = torch.Generator().manual_seed(seed)
    F, obs = cfg.data.num_features_used, cfg.data.obs_window_used
    X = torch.randn(B, G, obs * F, generator=g)
    y = X[..., (obs - 1) * F:] + 0.1 * torch.randn(B, G, F, generator=g)
    return X, y


@pytest.mark.parametrize("name,levels,B", [("baseline", [1, 2], 2), ("baseline", [3, 5], 3), ("baseline", [3, 5], 64),
                                           ("attention", [1, 2], 2),
                                           ("attention", [3, 5], 2), ("attention", [3, 5], 64), ("attention_h4", [1, 2], 2),
                                           ("sparse_attention", [1, 2], 2), ("wb2_512x256_19f_ar", [1, 2], 2),
                                           ("region_krsk_cds_19f", [1, 2], 2), ("region_krsk_cds_19f", [2, 3], 1),
                                           ("wb2_512x256_19f_ar_v2", [1, 2], 1),
                                           ("wb2_64x32_15f", [1, 2], 2), ("wb2_64x32_15f", [4, 6], 1),
                                           ("wb2_64x32_15f_gat", [1, 2], 2), ("wb2_64x32_15f_gat", [3, 5], 9),
                                           ("demo_low", [3], 3), ("product_graph", [1, 2], 2),
                                           ("product_graph", [1, 2], 1)])
def test_forward_backward_parity(name, levels, B):
    from graphcast_lite_amd.train import batch_loss, get_lat_weights

    cfg, m, o = make_pair(name, levels)
    X, y = data(cfg, m._num_grid_nodes, B)
    lw = T.get_lat_weights(32, 64)
    out_o = o(X)
    out_h = m(X.to(DEV))
    assert out_h.shape == out_o.shape
    assert rel(out_h, out_o) < 1e-5, f"forward differs: {rel(out_h, out_o):.3e}"
    if B >= 32:  # BASELINE.json configs[1] at its full batch: the batch equals its samples (first / middle / last)
        for i in (0, B // 2 - 1, B - 1):
            assert rel(out_h[i], o(X[i:i + 1])) < 1e-5, i
            assert rel(out_h[i], m(X[i:i + 1].to(DEV))) < 1e-6, i

    loss_o = T.train_step_loss(o, X, y, lat_weights=lw)
    loss_o.backward()
    loss_h = batch_loss(m, X.to(DEV), y.to(DEV), lat_weights=get_lat_weights(32, 64, DEV))
    loss_h.backward()
    assert rel(loss_h, loss_o) < 1e-5
    arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double(), lat_weights=lw.double()),
                          f"{name}{levels}B{B}")


@pytest.mark.parametrize("levels,B", [([1, 2], 2), ([3, 5], 2)])
def test_general_path_matches_compact_path(levels, B):
    """The compact pipeline (batch-invariant encoder rows computed once, dead decoder rows dropped)
    and the row-for-row general pipeline give the same outputs and gradients."""
    from graphcast_lite_amd.train import batch_loss

    cfg, m, o = make_pair("baseline", levels)
    X, y = data(cfg, m._num_grid_nodes, B)
    assert m._compact_eligible()
    out_c = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    gc = {n_: p.grad.clone() for n_, p in m.named_parameters()}
    if levels == [3, 5]:
        assert m._compact.Mi == 8302 and m._compact.U == 4804  # counts from the reference-built graphs
    m.zero_grad()
    m.compact = False
    assert not m._compact_eligible()
    out_g = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    assert rel(out_c, out_g) < 2e-6
    assert rel(out_c, o(X)) < 1e-5
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gc.values())))
    for n_, p in m.named_parameters():
        d = float((p.grad.double() - gc[n_].double()).norm())
        assert d <= 2e-5 * float(gc[n_].double().norm()) + 1e-6 * gn, n_


@pytest.mark.parametrize("levels,B", [([1, 2], 2), ([2, 4], 1), ([2, 4], 5), ([3, 5], 3), ([3, 5], 64)])
def test_folded_invariant_rows_equal_separate_pass(levels, B):
    """Compact pipeline: the batch-invariant mesh rows ride through the encoder's launches as r = ceil(Mi / B)
    isolated nodes per sample (models.py::_fold_setup, functional.py::MeshLatFn) - outputs are bit-identical to the
    separate B = 1 pass over those rows (every row sees the same arithmetic), gradients agree to summation order."""
    from graphcast_lite_amd.train import batch_loss

    cfg, m, o = make_pair("baseline", levels)
    X, y = data(cfg, m._num_grid_nodes, B)
    assert m._compact_eligible() and m._fold_invariant_rows
    out_f = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    gf = {n_: p.grad.clone() for n_, p in m.named_parameters()}
    if m._compact.Mi > 0:  # (the 162-node mesh has no batch-invariant row: every mesh node has a grid in-edge)
        f = m._compact.fold[B]
        assert f.r == -(-m._compact.Mi // B) and f.ne == m._num_grid_nodes + m._compact.Md + f.r
    else:  # the same path with r = 0 folded rows per sample (one shared gradient landing buffer, no folded tail)
        assert levels == [1, 2] and m._compact.fold[B].r == 0
    m.zero_grad()
    m._fold_invariant_rows = False
    out_s = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    assert torch.equal(out_f, out_s)
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gf.values())))
    for n_, p in m.named_parameters():
        d = float((p.grad.double() - gf[n_].double()).norm())
        assert d <= 1e-5 * float(gf[n_].double().norm()) + 1e-7 * gn, n_


@pytest.mark.parametrize("name,levels,B", [("baseline", [3, 5], 9), ("attention", [3, 5], 9), ("baseline", [2, 4], 3)])
def test_latents_through_row_table_equal_materialised(name, levels, B):
    """The first processor layer reads the mesh latents from the encoder output through a row table and runs its dense
    backward on the compact rows (functional.LatSource; GCNConv and one-head GATConv processors).  Against the same
    model with the latents gathered into [B, M, D] first (GCL_NO_LAT_TABLE): the prediction is bit-identical, the
    gradients agree to summation order."""
    from graphcast_lite_amd.train import batch_loss

    cfg, m, o = make_pair(name, levels)
    X, y = data(cfg, m._num_grid_nodes, B)
    assert m._compact_eligible() and m._fold_invariant_rows and m._lat_through_table
    out_t = m(X.to(DEV))
    c = m._compact
    f = c.fold[B]
    enc_probe = torch.empty(B, f.ne, cfg.pipeline.encoder.gcn.output_dim, device=DEV)
    assert m._lat_source(c, f, enc_probe, None) is not None, "the row-table path did not engage on this config"
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    gt = {n_: p.grad.clone() for n_, p in m.named_parameters()}
    m.zero_grad()
    m._lat_through_table = False
    out_g = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    assert torch.equal(out_t, out_g)
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gt.values())))
    for n_, p in m.named_parameters():
        d = float((p.grad.double() - gt[n_].double()).norm())
        assert d <= 1e-5 * float(gt[n_].double().norm()) + 1e-7 * gn, n_


@pytest.mark.parametrize("name", ["baseline", "attention"])
def test_processor_layernorm_into_decoder_input_equals_gather(name):
    """forward(): nobody sees the processor's output, so its final LayerNorm writes only the rows the decoder reads,
    straight into the decoder's input (gcl_layernorm_fwd_map, functional.GradLanding.dec_buf) and the gather copies
    just the grid rows.  Against the same model with the dense LayerNorm + gather: prediction bit-identical, gradients
    identical too (the backward is the same code on the same saved tensors)."""
    from graphcast_lite_amd.train import batch_loss

    cfg, m, o = make_pair(name, [3, 5])
    X, y = data(cfg, m._num_grid_nodes, 5)
    assert m._ln_into_decoder_input and m._grad_landing
    out_m = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    gm = {n_: p.grad.clone() for n_, p in m.named_parameters()}
    m.zero_grad()
    m._ln_into_decoder_input = False
    out_g = m(X.to(DEV))
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    assert torch.equal(out_m, out_g)
    for n_, p in m.named_parameters():
        assert torch.equal(p.grad, gm[n_]), n_
    # a caller that DOES look at the processor's output gets the real rows, whatever the landing flag says
    m._ln_into_decoder_input = True
    _, _, processed = m.forward_with_latents(X.to(DEV), _landing=True)
    assert float(processed.detach().abs().max()) > 0


def test_graph_mode_layernorm_model():
    """A pipeline whose MLP and processor use layer_norm_mode="graph" (SURVEY.md §8a row 9)."""
    from graphcast_lite_amd.models import WeatherPrediction
    from graphcast_lite_amd.train import batch_loss

    cfg = experiment("baseline", mesh_levels=[1, 2])
    cfg.pipeline.encoder.mlp.layer_norm_mode = "graph"
    cfg.pipeline.processor.gcn.layer_norm_mode = "graph"
    torch.manual_seed(7)
    lats, lons = np.linspace(-90, 90, 32), np.linspace(0, 360, 64, endpoint=False)
    m = WeatherPrediction((lats, lons), cfg.graph, cfg.pipeline, cfg.data, torch.device(DEV))
    o = omodel.WeatherPrediction(
        cfg.pipeline, cfg.data, num_grid_nodes=m._num_grid_nodes, num_mesh_nodes=m._num_mesh_nodes,
        encoding_graph=m.encoding_graph.cpu(), processing_graph=m.processing_graph.cpu(),
        decoding_graph=m.decoding_graph.cpu(), init_grid_features=m.init_grid_features.cpu(),
        init_mesh_features=m.init_mesh_features.cpu(), processing_edge_features=m._processing_edge_features.cpu())
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    m.compact = False  # graph-mode statistics couple all rows of a sample: no row may be dropped or shared
    X, y = data(cfg, m._num_grid_nodes, 2)
    assert rel(m(X.to(DEV)), o(X)) < 1e-5
    T.train_step_loss(o, X, y).backward()
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double()), "graph-mode LN")


def test_batch_one_follows_reference_squeeze():
    cfg, m, o = make_pair("baseline", [1, 2])
    X, _ = data(cfg, m._num_grid_nodes, 1)
    out = m(X=X.to(DEV), attention_threshold=0.0, epoch=3, batch_num=7)  # extra kwargs tolerated (src/train.py:197)
    assert tuple(out.shape) == (m._num_grid_nodes, 33)
    assert rel(out, o(X)) < 1e-5
    out2, grid_lat, mesh_lat = m.forward_with_latents(X.to(DEV))
    o2, og, om = o.forward_with_latents(X)
    assert rel(grid_lat, og) < 1e-5 and rel(mesh_lat, om) < 1e-5 and rel(out2, o2) < 1e-5


def test_batched_equals_per_sample():
    cfg, m, _ = make_pair("baseline", [1, 2])
    X, _ = data(cfg, m._num_grid_nodes, 9)  # 9 >= 8 exercises the XCD-aware block mapping
    Xd = X.to(DEV)
    with torch.no_grad():
        full = m(Xd)
        for i in (0, 4, 8):
            assert rel(full[i], m(Xd[i:i + 1])) < 1e-6


def test_sparse_gat_prune_flow():
    """`batch_num == 0` prunes the processing graph from sample 0's attention (src/models.py:138-149,846)."""
    cfg, m, o = make_pair("sparse_attention", [1, 2])
    X, _ = data(cfg, m._num_grid_nodes, 2)
    e0 = m.processing_graph.shape[1]
    with torch.no_grad():
        out_h = m(X=X.to(DEV), attention_threshold=0.12, batch_num=0)
        out_o = o(X=X, attention_threshold=0.12, batch_num=0)
    assert rel(out_h, out_o) < 1e-5
    assert torch.equal(m.processing_graph.cpu(), o.processing_graph)
    assert m.processing_graph.shape[1] < e0 + m._num_mesh_nodes
    with torch.no_grad():  # next forward runs on the pruned graph
        assert rel(m(X=X.to(DEV), attention_threshold=0.12, batch_num=1), o(X=X, attention_threshold=0.12, batch_num=1)) < 1e-5
    g1 = m.processing_graph
    with torch.no_grad():
        m(X=X.to(DEV), attention_threshold=0.12, batch_num=2)
    assert m.processing_graph is g1  # stable identity => cached CSR, no rebuild per step


def test_train_step_matches_torch_adam_on_oracle():
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    cfg, m, o = make_pair("baseline", [1, 2])
    X, y = data(cfg, m._num_grid_nodes, 4)
    lw = T.get_lat_weights(32, 64)
    opt = torch.optim.Adam(o.parameters(), lr=1e-3)
    step = TrainStep(m, lr=1e-3, lat_weights=get_lat_weights(32, 64, DEV))
    for _ in range(3):
        opt.zero_grad()
        lo = T.train_step_loss(o, X, y, lat_weights=lw)
        lo.backward()
        opt.step()
        lh = step(X.to(DEV), y.to(DEV))
        assert rel(lh, lo) < 1e-4
    od = dict(o.named_parameters())
    for n_, p in m.named_parameters():
        assert rel(p, od[n_]) < 1e-4, n_


def test_train_step_with_frozen_processor_and_rollout(monkeypatch):
    """`freeze_processor_epochs` (src/main.py:197-203) under TrainStep's deferred final passes and a 3-step
    autoregressive rollout: the gradient slots of frozen parameters are temporaries, which the queued passes write
    AFTER the autograd Function that made them has returned - they must be kept alive until the flush.  The
    deferred step must follow the step with immediate reductions parameter for parameter, frozen ones untouched."""
    from graphcast_lite_amd import train as TR

    cfg, m1, _ = make_pair("baseline", [1, 2])
    _, m2, _ = make_pair("baseline", [1, 2])
    for m in (m1, m2):
        for p in m.processor.parameters():
            p.requires_grad = False
    G, F = m1._num_grid_nodes, cfg.data.num_features_used
    X, _ = data(cfg, G, 3)
    g = torch.Generator().manual_seed(9)
    y = torch.randn(3, G, 3 * F, generator=g)
    lw = TR.get_lat_weights(32, 64, DEV)
    before = {n_: p.detach().clone() for n_, p in m1.processor.named_parameters()}
    s1 = TR.TrainStep(m1, lr=1e-3, lat_weights=lw, ar_steps=3, use_graph=False)
    monkeypatch.setattr(TR, "_DEFER_REDUCTIONS", False)
    s2 = TR.TrainStep(m2, lr=1e-3, lat_weights=lw, ar_steps=3, use_graph=False)
    for i in range(3):
        monkeypatch.setattr(TR, "_DEFER_REDUCTIONS", True)
        l1 = s1(X.to(DEV) * (1 + 0.1 * i), y.to(DEV))
        junk = [torch.full((64, 64), float(i), device=DEV) for _ in range(8)]  # would reuse a freed gradient slot
        monkeypatch.setattr(TR, "_DEFER_REDUCTIONS", False)
        l2 = s2(X.to(DEV) * (1 + 0.1 * i), y.to(DEV))
        assert rel(l1, l2) < 1e-6
        assert all(bool((j == float(i)).all()) for j in junk), "a queued final pass wrote into a tensor that is not its own"
    p2 = dict(m2.named_parameters())
    for n_, p in m1.named_parameters():
        assert rel(p, p2[n_]) < 1e-6, n_
    for n_, p in m1.processor.named_parameters():
        assert torch.equal(p, before[n_]), f"frozen parameter {n_} moved"


@pytest.mark.parametrize("name", ["baseline", "attention"])
def test_second_consumer_of_the_processor_output_keeps_its_gradient(name):
    """`WeatherPrediction.forward` lets the decoder-input gather hand the processor's output gradient over as a
    stride-0 token + row map (functional.GradLanding).  If anything else ALSO consumes the processor's output, autograd
    sums its gradient with the token: the stack's backward must notice that the incoming gradient is no longer the
    token and add the gather's part to it instead of dropping the other consumer's share."""
    cfg, m1, _ = make_pair(name, [1, 2])
    _, m2, _ = make_pair(name, [1, 2])
    X, _ = data(cfg, m1._num_grid_nodes, 3)
    m2._grad_landing = False  # plain autograd accumulation: the reference behaviour
    grads = []
    for m in (m1, m2):
        out, _, processed = m.forward_with_latents(X.to(DEV), _landing=True)
        (out.pow(2).mean() + 0.5 * processed.pow(2).mean()).backward()
        grads.append({n_: p.grad.clone() for n_, p in m.named_parameters()})
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads[1].values())))
    for n_, g in grads[1].items():
        d = float((grads[0][n_].double() - g.double()).norm())
        assert d <= 1e-5 * float(g.double().norm()) + 1e-7 * gn, n_


def test_graph_captured_step_equals_eager():
    """TrainStep replayed from a captured hipGraph follows the same trajectory as eager launches."""
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    cfg, m1, _ = make_pair("baseline", [1, 2])
    _, m2, _ = make_pair("baseline", [1, 2])
    _, m3, _ = make_pair("baseline", [1, 2])
    X, y = data(cfg, m1._num_grid_nodes, 4)
    Xd, yd = X.to(DEV), y.to(DEV)
    lw = get_lat_weights(32, 64, DEV)
    s1 = TrainStep(m1, lr=1e-3, lat_weights=lw, use_graph=True)
    s2 = TrainStep(m2, lr=1e-3, lat_weights=lw, use_graph=False)
    # the multi-GPU arrangement: forward + backward replayed, all-reduce + Adam launched after the replay
    s3 = TrainStep(m3, lr=1e-3, lat_weights=lw, use_graph=True, split_finish=True)
    for i in range(6):
        l1, l2, l3 = s1(Xd * (1 + 0.01 * i), yd), s2(Xd * (1 + 0.01 * i), yd), s3(Xd * (1 + 0.01 * i), yd)
        assert rel(l1, l2) < 1e-6 and rel(l3, l2) < 1e-6, (i, float(l1), float(l2), float(l3))
    assert s1.opt.t == s2.opt.t == s3.opt.t == 6
    for s_ in (s1, s3):  # the capture itself must have succeeded on this box (no silent eager fallback)
        assert s_.use_graph and s_._graph is not None
    p2, p3 = dict(m2.named_parameters()), dict(m3.named_parameters())
    for n_, p in m1.named_parameters():
        assert rel(p, p2[n_]) < 1e-6 and rel(p3[n_], p2[n_]) < 1e-6, n_


def test_reference_style_training_loop():
    """The drop-in boundary: `train_epoch` with a stock torch optimiser, as src/main.py:212 + src/train.py:466 do."""
    from graphcast_lite_amd.train import get_lat_weights, train_epoch

    cfg, m, o = make_pair("baseline", [1, 2])
    X, y = data(cfg, m._num_grid_nodes, 2)
    loader = [(X[0:1], y[0:1]), (X[1:2], y[1:2])]  # batch_size 1, as every reference config
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    l1 = train_epoch(m, loader, opt, None, DEV, 0.0, 0, lat_weights=get_lat_weights(32, 64, DEV))
    l2 = train_epoch(m, loader, opt, None, DEV, 0.0, 1, lat_weights=get_lat_weights(32, 64, DEV))
    assert np.isfinite(l1) and l2 < l1
    # freeze / unfreeze of the processor (src/main.py:197-203, src/train.py:445)
    for p in m.processor.parameters():
        p.requires_grad = False
    opt.zero_grad()
    from graphcast_lite_amd.train import batch_loss

    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    assert all(p.grad is None for p in m.processor.parameters())
    assert all(p.grad is not None for p in m.encoder.parameters())


def test_state_dict_round_trip_with_reference_keys():
    cfg, m, o = make_pair("attention", [1, 2])
    sd = o.state_dict()
    sd2 = {k.replace("lin.weight", "lin_src.weight") if ".layers." in k and "processor" in k else k: v for k, v in sd.items()}
    res = m.load_state_dict(sd2, strict=False)  # older-PyG alias lin_src is accepted
    assert not [k for k in res.missing_keys if "lin" in k]
    assert "processor.graph_layer.layers.1.weight" in m.state_dict()  # alias of the shared PReLU
    assert m.processor.graph_layer.layers[1] is m.processor.graph_layer.activation


@pytest.mark.parametrize("obs,C,B,static,forcing,residual,with_y", [
    (2, 5, 3, [1], [3, 4], True, True), (2, 5, 2, None, None, False, False), (1, 4, 2, [0, 2], [2], True, True),
    (3, 7, 1, [6], [0], True, False)])
def test_ar_advance_kernel(obs, C, B, static, forcing, residual, with_y):
    """gcl_ar_advance == the per-step glue of scripts/predict.py:512-535 (bit-exact: one add at most)."""
    from graphcast_lite_amd import hip
    from graphcast_lite_amd.predict import channel_kinds

    G, steps, s = 37, 3, 1
    g = torch.Generator().manual_seed(5)
    state = torch.randn(B, G, obs, C, generator=g)
    delta = torch.randn(B, G, C, generator=g)
    y = torch.randn(B, G, steps * C, generator=g) if with_y else None
    out = torch.full((B, G, steps * C), -7.0)
    want = (state[:, :, -1] + delta) if residual else delta.clone()
    for ch in static or []:
        want[:, :, ch] = state[:, :, -1, ch]
    if with_y:
        for ch in forcing or []:
            want[:, :, ch] = y[:, :, s * C + ch]
    want_state = torch.cat([state[:, :, 1:], want.unsqueeze(2)], dim=2)
    out_d = out.to(DEV)
    yd = y.to(DEV)[:, :, s * C:(s + 1) * C] if with_y else None
    new = hip.ar_advance(state.to(DEV), delta.to(DEV), yd, channel_kinds(C, static, forcing, DEV), out_d, s * C,
                         residual)
    assert torch.equal(new.cpu(), want_state)
    assert torch.equal(out_d.cpu()[:, :, s * C:(s + 1) * C], want)
    assert (out_d.cpu()[:, :, :s * C] == -7.0).all() and (out_d.cpu()[:, :, (s + 1) * C:] == -7.0).all()


@pytest.mark.parametrize("name,levels", [("baseline", [1, 2]), ("wb2_512x256_19f_ar", [1, 2])])
def test_rollout_matches_oracle(name, levels):
    """AR inference caller (scripts/predict.py:499-538) on the device vs the oracle's restatement."""
    from graphcast_lite_amd.predict import rollout

    cfg, m, o = make_pair(name, levels)
    G = m._num_grid_nodes
    F = cfg.data.num_features_used
    obs = m.obs_window
    steps = 4
    g = torch.Generator().manual_seed(77)
    X = torch.randn(2, G, obs * F, generator=g)
    y = torch.randn(2, G, 3 * F, generator=g)  # covers 3 of the 4 steps: the last keeps the model's forcing values
    static, forcing = [F - 1], [0, 2]
    m.eval(), o.eval()
    got = rollout(m, X.to(DEV), steps, y=y.to(DEV), static_channels=static, forcing_channels=forcing)
    want = T.ar_rollout(o, X, steps, y=y, static_channels=static, forcing_channels=forcing)
    assert got.shape == (2, G, steps * F)
    for s in range(steps):  # error compounds through the window; stays at fp32 round-off
        assert rel(got[..., s * F:(s + 1) * F], want[..., s * F:(s + 1) * F]) < 1e-5, s
    # carried / forced channels are exact copies
    assert torch.equal(got[..., F - 1].cpu(), X[..., obs * F - 1])
    assert torch.equal(got[..., F + 2].cpu(), y[..., F + 2])
    # batch-1, 2-D input follows the reference's [G, AR*C] output
    got1 = rollout(m, X[0].to(DEV), 2, use_residual=False)
    want1 = T.ar_rollout(o, X[:1], 2, use_residual=False)[0]
    assert got1.shape == (G, 2 * F) and rel(got1, want1) < 1e-5


def test_checkpoint_resume_interchange(tmp_path):
    """save/load_checkpoint keep the reference's dictionary (src/train.py:22-49): a checkpoint written
    here loads into the oracle (reference-keyed) model and resumes torch.optim.Adam exactly."""
    from graphcast_lite_amd.train import FileNames, load_checkpoint, save_checkpoint

    cfg, m, o = make_pair("baseline", [1, 2])
    G, F = m._num_grid_nodes, cfg.data.num_features_used
    X, y = data(cfg, G, 2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def one_step(model, optim):
        optim.zero_grad()
        loss = ((model(X=X.to(DEV)) + X.to(DEV)[..., F:] - y.to(DEV)) ** 2).mean()
        loss.backward()
        optim.step()
        return loss.item()

    one_step(m, opt)
    path = tmp_path / FileNames.CHECKPOINT
    save_checkpoint(path, m, opt, epoch=3, ar_steps=2, best_val_loss=0.5, patience_counter=1, train_losses=[1.0, 0.7],
                    val_losses=[0.9])
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"epoch", "ar_steps", "best_val_loss", "patience_counter", "train_losses", "val_losses",
                        "model_state_dict", "optimizer_state_dict"}
    o.load_state_dict(raw["model_state_dict"], strict=True)  # reference-side load
    after = one_step(m, opt)

    cfg2, m2, _ = make_pair("baseline", [1, 2], seed=7)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    st = load_checkpoint(path, m2, opt2, torch.device(DEV))
    assert st == {"start_epoch": 4, "ar_steps": 2, "best_val_loss": 0.5, "patience_counter": 1,
                  "train_losses": [1.0, 0.7], "val_losses": [0.9]}
    assert one_step(m2, opt2) == after  # same weights + same Adam moments -> identical step
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_fused_adam_state_interchanges_with_torch_adam(tmp_path):
    """FusedAdam.state_dict() is what torch.optim.Adam holds after the same steps: a checkpoint written
    from TrainStep resumes torch.optim.Adam on the oracle (the reference's load_checkpoint), and a
    torch-written checkpoint resumes the fused optimiser."""
    from graphcast_lite_amd.train import TrainStep, get_lat_weights, load_checkpoint, save_checkpoint

    cfg, m, o = make_pair("baseline", [1, 2])
    X, y = data(cfg, m._num_grid_nodes, 4)
    lw, lwd = T.get_lat_weights(32, 64), get_lat_weights(32, 64, DEV)
    opt = torch.optim.Adam(o.parameters(), lr=2e-3)
    step = TrainStep(m, lr=2e-3, lat_weights=lwd, use_graph=False)

    def oracle_step():
        opt.zero_grad()
        T.train_step_loss(o, X, y, lat_weights=lw).backward()
        opt.step()

    for _ in range(2):
        oracle_step()
        step(X.to(DEV), y.to(DEV))
    sd_f, sd_t = step.opt.state_dict(), opt.state_dict()
    assert sorted(sd_f["state"]) == sorted(sd_t["state"])
    assert sd_f["param_groups"][0]["params"] == sd_t["param_groups"][0]["params"]
    for i, st in sd_t["state"].items():
        assert float(sd_f["state"][i]["step"]) == float(st["step"]) == 2.0
        assert rel(sd_f["state"][i]["exp_avg"], st["exp_avg"]) < 1e-4
        assert rel(sd_f["state"][i]["exp_avg_sq"], st["exp_avg_sq"]) < 1e-4

    # fused -> file -> torch.optim.Adam on a fresh oracle, then one more step on both sides
    path = tmp_path / "checkpoint.pth"
    save_checkpoint(path, m, step.opt, 0, 1, 1.0, 0, [], [])
    raw = torch.load(path, map_location="cpu", weights_only=True)
    o.load_state_dict(raw["model_state_dict"], strict=True)
    opt = torch.optim.Adam(o.parameters(), lr=123.0)
    opt.load_state_dict(raw["optimizer_state_dict"])
    assert opt.param_groups[0]["lr"] == 2e-3
    oracle_step()
    step(X.to(DEV), y.to(DEV))
    od = dict(o.named_parameters())
    for n_, p in m.named_parameters():
        assert rel(p, od[n_]) < 1e-5, n_

    # torch -> file -> fused on a fresh product model
    torch.save({"epoch": 0, "ar_steps": 1, "best_val_loss": 1.0, "patience_counter": 0, "train_losses": [],
                "val_losses": [], "model_state_dict": o.state_dict(), "optimizer_state_dict": opt.state_dict()}, path)
    _, m2, _ = make_pair("baseline", [1, 2], seed=11)
    step2 = TrainStep(m2, lr=5.0, lat_weights=lwd, use_graph=False)
    load_checkpoint(path, m2, step2.opt, torch.device(DEV))
    assert step2.opt.t == 3 and step2.opt.lr == 2e-3
    oracle_step()
    step2(X.to(DEV), y.to(DEV))
    for n_, p in m2.named_parameters():
        assert rel(p, od[n_]) < 1e-5, n_


@pytest.mark.parametrize("act", ["relu", "prelu", "swish"])
@pytest.mark.parametrize("family", ["gcn", "gat", "interaction"])
def test_activation_variants(family, act):
    """`_get_activation` (src/models.py:154-163): swish / relu / prelu inside the GCN / GAT stacks and the
    InteractionNet MLPs (relu runs as a PReLU with a constant zero slope)."""
    from graphcast_lite_amd.train import batch_loss, get_lat_weights

    name = {"gcn": "baseline", "gat": "attention", "interaction": "region_krsk_cds_19f"}[family]
    import conftest
    base = conftest.experiment

    def patched(nm, mesh_levels=None):
        cfg = base(nm, mesh_levels=mesh_levels)
        for blk in (cfg.pipeline.encoder.gcn, cfg.pipeline.processor.gcn, cfg.pipeline.decoder.gcn):
            blk.activation = act
        return cfg

    global experiment
    saved, experiment = experiment, patched
    try:
        cfg, m, o = make_pair(name, [1, 2])
    finally:
        experiment = saved
    X, y = data(cfg, m._num_grid_nodes, 2)
    assert rel(m(X.to(DEV)), o(X)) < 1e-5
    lw = T.get_lat_weights(32, 64)
    T.train_step_loss(o, X, y, lat_weights=lw).backward()
    batch_loss(m, X.to(DEV), y.to(DEV), lat_weights=get_lat_weights(32, 64, DEV)).backward()
    if act == "relu":
        # ReLU's derivative jumps by the full gradient at 0: a pre-activation that rounds to the other side of 0 in
        # one of the two fp32 implementations flips a whole element, so the smooth-function bound does not apply;
        # compared directly with the fp32 oracle at 3e-3 (module-local floor as in tests/parity.py)
        og = dict(o.named_parameters())
        for n_, p in m.named_parameters():
            if og[n_].grad is None:
                continue
            mk = n_.rsplit(".", 1)[0]
            floor = 1e-6 * max(float(q.grad.double().norm()) for k, q in og.items() if q.grad is not None and k.startswith(mk.rsplit(".lin", 1)[0]))
            d = float((p.grad.double().cpu() - og[n_].grad.double()).norm())
            assert d <= 3e-3 * float(og[n_].grad.double().norm()) + floor, (n_, d)
    else:
        arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double(), lat_weights=lw.double()),
                              f"{family}-{act}")


def test_interaction_net_without_layer_norm_and_graph_capture():
    """InteractionNet with use_layer_norm=False (src/models.py:381) matches the oracle, and a TrainStep on an
    InteractionNet model replays from a captured hipGraph like eager launches."""
    from graphcast_lite_amd.train import TrainStep, batch_loss, get_lat_weights

    import conftest
    base = conftest.experiment

    def patched(nm, mesh_levels=None):
        cfg = base(nm, mesh_levels=mesh_levels)
        cfg.pipeline.processor.gcn.use_layer_norm = False
        cfg.pipeline.processor.gcn.num_message_passing_steps = 3
        return cfg

    global experiment
    saved, experiment = experiment, patched
    try:
        cfg, m, o = make_pair("region_krsk_cds_19f", [1, 2])
    finally:
        experiment = saved
    assert not hasattr(m.processor.graph_layer.layers.steps[0], "edge_norm")
    X, y = data(cfg, m._num_grid_nodes, 2)
    assert rel(m(X.to(DEV)), o(X)) < 1e-5
    lw = T.get_lat_weights(32, 64)
    T.train_step_loss(o, X, y, lat_weights=lw).backward()
    batch_loss(m, X.to(DEV), y.to(DEV), lat_weights=get_lat_weights(32, 64, DEV)).backward()
    arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double(), lat_weights=lw.double()),
                          "interaction-net without LN")

    _, m1, _ = make_pair("region_krsk_cds_19f", [1, 2])
    _, m2, _ = make_pair("region_krsk_cds_19f", [1, 2])
    lwd = get_lat_weights(32, 64, DEV)
    s1, s2 = TrainStep(m1, lr=1e-3, lat_weights=lwd, use_graph=True), TrainStep(m2, lr=1e-3, lat_weights=lwd, use_graph=False)
    Xd, yd = X.to(DEV), y.to(DEV)
    for i in range(5):
        l1, l2 = s1(Xd * (1 + 0.01 * i), yd), s2(Xd * (1 + 0.01 * i), yd)
        assert rel(l1, l2) < 1e-6, (i, float(l1), float(l2))
    assert s1.use_graph and s1._graph is not None


def test_captured_rollout_equals_eager_rollout():
    """predict.CapturedRollout (the K-step forecast replayed from one hipGraph) == predict.rollout."""
    from graphcast_lite_amd.predict import CapturedRollout, rollout

    cfg, m, _ = make_pair("baseline", [1, 2])
    G, F = m._num_grid_nodes, cfg.data.num_features_used
    m.eval()
    cap = CapturedRollout(m, 3, static_channels=[F - 1], forcing_channels=[0])
    g = torch.Generator().manual_seed(3)
    for i in range(5):  # two eager calls, the capturing call, two replays
        X = torch.randn(1, G, 2 * F, generator=g).to(DEV)
        y = torch.randn(1, G, 3 * F, generator=g).to(DEV)
        want = rollout(m, X, 3, y=y, static_channels=[F - 1], forcing_channels=[0])
        got = cap(X, y)
        assert torch.equal(got, want), i
    assert cap.enabled and cap._graph is not None  # the capture itself must have succeeded on this box
    # a new input signature starts over (eager, then a fresh capture)
    X2 = torch.randn(2, G, 2 * F, generator=g).to(DEV)
    assert torch.equal(cap(X2), rollout(m, X2, 3, static_channels=[F - 1], forcing_channels=[0]))


def test_sparse_gat_train_step_graph_replay_follows_pruning():
    """TrainStep on a SparseGAT model: ordinary steps replay a captured hipGraph, the pruning step
    (batch_num == 0) runs eagerly and forces a re-capture over the pruned edge list; the trajectory
    equals all-eager launches."""
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    cfg, m1, _ = make_pair("sparse_attention", [1, 2])
    _, m2, _ = make_pair("sparse_attention", [1, 2])
    X, y = data(cfg, m1._num_grid_nodes, 3)
    Xd, yd = X.to(DEV), y.to(DEV)
    lw = get_lat_weights(32, 64, DEV)
    s1 = TrainStep(m1, lr=1e-3, lat_weights=lw, use_graph=True)
    s2 = TrainStep(m2, lr=1e-3, lat_weights=lw, use_graph=False)
    e0 = int(m1.processing_graph.shape[1])
    schedule = [1, 2, 3, 4, 0, 1, 2, 3, 4]  # the 5th step prunes (as batch 0 of an epoch does, src/train.py:197)
    for i, bn in enumerate(schedule):
        l1 = s1(Xd, yd, threshold=0.12, epoch=0, batch_num=bn)
        l2 = s2(Xd, yd, threshold=0.12, epoch=0, batch_num=bn)
        assert rel(l1, l2) < 1e-6, (i, float(l1), float(l2))
        if i == 3:
            assert s1._graph is not None  # replaying before the prune
    assert torch.equal(m1.processing_graph, m2.processing_graph)
    assert int(m1.processing_graph.shape[1]) < e0 + m1._num_mesh_nodes  # pruned
    assert s1.use_graph and s1._graph is not None  # re-captured after the prune
    for (n_, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert rel(p, q) < 1e-6, n_


@pytest.mark.parametrize("name", ["baseline", "region_krsk_cds_19f"])
def test_autoregressive_training_step_parity(name):
    """The AR inner loop of training (src/train.py:186-231): 3 rollout steps with static and forcing channel
    overwrites, loss averaged over the steps, gradients through the whole rollout - against the oracle;
    and the same step through TrainStep (graph replay) against eager."""
    from graphcast_lite_amd.train import TrainStep, batch_loss, get_lat_weights

    cfg, m, o = make_pair(name, [1, 2])
    G, F = m._num_grid_nodes, cfg.data.num_features_used
    g = torch.Generator().manual_seed(21)
    X = torch.randn(2, G, 2 * F, generator=g)
    y = torch.randn(2, G, 3 * F, generator=g) * 0.5
    static, forcing = [F - 1], [0, 2]
    lw, lwd = T.get_lat_weights(32, 64), get_lat_weights(32, 64, DEV)
    lo = T.train_step_loss(o, X, y, lat_weights=lw, ar_steps=3, static_channels=static, forcing_channels=forcing)
    lo.backward()
    lh = batch_loss(m, X.to(DEV), y.to(DEV), lat_weights=lwd, current_ar_steps=3, static_channels=static,
                    forcing_channels=forcing)
    lh.backward()
    assert rel(lh, lo) < 1e-5
    arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double(), lat_weights=lw.double(), ar_steps=3,
                                                              static_channels=static, forcing_channels=forcing),
                          f"AR3 {name}")

    _, m1, _ = make_pair(name, [1, 2])
    _, m2, _ = make_pair(name, [1, 2])
    kw = dict(lr=1e-3, lat_weights=lwd, ar_steps=3, static_channels=static, forcing_channels=forcing)
    s1, s2 = TrainStep(m1, use_graph=True, **kw), TrainStep(m2, use_graph=False, **kw)
    for i in range(4):
        l1, l2 = s1(X.to(DEV), y.to(DEV)), s2(X.to(DEV), y.to(DEV))
        assert rel(l1, l2) < 1e-6, i
    assert s1.use_graph and s1._graph is not None


@pytest.mark.parametrize("name,levels", [("baseline", [1, 2]), ("demo_low", [3]), ("wb2_512x256_19f_ar", [1, 2])])
def test_ar_gradients_accumulate_in_flat_bucket(name, levels):
    """The decoder backward runs once per AR step inside ONE loss.backward() and every run ACCUMULATES into
    the live `.grad` slices of the flat bucket (TrainStep).  Each destination of the fused backward has its
    own accumulate flag: the bias gradient of conv k-1 (column sums of dX) must not follow the flag of the
    width-padded dW scratch of the last conv (the round-1 bug: it kept only the last-run contribution)."""
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    cfg, m, o = make_pair(name, levels)
    G, F = m._num_grid_nodes, cfg.data.num_features_used
    g = torch.Generator().manual_seed(33)
    obs = m.obs_window
    X = torch.randn(2, G, obs * F, generator=g)
    y = torch.randn(2, G, 3 * F, generator=g) * 0.5
    static, forcing = [F - 1], [0, 2]
    lw, lwd = T.get_lat_weights(32, 64), get_lat_weights(32, 64, DEV)
    lo = T.train_step_loss(o, X, y, lat_weights=lw, ar_steps=3, static_channels=static, forcing_channels=forcing)
    lo.backward()
    step = TrainStep(m, lr=1e-3, lat_weights=lwd, ar_steps=3, static_channels=static, forcing_channels=forcing,
                     use_graph=False)
    o64g = None
    for rep in range(2):  # twice: the second call starts from a zeroed, EXISTING bucket again
        lh = step._fwd_bwd(X.to(DEV), y.to(DEV))
        assert rel(lh, lo) < 1e-5
        for n_, p in m.named_parameters():
            if p.grad is not None:
                assert p.grad.data_ptr() >= step.flat.grad.data_ptr()  # the bucket slice, accumulated in place
        if o64g is None:
            o64 = oracle_fp64(o)
            T.train_step_loss(o64, X.double(), y.double(), lat_weights=lw.double(), ar_steps=3, static_channels=static,
                              forcing_channels=forcing).backward()
            o64g = grads_of(o64)
        check_grads(grads_of(m), grads_of(o), o64g, tag=f"AR3 flat bucket {name} rep {rep}", verbose=rep == 0)


@pytest.mark.parametrize("flat", [False, True])
def test_regional_model_parity(flat):
    """The regional arrangement of the region_* experiments (src/main.py:146-173): a 61x41 grid over
    lat 50-60 / lon 85-100 on a mesh pruned to the region (+15 deg), regular or flat (per-node coordinates),
    InteractionNet processor - forward and gradients against the oracle."""
    from graphcast_lite_amd.models import WeatherPrediction
    from graphcast_lite_amd.train import batch_loss

    cfg = experiment("region_krsk_cds_19f", mesh_levels=[3, 5])
    cfg.pipeline.processor.gcn.num_message_passing_steps = 2
    lats, lons = np.linspace(50, 60, 41).astype(np.float32), np.linspace(85, 100, 61).astype(np.float32)
    if flat:
        lon2, lat2 = np.meshgrid(lons, lats)
        coords = (lat2.reshape(-1), lon2.reshape(-1))
    else:
        coords = (lats, lons)
    torch.manual_seed(5)
    m = WeatherPrediction(coords, cfg.graph, cfg.pipeline, cfg.data, torch.device(DEV),
                          region_bounds=(50.0, 60.0, 85.0, 100.0), mesh_buffer=15.0, flat_grid=flat)
    assert m._num_grid_nodes == 61 * 41 and m._num_mesh_nodes == 259
    o = omodel.WeatherPrediction(
        cfg.pipeline, cfg.data, num_grid_nodes=m._num_grid_nodes, num_mesh_nodes=m._num_mesh_nodes,
        encoding_graph=m.encoding_graph.cpu(), processing_graph=m.processing_graph.cpu(),
        decoding_graph=m.decoding_graph.cpu(), init_grid_features=m.init_grid_features.cpu(),
        init_mesh_features=m.init_mesh_features.cpu(), processing_edge_features=m._processing_edge_features.cpu())
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)
    X, y = data(cfg, m._num_grid_nodes, 2)
    assert rel(m(X.to(DEV)), o(X)) < 1e-5
    T.train_step_loss(o, X, y).backward()
    batch_loss(m, X.to(DEV), y.to(DEV)).backward()
    arbitrated_grad_check(m, o, lambda o64: T.train_step_loss(o64, X.double(), y.double()), f"regional flat={flat}")


def test_evaluation_loop_matches_reference_semantics():
    """train.test (src/train.py:241-308: one-step prediction with carry-forward, weighted MSE, spatial ACC,
    raw RMSE) against the oracle's restatement, on batches with multi-step targets."""
    from graphcast_lite_amd.train import get_lat_weights, test as hip_test

    cfg, m, o = make_pair("baseline", [1, 2])
    G, F = m._num_grid_nodes, cfg.data.num_features_used
    g = torch.Generator().manual_seed(9)
    batches = [(torch.randn(2, G, 2 * F, generator=g), torch.randn(2, G, 3 * F, generator=g)) for _ in range(3)]
    static, forcing = [F - 1], [0, 5]
    chan = torch.ones(F)
    chan[F - 1] = 0.0
    got = hip_test(m, batches, None, DEV, lat_weights=get_lat_weights(32, 64, DEV), channel_mask=chan.to(DEV),
                   static_channels=static, forcing_channels=forcing)
    o.eval()
    want = T.evaluate(o, batches, lat_weights=T.get_lat_weights(32, 64), channel_mask=chan, static_channels=static,
                      forcing_channels=forcing)
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (got, want)


@pytest.mark.parametrize("name", ["baseline", "attention", "region_krsk_cds_19f"])
def test_step_is_bitwise_deterministic(name):
    """No float atomics anywhere (all reductions are fixed-order two-stage ones): the same step run twice,
    and run on a second identically seeded model, gives bit-identical outputs and gradients."""
    from graphcast_lite_amd.train import batch_loss, get_lat_weights

    cfg, m1, _ = make_pair(name, [1, 2])
    _, m2, _ = make_pair(name, [1, 2])
    X, y = data(cfg, m1._num_grid_nodes, 3)
    Xd, yd, lw = X.to(DEV), y.to(DEV), get_lat_weights(32, 64, DEV)

    def grads(m):
        for p in m.parameters():
            p.grad = None
        loss = batch_loss(m, Xd, yd, lat_weights=lw)
        loss.backward()
        return loss.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]

    l1, g1 = grads(m1)
    l1b, g1b = grads(m1)
    l2, g2 = grads(m2)
    assert torch.equal(l1, l1b) and torch.equal(l1, l2)
    for a, b, c in zip(g1, g1b, g2):
        if a is None:
            assert b is None and c is None
            continue
        assert torch.equal(a, b) and torch.equal(a, c)
