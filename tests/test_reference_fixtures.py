"""Fixtures produced by RUNNING the reference's own code (tests/golden/make_golden.py, build container only)
against (a) the CPU oracle - here, `-m "not gpu"` - and (b) the HIP path - `-m gpu`, bottom of the file.

  loader_vectors.npz      TimeseriesChunkDataset.__getitem__        src/data/dataloader_chunked.py:33-223
  mlp_vectors.npz         MLP (no LayerNorm) forward + autograd      src/models.py:54-109
  train_loop_vectors.npz  train_epoch / test / spatial_corr on a stub model   src/train.py:114-130,138-308
  assemble_vectors.npz    WeatherPrediction._preprocess_input        src/models.py:776-806

The fixtures are data only; nothing here reads /root/reference."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
from stub_model import CASES, EVAL_CASES, StubModel  # noqa: E402

DEV = "cuda:0"


def _npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def relmax(a, b):
    """(Frobenius relative error, max|diff| / max|ref|)"""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300)), float((a - b).abs().max() / (b.abs().max() + 1e-300))


# ------------------------------------------------------------------------------------------------
# (a) CPU oracle vs the reference-run fixtures
# ------------------------------------------------------------------------------------------------
def test_oracle_window_loader_matches_reference_bit_for_bit():
    from oracle.data import window_sample

    z = _npz("loader_vectors.npz")
    C = int(z["grid_C"])
    mean, std = z["grid_mean"].astype(np.float32)[:C], z["grid_std"].astype(np.float32)[:C]
    for split in ("train", "test", "val", "test_only", "all"):
        for i, t in enumerate(z[f"grid_{split}_t0"]):
            X, Y = window_sample(z["grid_series"], int(t), 2, 3, C, mean, std, flat=False)
            assert np.array_equal(X, z[f"grid_{split}_X"][i]) and np.array_equal(Y, z[f"grid_{split}_Y"][i])
    fm, fs = z["flat_mean"].astype(np.float32), z["flat_std"].astype(np.float32)
    for i, t in enumerate(z["flat_all_t0"]):
        X, Y = window_sample(z["flat_series"], int(t), 3, 1, 4, fm, fs, flat=True)
        assert np.array_equal(X, z["flat_all_X"][i]) and np.array_equal(Y, z["flat_all_Y"][i])


def test_product_sample_indices_match_reference_splits():
    """Host logic of the product loader (no GPU): the (chunk, t) list of every split, incl. windows that may
    not cross a chunk boundary."""
    from graphcast_lite_amd.data import sample_indices

    z = _npz("loader_vectors.npz")
    T = z["grid_series"].shape[0]
    for split in ("train", "test", "val", "test_only", "all"):
        got = sample_indices([T], 2, 3, split, 0.2)
        assert [t for _, t in got] == list(z[f"grid_{split}_t0"]) and len(got) == int(z[f"grid_{split}_len"])
    got = sample_indices([6, T - 6], 2, 1, "all", 0.2)
    assert [list(p) for p in got] == z["chunks_all_index"].tolist()


@pytest.mark.parametrize("tag", ["enc", "dec", "odd", "single"])
def test_oracle_mlp_matches_reference(tag):
    from graphcast_lite_amd.config import MLPBlock
    from oracle.model import MLP

    z = _npz("mlp_vectors.npz")
    m = _mlp_from_fixture(MLP, MLPBlock, z, tag)
    x = torch.tensor(z[f"{tag}_x"], requires_grad=True)
    y = m(x)
    y.backward(torch.tensor(z[f"{tag}_dy"]))
    assert max(relmax(y, z[f"{tag}_y"])) < 1e-6 and max(relmax(x.grad, z[f"{tag}_dx"])) < 1e-6
    for k, p in m.named_parameters():
        assert max(relmax(p.grad, z[f"{tag}_g_{k}"])) < 2e-6, k


def _mlp_from_fixture(cls, block_cls, z, tag, device="cpu"):
    keys = [str(k) for k in z[f"{tag}_keys"]]
    lin = [k for k in keys if k.endswith(".weight") and z[f"{tag}_w_{k}"].ndim == 2]
    dims = [z[f"{tag}_w_{k}"].shape for k in lin]
    m = cls(block_cls(mlp_hidden_dims=[d[0] for d in dims[:-1]], output_dim=dims[-1][0], use_layer_norm=False),
            input_dim=dims[0][1])
    sd = {k: torch.tensor(z[f"{tag}_w_{k}"]) for k in keys}
    assert sorted(sd) == sorted(m.state_dict().keys())  # the reference's state-dict keys
    m.load_state_dict(sd)
    return m.to(device)


def _case_kwargs(z, spec, device="cpu"):
    t = lambda a: torch.tensor(a).to(device)
    return dict(lat_weights=t(z["lat_w"]) if spec.get("lat") else None,
                channel_mask=t(z["chan_mask"]) if spec.get("chan") else None,
                spatial_mask=t(z["spatial_mask"]) if spec.get("smask") else None,
                static_channels=spec.get("static"), forcing_channels=spec.get("forcing"),
                use_residual=spec.get("residual", True))


def _batches(z, device="cpu"):
    return [(torch.tensor(x).to(device), torch.tensor(y).to(device)) for x, y in zip(z["X"], z["Y"])]


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_training_loop_matches_reference(tag):
    """oracle.train_step.train_step_loss (+ torch Adam) == the reference's train_epoch on the stub model:
    loss and gradients of the first batch, then 2 epochs x 3 batches of Adam steps."""
    from oracle import train_step as T

    z = _npz("train_loop_vectors.npz")
    spec, batches = CASES[tag], _batches(z)
    kw = _case_kwargs(z, spec)
    m = StubModel(z["obs"], z["W0"], z["b0"])
    X, y = batches[0]
    loss = T.train_step_loss(m, X, y, ar_steps=spec["ar"], batch_num=0, **kw)
    loss.backward()
    assert abs(float(loss.detach()) - float(z[f"{tag}_loss_first"])) <= 1e-6 * abs(float(z[f"{tag}_loss_first"]))
    assert max(relmax(m.W.grad, z[f"{tag}_gW_first"])) < 1e-5 and max(relmax(m.b.grad, z[f"{tag}_gb_first"])) < 1e-5
    m = StubModel(z["obs"], z["W0"], z["b0"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    for ep in range(2):
        tot = 0.0
        for i, (X, y) in enumerate(batches):
            opt.zero_grad()
            loss = T.train_step_loss(m, X, y, ar_steps=spec["ar"], epoch=ep, batch_num=i, **kw)
            loss.backward()
            opt.step()
            tot += float(loss)
        assert abs(tot / 3 - z[f"{tag}_epoch_losses"][ep]) <= 2e-6 * abs(z[f"{tag}_epoch_losses"][ep])
    assert max(relmax(m.W, z[f"{tag}_W"])) < 1e-5 and max(relmax(m.b, z[f"{tag}_b"])) < 1e-5
    assert max(relmax(m.W.grad, z[f"{tag}_gW_last"])) < 1e-4


@pytest.mark.parametrize("tag", list(EVAL_CASES) + ["eval_single_target"])
def test_oracle_evaluation_loop_matches_reference(tag):
    from oracle import train_step as T

    z = _npz("train_loop_vectors.npz")
    m = StubModel(z["obs"], z["W0"], z["b0"])
    batches = _batches(z)
    if tag == "eval_single_target":
        C = int(z["C"])
        batches = [(X, Y[..., :C].contiguous()) for X, Y in batches]
        kw = _case_kwargs(z, dict(lat=True))
    else:
        kw = _case_kwargs(z, EVAL_CASES[tag])
    got = T.evaluate(m, batches, **kw)
    for a, b in zip(got, z[tag]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (tag, got, z[tag])


def test_spatial_corr_matches_reference_oracle_and_product():
    """`spatial_corr` is plain torch arithmetic on both sides (host-side metric, src/train.py:114-130)."""
    from graphcast_lite_amd.train import spatial_corr as product_sc
    from oracle.train_step import spatial_corr as oracle_sc

    z = _npz("train_loop_vectors.npz")
    p, t = torch.tensor(z["sc_pred"]), torch.tensor(z["sc_true"])
    for fn in (oracle_sc, product_sc):
        assert abs(fn(p, t) - float(z["sc_batched"])) < 1e-7
        assert abs(fn(p[0], t[0]) - float(z["sc_sample0"])) < 1e-7
        assert abs(fn(p, t, exclude_channels=[1, 3]) - float(z["sc_excl"])) < 1e-7
        assert abs(fn(torch.ones_like(t[0]), t[0]) - float(z["sc_const"])) < 1e-7


def test_oracle_input_assembly_matches_reference():
    from types import SimpleNamespace

    from oracle.model import WeatherPrediction as OWP

    z = _npz("assemble_vectors.npz")
    for tag in ("a", "b"):
        gs, ms = torch.tensor(z[f"{tag}_gs"]), torch.tensor(z[f"{tag}_ms"])
        stub = SimpleNamespace(init_grid_features=gs, init_mesh_features=ms, _num_mesh_nodes=ms.shape[0],
                               _dyn_size=z[f"{tag}_x"].shape[1])
        out = OWP._preprocess_input(stub, torch.tensor(z[f"{tag}_x"]))
        assert torch.equal(out, torch.tensor(z[f"{tag}_out"]))


# ------------------------------------------------------------------------------------------------
# (b) the HIP path (through the C ABI) vs the same fixtures
# ------------------------------------------------------------------------------------------------
def _write_dataset(d, series, mean, std, flat):
    os.makedirs(d, exist_ok=True)
    series.tofile(os.path.join(d, "data.npy"))
    np.savez(os.path.join(d, "scalers.npz"), mean=mean, std=std, n=series.shape[0])
    info = ({"n_time": series.shape[0], "n_nodes": series.shape[1], "n_feat": series.shape[2], "flat": True} if flat else
            {"n_time": series.shape[0], "n_lon": series.shape[1], "n_lat": series.shape[2], "n_feat": series.shape[3]})
    with open(os.path.join(d, "dataset_info.json"), "w") as fh:
        json.dump(info, fh)


@pytest.mark.gpu
def test_hip_window_loader_matches_reference_bit_for_bit(tmp_path, lib_built):
    """graphcast-lite_amd/data.py::TimeseriesChunkDataset (series in HBM, gcl_window_pack) on the files the
    reference loader read: every split, grid / flat / multi-chunk layouts, bit-identical windows."""
    from graphcast_lite_amd.data import TimeseriesChunkDataset

    z = _npz("loader_vectors.npz")
    C = int(z["grid_C"])
    _write_dataset(str(tmp_path / "grid"), z["grid_series"], z["grid_mean"], z["grid_std"], False)
    for split in ("train", "test", "val", "test_only", "all"):
        ds = TimeseriesChunkDataset(str(tmp_path / "grid"), 2, 3, split, n_features=C, device=DEV)
        assert len(ds) == int(z[f"grid_{split}_len"])
        if len(ds):
            X, Y = ds.batch(range(len(ds)))
            assert np.array_equal(X.cpu().numpy(), z[f"grid_{split}_X"]) and np.array_equal(Y.cpu().numpy(), z[f"grid_{split}_Y"])
            x0, y0 = ds[len(ds) - 1]
            assert np.array_equal(x0.cpu().numpy(), z[f"grid_{split}_X"][-1])
    _write_dataset(str(tmp_path / "flat"), z["flat_series"], z["flat_mean"], z["flat_std"], True)
    ds = TimeseriesChunkDataset(str(tmp_path / "flat"), 3, 1, "all", device=DEV)
    X, Y = ds.batch(range(len(ds)))
    assert np.array_equal(X.cpu().numpy(), z["flat_all_X"]) and np.array_equal(Y.cpu().numpy(), z["flat_all_Y"])
    cd = tmp_path / "chunks"
    os.makedirs(cd)
    np.save(cd / "chunk_0.npy", z["grid_series"][:6])
    np.save(cd / "chunk_1.npy", z["grid_series"][6:])
    np.savez(cd / "scalers.npz", mean=z["grid_mean"], std=z["grid_std"], n=14)
    ds = TimeseriesChunkDataset(str(cd), 2, 1, "all", n_features=C, device=DEV)
    assert [list(p) for p in ds._sample_indices] == z["chunks_all_index"].tolist()
    X, Y = ds.batch(range(len(ds)))  # spans both chunks
    assert np.array_equal(X.cpu().numpy(), z["chunks_all_X"]) and np.array_equal(Y.cpu().numpy(), z["chunks_all_Y"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["enc", "dec", "odd", "single"])
def test_hip_mlp_matches_reference(tag, lib_built):
    """The product's MLP (gcl_linear_fwd / gcl_linear_bwd_all chain) with the reference's weights:
    output, input gradient and every parameter gradient, Frobenius AND element-wise."""
    from graphcast_lite_amd.config import MLPBlock
    from graphcast_lite_amd.models import MLP

    z = _npz("mlp_vectors.npz")
    m = _mlp_from_fixture(MLP, MLPBlock, z, tag, DEV)
    x = torch.tensor(z[f"{tag}_x"], device=DEV, requires_grad=True)
    y = m(x)
    y.backward(torch.tensor(z[f"{tag}_dy"], device=DEV))
    for name, got, want in [("y", y, z[f"{tag}_y"]), ("dx", x.grad, z[f"{tag}_dx"])] + [
            (k, p.grad, z[f"{tag}_g_{k}"]) for k, p in m.named_parameters()]:
        fro, mx = relmax(got, want)
        # PReLU slope gradients are sums over rows x width terms with cancellation: fp32 order noise
        tol = 1e-4 if np.asarray(want).size == 1 else 1e-5
        assert fro < tol and mx < 2 * tol, (tag, name, fro, mx)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_training_loop_matches_reference(tag, lib_built):
    """graphcast-lite_amd/train.py::batch_loss / train_epoch (loss + d_delta from gcl_wmse_fwd_bwd, the AR
    advance on the device) driving the stub model == the reference's train_epoch."""
    from graphcast_lite_amd.train import batch_loss, train_epoch

    z = _npz("train_loop_vectors.npz")
    spec, batches = CASES[tag], _batches(z, DEV)
    kw = _case_kwargs(z, spec, DEV)
    m = StubModel(z["obs"], z["W0"], z["b0"], DEV)
    X, y = batches[0]
    loss = batch_loss(m, X, y, 0.0, 0, 0, kw["lat_weights"], spec["ar"], kw["channel_mask"], kw["spatial_mask"],
                      kw["static_channels"], kw["forcing_channels"], kw["use_residual"])
    loss.backward()
    assert abs(float(loss.detach()) - float(z[f"{tag}_loss_first"])) <= 2e-6 * abs(float(z[f"{tag}_loss_first"]))
    for got, want in ((m.W.grad, z[f"{tag}_gW_first"]), (m.b.grad, z[f"{tag}_gb_first"])):
        fro, mx = relmax(got, want)
        assert fro < 1e-5 and mx < 2e-5, (tag, fro, mx)
    m = StubModel(z["obs"], z["W0"], z["b0"], DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    for ep in range(2):
        got = train_epoch(m, batches, opt, None, DEV, 0.0, ep, current_ar_steps=spec["ar"], **kw)
        assert abs(got - z[f"{tag}_epoch_losses"][ep]) <= 5e-6 * abs(z[f"{tag}_epoch_losses"][ep])
    assert max(relmax(m.W, z[f"{tag}_W"])) < 2e-5 and max(relmax(m.b, z[f"{tag}_b"])) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(EVAL_CASES) + ["eval_single_target"])
def test_hip_evaluation_loop_matches_reference(tag, lib_built):
    from graphcast_lite_amd.train import test as hip_test

    z = _npz("train_loop_vectors.npz")
    m = StubModel(z["obs"], z["W0"], z["b0"], DEV)
    batches = _batches(z, DEV)
    if tag == "eval_single_target":
        C = int(z["C"])
        batches = [(X, Y[..., :C].contiguous()) for X, Y in batches]
        kw = _case_kwargs(z, dict(lat=True), DEV)
    else:
        kw = _case_kwargs(z, EVAL_CASES[tag], DEV)
    got = hip_test(m, batches, None, DEV, **kw)
    for a, b in zip(got, z[tag]):
        assert abs(a - b) <= 5e-6 * max(1.0, abs(b)), (tag, got, z[tag])


@pytest.mark.gpu
def test_hip_input_assembly_matches_reference(lib_built):
    from graphcast_lite_amd import hip

    z = _npz("assemble_vectors.npz")
    for tag in ("a", "b"):
        out = hip.assemble_input(torch.tensor(z[f"{tag}_x"], device=DEV).unsqueeze(0), torch.tensor(z[f"{tag}_gs"], device=DEV),
                                 torch.tensor(z[f"{tag}_ms"], device=DEV))
        assert torch.equal(out[0].cpu(), torch.tensor(z[f"{tag}_out"]))
