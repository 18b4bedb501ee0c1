"""Headline benchmark: training samples/sec of the encode-process-decode GNN on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step on a batch of synthetic 6-hour samples (forward, weighted-MSE
loss with residual add, backward, gradient all-reduce when N > 1, Adam), batch fixed PER GPU
(weak scaling).  Default workload = BASELINE.json configs[1]: Baseline GCN processor on
wb2_64x32 (2048 grid nodes, 33 features, obs window 2, mesh levels [3,5]), 64 samples per GPU.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the mesh
GCNConv aggregation kernel (HIP events on the launch stream, inside the timed region) and
`cpu_baseline` = the CPU oracle (PyG op sequence, batch 1) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC is the only mode this pool's driver supports (RCCL between processes needs it); the box
# exports it already - keep it if the launcher's environment was rebuilt
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4 copy is 6290
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's headline 5 PF includes 2:1 sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md): 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz


class LaunchProbe:
    """HIP events (recorded on the launch stream) around the launches of the roofline kernels; installed as
    `hip.PROBE` only while the roofline leg of the benchmark runs."""

    def __init__(self, match):
        self.match, self.events = match, {}

    def begin(self, kind, **info):
        key = self.match(kind, info)
        if key is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return key, e0, e1

    def end(self, tok):
        key, e0, e1 = tok
        e1.record()
        self.events.setdefault(key, []).append((e0, e1))

    def mean_ms(self, key):
        ev = self.events.get(key, [])
        return (sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)) if ev else (None, 0)


def build_model(name, device, seed=42):
    from graphcast_lite_amd.experiments import GRID, experiment
    from graphcast_lite_amd.models import WeatherPrediction

    cfg = experiment(name)
    nlat, nlon = GRID[name]
    lats = np.linspace(-90, 90, nlat, endpoint=True)
    lons = np.linspace(0, 360, nlon, endpoint=False)
    torch.manual_seed(seed)
    return cfg, WeatherPrediction((lats, lons), cfg.graph, cfg.pipeline, cfg.data, device), (nlat, nlon)


def synthetic_batch(cfg, G, B, seed):
    g = torch.Generator().manual_seed(seed)
    F = cfg.data.num_features_used
    X = torch.randn(B, G, cfg.data.obs_window_used * F, generator=g)
    y = X[..., -F:] + 0.1 * torch.randn(B, G, F, generator=g)
    return X, y


def cpu_baseline(cfg, model, grid, budget_s=12.0, max_steps=2000):
    """The CPU oracle executing PyG's op sequence at batch 1: forward +
    weighted MSE + backward + torch Adam, on all host cores of this box."""
    from oracle import model as omodel
    from oracle import train_step as ostep

    # threads actually available to this process (the GPU box gives a 1-GPU job a 16-CPU share;
    # os.cpu_count() reports the whole host and oversubscribing it is >100x slower)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    o = omodel.WeatherPrediction(
        cfg.pipeline, cfg.data, num_grid_nodes=model._num_grid_nodes, num_mesh_nodes=model._num_mesh_nodes,
        encoding_graph=model.encoding_graph.cpu(), processing_graph=model.processing_graph.cpu(),
        decoding_graph=model.decoding_graph.cpu(), init_grid_features=model.init_grid_features.cpu(),
        init_mesh_features=model.init_mesh_features.cpu(),
        processing_edge_features=model._processing_edge_features.cpu())
    o.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    opt = torch.optim.Adam(o.parameters(), lr=1e-3)
    X, y = synthetic_batch(cfg, model._num_grid_nodes, 1, seed=1234)
    lw = ostep.get_lat_weights(*grid)

    def one():
        opt.zero_grad()
        loss = ostep.train_step_loss(o, X, y, lat_weights=lw)
        loss.backward()
        opt.step()

    tw = time.perf_counter()
    one()
    if time.perf_counter() - tw < 4.0:  # a second warm-up only when steps are short
        one()
    n, t0 = 0, time.perf_counter()
    while n < max_steps and (n == 0 or (time.perf_counter() - t0) < budget_s):
        one()
        n += 1
    dt = time.perf_counter() - t0
    # the same step on ONE thread (SURVEY.md §8d asks for both), a few seconds of it
    torch.set_num_threads(1)
    n1, t1 = 0, time.perf_counter()
    while n1 == 0 or (n1 < max_steps and (time.perf_counter() - t1) < min(4.0, budget_s / 3)):
        one()
        n1 += 1
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    return {"value": n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} optimiser steps at batch 1 (fwd+loss+bwd+Adam) of the same config, {dt:.1f} s",
            "value_1thread": n1 / dt1}


WORKLOADS = {  # --config -> the BASELINE.json entry it measures (None: not a BASELINE.json config)
    "baseline": "BASELINE.json configs[1]: Baseline GCN processor, wb2_64x32 33-feat, batch=64 samples",
    "attention": "BASELINE.json configs[2]: Attention (GATConv processor, H=1) wb2_64x32, batch=64",
    "wb2_512x256_19f_ar": "BASELINE.json configs[3]: Baseline GCN, wb2_512x256 19-feat, batch=8 per GPU",
    "wb2_512x256_sparse_gat": "BASELINE.json configs[4]: SparseGATConv processor with edge pruning, wb2_512x256",
}


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset): start N fresh rank processes,
    one per GPU, BEFORE this process makes any GPU call (it never makes one), with the torchrun
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 prints the JSON
    line on the inherited stdout.  Returns non-zero - and prints no benchmark line of its own - when the
    ranks cannot be started or any of them fails: an N-GPU number is only ever produced by N ranks."""
    import subprocess

    one_dev = os.environ.get("GCL_BENCH_ONE_DEVICE", "0") == "1"
    selftest = os.environ.get("GCL_BENCH_SELFTEST", "0") == "1"
    if not (one_dev or selftest):
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < n:
            print(f"bench.py: --gpus {n} requested but this node exposes {have} GPU(s); refusing to report an "
                  f"{n}-GPU number from fewer devices", file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GCL_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        deadline = time.time() + float(os.environ.get("GCL_BENCH_RANK_TIMEOUT", "1500"))
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is not None:
                    pending.remove(pr)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
            if rc != 0 or time.time() > deadline:
                rc = rc or 3
                break
            time.sleep(0.05)
    finally:
        for pr in procs:  # exactly the processes started above, by handle
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            pr.wait()
    if rc != 0:
        print(f"bench.py: a rank process failed (rc {rc}); no benchmark line", file=sys.stderr, flush=True)
    return rc


def init_dist(args):
    """(world, rank, local) from the launcher's environment; the process group is RCCL ("nccl") unless the
    one-GPU rehearsal knob GCL_DIST_BACKEND=gloo is set.  `--gpus` must equal the real world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GCL_DIST_BACKEND", "nccl")
        if backend == "nccl" and local < torch.cuda.device_count():
            torch.cuda.set_device(local)  # RCCL binds its communicator to the current device: set it BEFORE the group exists
        dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()  # what the process group really has
    return world, rank, local


def selftest(args) -> None:
    """GCL_BENCH_SELFTEST=1 (CPU, gloo): exercises only the launcher + rendezvous + reduction plumbing of
    this file and prints a line whose `n_gpus` comes from dist.get_world_size()."""
    os.environ.setdefault("GCL_DIST_BACKEND", "gloo")
    world, rank, _ = init_dist(args)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": float(t.item()),
                          "spawned": os.environ.get("GCL_BENCH_SPAWNED", "0") == "1"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (default: 64 at 64x32, 8 at 512x256)")
    ap.add_argument("--config", default="baseline",
                    choices=["baseline", "attention", "attention_h4", "sparse_attention", "wb2_512x256_19f_ar",
                             "wb2_512x256_19f_ar_v2", "region_krsk_cds_19f", "wb2_512x256_sparse_gat"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel of the timed steps individually instead of replaying the captured "
                         "hipGraph (then the roofline events are recorded inside the timed region itself)")
    ap.add_argument("--graph", action="store_true", help="(default) kept for older command lines")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher (one child process per GPU; this process stays off the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("GCL_BENCH_SELFTEST", "0") == "1":
        return selftest(args)

    # rehearsal knobs for a one-GPU box (never used by the driver): GCL_DIST_BACKEND=gloo lets several
    # ranks share device 0 (RCCL refuses two ranks on one device), GCL_BENCH_ONE_DEVICE=1 maps them there
    world, rank, local = init_dist(args)
    if os.environ.get("GCL_BENCH_ONE_DEVICE", "0") == "1":
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from graphcast_lite_amd import hip, models
    from graphcast_lite_amd.train import TrainStep, get_lat_weights

    hip.lib()
    cfg, model, grid = build_model(args.config, dev)
    B = args.batch or {"wb2_512x256_19f_ar": 8, "wb2_512x256_sparse_gat": 8, "wb2_512x256_19f_ar_v2": 1}.get(args.config, 64)
    G, M = model._num_grid_nodes, model._num_mesh_nodes
    X, y = synthetic_batch(cfg, G, B, seed=1234 + rank)  # every rank its own samples
    X, y = X.to(dev), y.to(dev)                           # resident in HBM before the timed region
    step = TrainStep(model, lr=1e-3, lat_weights=get_lat_weights(grid[0], grid[1], dev), world_size=world,
                     use_graph=False if args.eager else None)

    # Untimed settle phase before the W official warm-up steps: a freshly acquired box can run the
    # first seconds several times slower (clock ramp / code-object and allocator warm-up).  Step until
    # three consecutive steps are within 15 % of the fastest seen (at most 60 steps / 5 s).
    # With several ranks every step contains a collective, so all ranks must run the SAME number of
    # steps: a fixed count there, the adaptive rule only in the single-process case.
    settle, best, streak, t_end = 0, float("inf"), 0, time.perf_counter() + 5.0
    while settle < (12 if world > 1 else 60) and (world > 1 or time.perf_counter() < t_end):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step(X, y)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        settle += 1
        best = min(best, dt1)
        streak = streak + 1 if dt1 <= 1.15 * best else 0
        if world == 1 and settle >= 5 and streak >= 3:
            break
    for _ in range(args.warmup):
        loss = step(X, y)
    is_gcn = cfg.pipeline.processor.gcn.layer_type.value == "conv_gcn"
    is_inet = cfg.pipeline.processor.gcn.layer_type.value == "interaction_net"

    is_gat = cfg.pipeline.processor.gcn.layer_type.value in ("conv_gat", "sparse_gat")

    def arm_profile():
        # HIP events around the roofline kernel's launches (recorded on the launch stream)
        pg = models._graphs.get(model.processor_graph(), M, hip.GRAPH_GCN) if is_gcn else None
        D = cfg.pipeline.processor.gcn.output_dim
        E = int(model.processing_graph.shape[1])

        def match(kind, info):
            if is_gcn and kind in ("aggregate", "gcn_layer_fwd") and info["graph"] is pg:
                return kind + ("_T" if info.get("transpose") else "")
            if is_gat and kind in ("gat_fwd", "gat_bwd") and info["graph"].n == M:
                gat_seen["graph"], gat_seen["H"], gat_seen["C"] = info["graph"], info["H"], info["C"]
                gat_seen["alpha"] = bool(info.get("alpha", True)) or gat_seen.get("alpha", False)
                return kind
            if is_inet and kind == "dense_fwd" and (info["rows"], info["Fin"], info["Fout"]) == (B * E, D, D):
                return "edge_mlp"  # the edge-MLP contractions [B*E, D] x [D, D]: 2 per message-passing step, forward
            return None

        hip.PROBE = LaunchProbe(match)
        return hip.PROBE

    gat_seen = {}
    bufs = step.input_buffers()
    if bufs is not None:  # the synthetic batch lives in the buffers the captured graph reads (a loader would fill them in place)
        bufs[0].copy_(X)
        bufs[1].copy_(y)
        X, y = bufs
    probe = None
    replayed = step.graph_active
    if not replayed:
        probe = arm_profile()  # eager timed region: the events sit inside it
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(X, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    roof_steps, roof_ms = args.steps, None
    roof_pass = replayed
    if world > 1:  # the eager roofline pass contains collectives: every rank runs it if any rank needs it
        f = torch.tensor([1.0 if replayed else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        roof_pass = bool(f.item() > 0)
    if roof_pass:
        # Kernels replayed from a hipGraph cannot be bracketed by timing events on ROCm, so the
        # roofline kernel is timed on a few eager steps of the SAME step function right after the
        # timed region (same buffers, same launches; the rocprofv3 trace in profiles/ covers both).
        for _ in range(2):  # the eager path allocates its intermediates outside the graph's pool: let the allocator settle
            loss = step._eager(X, y)
        probe = arm_profile()
        roof_steps = max(3, min(args.steps, 10))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(roof_steps):
            loss = step._eager(X, y)
        torch.cuda.synchronize()
        roof_ms = (time.perf_counter() - t1) / roof_steps * 1e3
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # HBM-side bytes per launch of the reported kernel from this round's rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    # correction + WRITE_SIZE; separate --pmc runs: tools/pmc.sh, written by tools/pmc_roofline.py), committed under
    # profiles/ per config: {"config", "batch", "kernels": {kernel-name substring: bytes per launch}}
    def traffic_of(kernel_tag):
        try:
            with open(os.path.join(ROOT, "profiles", f"pmc_roofline_{args.config}.json")) as fh:
                pmc = json.load(fh)
            if pmc.get("config") == args.config and pmc.get("batch") == B:
                return pmc["kernels"].get(kernel_tag)
        except Exception:
            pass
        return None

    roof = None
    # more than a second of replays after the official region: the +-7 % box / thermal spread shows in the record
    sustained = None
    if world == 1 and step.graph_active:
        torch.cuda.synchronize()
        ts, ns = time.perf_counter(), 0
        while ns < 20 or time.perf_counter() - ts < 1.2:
            loss = step(X, y)
            ns += 1
        torch.cuda.synchronize()
        sustained = {"samples_per_s": B * ns / (time.perf_counter() - ts), "steps": ns}

    def copy_ceiling_gbs():
        """Stream-copy ceiling of THIS box: 256 MiB device-to-device copy, read + write bytes."""
        a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        return 2.0 * a.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    hip.PROBE = None
    if is_gcn and probe is not None:
        F = cfg.pipeline.processor.gcn.output_dim
        pg = models._graphs.get(model.processor_graph(), M, hip.GRAPH_GCN)
        Ep = pg.e
        per_sample = 4 * M * (F + F) + 4 * Ep + 4 * (M + 1) + 4 * M  # one mesh GCNConv layer, per sample
        copy_gbs = copy_ceiling_gbs()

        def line(key, what):
            ms, cnt = probe.mean_ms(key)
            if ms is None:
                return None
            ach = B * per_sample / (ms * 1e-3) / 1e9
            return {"kernel": what, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "avg_launch_us": ms * 1e3, "launches_timed": cnt, "frac_of_copy_ceiling": ach / copy_gbs}

        # the scatter-gather kernel of the mesh processor: forward launches when the layer runs as linear +
        # aggregate, transposed (backward) launches otherwise - the mesh graph is symmetric, both move the same
        # algorithmic bytes through the same agg_kernel instantiation
        halo = pg.halo_info(True, 64) or pg.halo_info(True, 32)
        kname = "agg_halo_loop_kernel" if halo else "agg_kernel"
        agg = line("aggregate", f"{kname} (mesh GCNConv aggregate, forward)") or \
            line("aggregate_T", f"{kname} (mesh GCNConv aggregate, transposed CSR = backward of the layer)")
        fh = pg.halo_info(False, 64)
        fname = "gcn_halo_fwd_kernel" if (fh and F % 16 == 0 and 32 < F <= 64 and Ep >= 6 * M and os.environ.get("GCL_GCN_HALO") != "0") else "gcn_fwd_kernel"
        fused = line("gcn_layer_fwd", f"{fname} (whole mesh GCNConv layer forward in one launch: gather + dense)")
        roof = agg or fused
        if roof is not None:
            roof = dict(roof, bound="hbm", traffic=traffic_of(kname if roof is agg else fname),
                        bytes_per_launch=B * per_sample, copy_ceiling_gbs=copy_gbs,
                        achieved_gather_counted_gbs=B * (per_sample + 4 * Ep * F) / (roof["avg_launch_us"] * 1e-6) / 1e9)
            if fused is not None and roof is not fused:
                # same algorithmic bytes (X in, Y out, CSR): the one-kernel layer replaces linear + aggregate
                roof["gcn_layer_one_kernel"] = dict(
                    fused, traffic=traffic_of(fname),
                    note="average over the mesh layers of a step; where the compact pipeline is on, the FIRST of them reads its "
                         "input rows through a row table from the encoder output (batch-invariant rows from one shared copy), "
                         "so it moves fewer bytes than the algorithmic count of a dense [B, n, F] input")
    if is_gat and probe is not None and probe.events.get("gat_fwd"):
        # GATConv / SparseGATConv processor: the attention aggregation kernel (scores + neighbour softmax + weighted sum).
        # Algorithmic bytes per sample: the GCN aggregation's + 8 n (a_src, a_dst) + 4 E' H when alpha is kept
        gg, Hh, Cc = gat_seen["graph"], gat_seen["H"], gat_seen["C"]
        per_sample = 4 * M * (Hh * Cc + Cc) + 4 * gg.e + 4 * (M + 1) + 4 * M + 8 * M * Hh + (4 * gg.e * Hh if gat_seen.get("alpha") else 0)
        copy_gbs = copy_ceiling_gbs()
        ms, cnt = probe.mean_ms("gat_fwd")
        ach = B * per_sample / (ms * 1e-3) / 1e9
        hi = gg.halo_info(False, 64)
        staged = Hh == 1 and hi is not None and (hi[2] + 1) * Cc * 4 + (hi[2] + 65) * 4 <= 80 * 1024 and os.environ.get("GCL_GAT_HALO") != "0"
        gname = "gat_halo_fwd_kernel" if staged else "gat_fwd_kernel"
        roof = {"bound": "hbm", "kernel": f"{gname} (mesh GATConv: scores, softmax over neighbours, aggregation; forward)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic_of(gname),
                "bytes_per_launch": B * per_sample, "avg_launch_us": ms * 1e3, "launches_timed": cnt,
                "copy_ceiling_gbs": copy_gbs, "frac_of_copy_ceiling": ach / copy_gbs}
        msb, cntb = probe.mean_ms("gat_bwd")
        if msb is not None:
            per_b = per_sample + 4 * M * Hh * Cc + 4 * M * Cc  # + dy read, dh written
            roof["gat_bwd"] = {"kernel": "gat_bwd (dst-side, src-side and attention-vector kernels of one layer)",
                               "achieved": B * per_b / (msb * 1e-3) / 1e9, "frac": B * per_b / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "avg_launch_us": msb * 1e3, "launches_timed": cntb, "bytes_per_launch": B * per_b}
    if is_inet and probe is not None and probe.events.get("edge_mlp"):
        ms, cnt = probe.mean_ms("edge_mlp")
        D, E = cfg.pipeline.processor.gcn.output_dim, int(model.processing_graph.shape[1])
        flops = 2.0 * B * E * D * D
        achieved = flops / (ms * 1e-3) / 1e12
        x3 = os.environ.get("GCL_X3", "1") != "0" and os.environ.get("GCL_X3_TILE", "1") != "0"
        if x3:
            # the contraction runs on the bf16 matrix pipe: every fp32 product costs SIX bf16 MFMA products (csrc/x3.h), so
            # the executed matrix flops are 6x the algorithmic ones and the peak is the dense bf16 one
            roof = {"bound": "mfma", "kernel": "gemm_tile_x3_kernel (InteractionNet edge MLP, [B*E, D] x [D, D], forward; "
                                               "fp32 operands split exactly into 3 bf16 pieces, 6 piece products each)",
                    "achieved": 6.0 * achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6.0 * achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "flops_per_launch": 6.0 * flops,
                    "fp32_equivalent_tflops": achieved, "fp32_operand_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                    "avg_launch_us": ms * 1e3, "launches_timed": cnt}
        else:
            roof = {"bound": "mfma", "kernel": "gemm_tile_kernel (InteractionNet edge MLP, [B*E, D] x [D, D], forward)",
                    "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                    "traffic": None, "flops_per_launch": flops, "avg_launch_us": ms * 1e3, "launches_timed": cnt}

    if rank == 0:
        out = {
            "metric": "training samples/sec (6h windows) + achieved HBM GB/s on mesh GCNConv",
            "value": world * B * args.steps / dt, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{WORKLOADS.get(args.config) or args.config + ' (not a BASELINE.json config; SURVEY.md 8f)'}"
                                   f" - grid {grid[1]}x{grid[0]}, G={G}, mesh M={M}, {cfg.data.num_features_used} feat, "
                                   f"obs {cfg.data.obs_window_used}, AR 1, fwd+loss+bwd+Adam",
                       "experiment": args.config, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "world_size": world,
                       "dist_backend": (dist.get_backend() if world > 1 else None), "final_loss": final_loss,
                       "launch_mode": step.launch_mode,
                       "arithmetic": "fp32 storage and accumulation; dense K, N <= 64 and the wide tile contractions run on the "
                                     "bf16 matrix pipe with every fp32 operand split exactly into 3 bf16 pieces (6 of the 9 piece "
                                     "products, the dropped ones <= 2^-25 relative); aggregations are plain fp32 (no matrix pipe)",
                       "sustained": sustained,
                       "settle_steps_before_warmup": settle,
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)},
            "roofline": roof,
        }
        if roof is not None:
            roof["measured_on"] = (f"{roof_steps} eager steps right after the timed region ({roof_ms:.2f} ms/step eager)"
                                   if roof_pass else "the timed steps themselves")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, model, grid)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
