"""Gradient parity by fp64 arbitration (test infrastructure).

Two fp32 implementations of the same sums (the CPU oracle and the HIP path) differ by rounding and by
summation order, so comparing them with each other needs a tolerance that nobody can derive.  Instead the
oracle is run ONCE MORE in float64 and both fp32 results are measured against that:

    ||g_hip - g64||  <=  2 ||g_oracle32 - g64||  +  1e-5 ||g64||                         (per parameter)
    max|g_hip - g64| <=  2 max|g_oracle32 - g64| +  1e-5 max|g64|   (element-wise, same factors)

i.e. the HIP gradient may be at most twice as far from the exact value as the reference's own fp32 arithmetic
is, plus the stated 1e-5 bar.  A gradient that is structurally ~0 (e.g. GAT `att_dst` when the pre-activations
of a node's in-edges share a sign: 5e-19 in fp64) holds only rounding noise in BOTH fp32 runs; for those a floor
of 1e-6 of the largest gradient norm inside the SAME module (the conv layer the parameter belongs to, not the
whole model) applies.  A scalar PReLU slope is a cancelling sum over every element of the layers that share it, so
its rounding noise scales with THEIR gradients: its floor group is the stack / MLP that owns it.
`report()` prints the per-parameter table into the pytest log (`-rP` / on failure)."""
import copy

import torch


def oracle_fp64(o):
    o64 = copy.deepcopy(o).double()
    for name in ("init_grid_features", "init_mesh_features"):
        if getattr(o64, name, None) is not None:
            setattr(o64, name, getattr(o64, name).double())
    for p in o64.parameters():
        p.grad = None
    return o64


def _module_key(name: str, numel: int = 0) -> str:
    """Prefix of the parameter names that form this parameter's floor group."""
    key = name.rsplit(".", 1)[0]
    if key.endswith(".lin"):
        return key[:-4]
    if numel == 1:  # PReLU slope: `<stack>.activation.weight` (alias `<stack>.layers.k.weight`) or `<mlp>.MLP.k.weight`
        return key.rsplit(".", 2)[0] if ".layers." in key or ".MLP." in key else key.rsplit(".", 1)[0]
    return key


def check_grads(hip_named, o32_named, o64_named, tag="", rel_bar=1e-5, factor=2.0, module_floor=1e-6, verbose=True):
    """hip_named / o32_named / o64_named: dicts name -> gradient tensor (None = no gradient)."""
    rows, failures = [], []
    g64n = {k: float(v.double().norm()) for k, v in o64_named.items() if v is not None}

    def group_max(name, numel):
        pre = _module_key(name, numel)
        return max(v for k, v in g64n.items() if k.startswith(pre))
    for name, g64 in o64_named.items():
        gh, g32 = hip_named.get(name), o32_named.get(name)
        if g64 is None:
            assert gh is None or float(gh.abs().sum()) == 0.0, f"{tag}: {name} has a gradient on the HIP path only"
            continue
        assert gh is not None, f"{tag}: {name} has no gradient on the HIP path"
        g64 = g64.double().cpu()
        dh, d32 = gh.double().cpu() - g64, g32.double().cpu() - g64
        n64, m64 = float(g64.norm()), float(g64.abs().max())
        floor = module_floor * group_max(name, g64.numel())
        fro_h, fro_32 = float(dh.norm()), float(d32.norm())
        max_h, max_32 = float(dh.abs().max()), float(d32.abs().max())
        ok_fro = fro_h <= factor * fro_32 + rel_bar * n64 + floor
        ok_max = max_h <= factor * max_32 + rel_bar * m64 + floor
        rows.append((name, n64, fro_h / (n64 + 1e-300), fro_32 / (n64 + 1e-300), max_h / (m64 + 1e-300),
                     max_32 / (m64 + 1e-300), ok_fro and ok_max))
        if not (ok_fro and ok_max):
            failures.append(name)
    if verbose:
        report(rows, tag)
    assert not failures, f"{tag}: gradients outside the fp64-arbitrated bound: {failures}"
    return rows


def report(rows, tag=""):
    print(f"\n[{tag}] per-parameter gradient error vs the float64 oracle (relative to ||g64|| / max|g64|)")
    print(f"{'parameter':58s} {'||g64||':>10s} {'hip fro':>9s} {'o32 fro':>9s} {'hip max':>9s} {'o32 max':>9s}")
    for name, n64, fh, f32, mh, m32, ok in rows:
        print(f"{name[-58:]:58s} {n64:10.3e} {fh:9.2e} {f32:9.2e} {mh:9.2e} {m32:9.2e} {'' if ok else '  <-- FAIL'}")


def grads_of(module):
    return {k: (None if p.grad is None else p.grad.detach()) for k, p in module.named_parameters()}
