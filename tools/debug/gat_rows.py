"""Debug aid: per-row error of the GAT backward (dh @ W vs the oracle's x.grad) on the [3,5] mesh, H=8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_hip_ops import build_graphs, experiment, rnd, P, DEV
from graphcast_lite_amd import hip

levels, H, C, B = [3, 5], int(os.environ.get("H", 8)), 64, int(os.environ.get("B", 9))
g = build_graphs(experiment("baseline", mesh_levels=levels))
n = g["M"]; Fin = 24
x = rnd(B, n, Fin, seed=1).double().requires_grad_()
W = rnd(H * C, Fin, seed=2, scale=0.3).double().requires_grad_()
a_s, a_d = rnd(1, H, C, seed=3, scale=0.3).double().requires_grad_(), rnd(1, H, C, seed=4, scale=0.3).double().requires_grad_()
b = rnd(C, seed=5).double().requires_grad_()
y_ref, ei2, alpha_ref = P.gat_conv(x, g["proc"], W, a_s, a_d, b, H)
dy = rnd(B, n, C, seed=6)
y_ref.backward(dy.double())
h_ref = (x.detach().float() @ W.detach().float().t())
G = hip.Graph(g["proc"], n, hip.GRAPH_GAT)
hd = h_ref.to(DEV)
f = lambda t: t.detach().float().reshape(-1).to(DEV)
y, s_src, s_dst, alpha = hip.gat_fwd(G, hd, f(a_s), f(a_d), b.detach().float().to(DEV), H, C)
print("fwd y err", float((y.cpu().double() - y_ref).abs().max() / y_ref.abs().max()))
d_as, d_ad, d_b = torch.empty(H * C, device=DEV), torch.empty(H * C, device=DEV), torch.empty(C, device=DEV)
dh = hip.gat_bwd(G, dy.to(DEV), hd, f(a_s), f(a_d), s_src, s_dst, alpha, d_as, d_ad, d_b, False, H, C)
dx = dh.cpu().double() @ W.detach()
err = (dx - x.grad).abs()
print("max err", float(err.max()), "max ref", float(x.grad.abs().max()), "fro", float(err.norm() / x.grad.norm()))
rowerr = err.amax(dim=(0, 2))
top = torch.topk(rowerr, 12)
ei = ei2
indeg = torch.bincount(ei[1], minlength=n); outdeg = torch.bincount(ei[0], minlength=n)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    print(f"row {i:6d} err {v:.3e} in-deg {int(indeg[i])} out-deg {int(outdeg[i])}  |x.grad|max {float(x.grad[:, i].abs().max()):.3e}")
print("rows with err > 1e-5*max:", int((rowerr > 1e-5 * x.grad.abs().max()).sum()), "of", n)
be = err.amax(dim=(1, 2)); print("per-batch max err", [f"{v:.2e}" for v in be.tolist()])
