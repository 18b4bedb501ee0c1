"""Debug aid: LayerNorm backward with a dense dy vs a row-mapped dy (mesh shape of cfg A)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graphcast_lite_amd import hip
dev = torch.device("cuda:0")
B, M, U, G, F = 64, 10242, 4804, 2048, 64
x = torch.randn(B * M, F, device=dev); gm = torch.rand(F, device=dev) + 0.5; bt = torch.randn(F, device=dev)
y, st = hip.layernorm_fwd(x, gm, bt)
g = torch.randn(B, G + U, F, device=dev)
used = torch.sort(torch.randperm(M)[:U]).values
pos = torch.full((M,), -1, dtype=torch.int32); pos[used] = (G + torch.arange(U)).to(torch.int32); pos = pos.to(dev)
dense = torch.zeros(B, M, F, device=dev); dense[:, used.to(dev)] = g[:, G:]
dg, db, cs = torch.empty(F, device=dev), torch.empty(F, device=dev), torch.empty(F, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
a = hip.layernorm_bwd(dense.view(B * M, F), x, gm, st, dg, db, False, colsum_dx=cs)
b = hip.layernorm_bwd(None, x, gm, st, dg, db, False, colsum_dx=cs, dy_map=(g, pos))
print("max diff", float((a - b).abs().max()))
print("dense  us", t(lambda: hip.layernorm_bwd(dense.view(B * M, F), x, gm, st, dg, db, False, colsum_dx=cs)))
print("mapped us", t(lambda: hip.layernorm_bwd(None, x, gm, st, dg, db, False, colsum_dx=cs, dy_map=(g, pos))))
