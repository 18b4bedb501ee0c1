"""Debug aid: error of the PReLU slope gradient of linear_bwd_all against fp64 (run with GCL_X3=0/1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from graphcast_lite_amd import hip
DEV = "cuda:0"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale
for rows, Fin, Fout in ((8192, 64, 64), (5000, 48, 36), (65536, 64, 64)):
    x, W, dy = rnd(rows, Fin, seed=21), rnd(Fout, Fin, seed=22, scale=0.3), rnd(rows, Fout, seed=23)
    a = torch.tensor([0.25])
    xr, Wr, ar = x.double().requires_grad_(), W.double().requires_grad_(), a.double().requires_grad_()
    (torch.where(xr > 0, xr, ar * xr) @ Wr.t()).backward(dy.double())
    dW, db = torch.empty(Fout, Fin, device=DEV), torch.empty(Fout, device=DEV)
    da = torch.zeros(1, device=DEV)
    dx = hip.linear_bwd_all(dy.to(DEV), W.to(DEV), x.to(DEV), a.to(DEV), da, dW, db, None, False)
    # the same sum formed from the kernel's OWN dx in fp64: separates "dx is off" from "the sum is off"
    dxpre = dx.cpu().double() / torch.where(x > 0, torch.ones(()), a).double()
    own = float((dxpre * x.double())[x <= 0].sum())
    print(f"[X3={os.environ.get('GCL_X3', '1')}] {rows}x{Fin}->{Fout}: d_slope {float(da):.6f} ref {float(ar.grad):.6f} err {float(da) - float(ar.grad):+.3e}"
          f"  (fp64 sum over the kernel's own dx: err {own - float(ar.grad):+.3e})")
