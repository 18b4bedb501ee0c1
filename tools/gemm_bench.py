"""Times the dense entry points on wide shapes (TFLOP/s of the exact-fp32 MFMA path)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from graphcast_lite_amd import hip  # noqa: E402

dev = "cuda:0"


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


import os
shapes = [(327680, 256, 256), (327680, 512, 256), (40962, 256, 512), (40962, 256, 256), (81924, 256, 256), (327680, 128, 128), (327680, 256, 128)]
if os.environ.get('GCL_DENSE_IMPL'):
    shapes = [(327680, 128, 128), (786432, 64, 128), (786432, 128, 64), (786432, 64, 64), (327680, 128, 256), (327680, 256, 64)]
for rows, K, N in shapes:
    x = torch.randn(rows, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    dy = torch.randn(rows, N, device=dev)
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    fl = 2.0 * rows * K * N
    for act in (0, 2):
        tf = t(lambda: hip.dense_fwd(x, W, b, act))
        tdx = t(lambda: hip.dense_bwd_dx(dy, W, x if act else None, act))
        tdw = t(lambda: hip.dense_bwd_dw(dy, x, dW, db, True, act))
        tt = t(lambda: torch.matmul(x, W.t()))
        print(f"rows={rows} K={K} N={N} act={act}: fwd {tf*1e3:.3f} ms {fl/tf/1e12:.1f} TF | dx {tdx*1e3:.3f} ms "
              f"{fl/tdx/1e12:.1f} TF | dw {tdw*1e3:.3f} ms {fl/tdw/1e12:.1f} TF | torch.matmul {tt*1e3:.3f} ms "
              f"{fl/tt/1e12:.1f} TF", flush=True)
