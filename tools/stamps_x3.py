"""Diagnostic: where does a wave of the split-operand dense forward spend its cycles?  Needs the stamped build:
    make -C graphcast-lite_amd/csrc STAMPS=1 && GCL_LIB=graphcast-lite_amd/libgcl_hip_stamps.so python tools/stamps_x3.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphcast_lite_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
L.gcl_debug_read_stamps_x3.argtypes = [C.c_void_p, C.c_int]
rows, F = 655488, 64
x, W, b = torch.randn(rows, F, device=dev), torch.randn(F, F, device=dev) * 0.1, torch.randn(F, device=dev)
sl = torch.tensor([0.25], device=dev)
for _ in range(3):
    y = hip.linear_fwd(x, W, b, sl)
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.uint64)
assert L.gcl_debug_read_stamps_x3(buf.ctypes.data, buf.size) == 0
st = buf.reshape(-1, 8)[:512 * 4].astype(np.float64)
names = ["loop top", "wait loads", "commit", "issue", "mfma", "store"]
tot = st[:, :6].sum(axis=1)
print(f"cycles per wave (median) {np.median(tot):.0f} over {(rows + 127) // 128 / 512:.1f} tiles; shares: "
      + ", ".join(f"{nm} {st[:, i].sum() / tot.sum():.2f}" for i, nm in enumerate(names)))
print("   per-phase median cycles per wave:", [int(np.median(st[:, i])) for i in range(6)])
print("   set-up cycles per wave (median / max):", int(np.median(st[:, 6])), int(st[:, 6].max()))

# fused backward, same shape
dy, P = torch.randn(rows, F, device=dev), torch.randn(rows, F, device=dev)
dW, db, cs, da = torch.empty(F, F, device=dev), torch.empty(F, device=dev), torch.empty(F, device=dev), torch.zeros(1, device=dev)
for _ in range(3):
    hip.linear_bwd_all(dy, W, P, sl, da, dW, db, cs, False)
torch.cuda.synchronize()
assert L.gcl_debug_read_stamps_x3(buf.ctypes.data, buf.size) == 0
st = buf.reshape(-1, 8)[:512 * 4].astype(np.float64)
names = ["wait loads", "split+commit", "barrier 1", "issue", "dX mfma+stage", "dW mfma", "barrier 2", "finish+store"]
tot = st.sum(axis=1)
print(f"backward: cycles per wave (median) {np.median(tot):.0f} over {(rows + 63) // 64 / 512:.1f} tiles; shares: "
      + ", ".join(f"{nm} {st[:, i].sum() / tot.sum():.2f}" for i, nm in enumerate(names)))
print("   per-phase median cycles per wave:", [int(np.median(st[:, i])) for i in range(8)])
