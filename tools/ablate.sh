#!/bin/bash
# Timing-only ablations of the dense forward kernels (GCL_ABLATE bits: 1 no stores, 2 no MFMA, 4 no HBM loads,
# 16 no operand split [x3 kernel]).   bash tools/ablate.sh [tag]  -> gpurun_out/ablate_linear_<tag>.txt
T=${1:-a}
mkdir -p gpurun_out
for A in 0 1 2 4 16 18 3 5 6 7 23; do
  echo "== GCL_ABLATE=$A" >> gpurun_out/ablate_linear_$T.txt
  GCL_ABLATE=$A python3 tools/kbench.py --only linear --iters 20 2>/dev/null | grep -E "linear_fwd" >> gpurun_out/ablate_linear_$T.txt
done
cat gpurun_out/ablate_linear_$T.txt
