#!/bin/bash
# Timing-only ablations of the dense forward kernel (GCL_ABLATE bits: 1 no stores, 2 no MFMA, 4 no HBM loads).
# bash tools/ablate.sh  -> gpurun_out/ablate_linear.txt
mkdir -p gpurun_out
for A in 0 1 2 4 3 5 6 7; do
  echo "== GCL_ABLATE=$A" >> gpurun_out/ablate_linear.txt
  GCL_ABLATE=$A python3 tools/kbench.py --only linear --iters 20 2>/dev/null | grep -E "linear_fwd" >> gpurun_out/ablate_linear.txt
done
cat gpurun_out/ablate_linear.txt
