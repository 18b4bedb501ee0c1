#!/bin/bash
# Round-2 PMC evidence: F=128 mesh aggregation of configs[3], the GAT kernels of configs[2]/[4],
# and the split-operand dense kernels.  Run on the GPU box from the repo root; writes gpurun_out/pmc_<tag>/.
bash tools/pmc.sh agg128 agg_kernel --config wb2_512x256_19f_ar --only agg > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_agg128 agg_kernel > gpurun_out/pmc_agg128/summary_by_kernel.txt
bash tools/pmc.sh gat gat_ --config attention --only gat > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_gat gat_ > gpurun_out/pmc_gat/summary_by_kernel.txt
bash tools/pmc.sh gat128 gat_ --config wb2_512x256_19f_ar --only gat > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_gat128 gat_ > gpurun_out/pmc_gat128/summary_by_kernel.txt
bash tools/pmc.sh x3 linear_x3 --only linear > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_x3 linear_x3 > gpurun_out/pmc_x3/summary_by_kernel.txt
python3 tools/kbench.py --config wb2_512x256_19f_ar --only agg,gat > gpurun_out/kb_wb2_agg_gat.txt 2>&1
python3 tools/kbench.py --config attention --only gat > gpurun_out/kb_attention_gat.txt 2>&1
tail -n 40 gpurun_out/pmc_agg128/summary_by_kernel.txt gpurun_out/pmc_gat/summary_by_kernel.txt gpurun_out/kb_wb2_agg_gat.txt gpurun_out/kb_attention_gat.txt
