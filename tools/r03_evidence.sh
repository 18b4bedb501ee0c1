#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root (bash tools/r03_evidence.sh <part>); everything lands in
# gpurun_out/r03/ and is copied into profiles/ afterwards.  Parts: bench | stats | pmc | micro
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
case "$1" in
bench)
  for c in baseline attention wb2_512x256_19f_ar wb2_512x256_sparse_gat wb2_512x256_19f_ar_v2; do
    extra=""; [ $c != baseline ] && extra="--no-cpu-baseline"
    timeout -k 10 400 python3 bench.py --config $c $extra > $O/bench_$c.json 2> $O/bench_$c.err || echo "bench $c failed"
  done ;;
stats)
  cd /tmp && export TMPDIR=/tmp
  for c in baseline attention wb2_512x256_19f_ar; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 $R/bench.py --config $c --steps 20 --no-cpu-baseline > $O/stats_$c.log 2>&1
    cp $O/stats_$c/*/*kernel_stats.csv $O/kernel_stats_$c.csv
  done ;;
pmc)
  PMC_SCRIPT=bench.py PMC_ITERS="" bash tools/pmc.sh r03_baseline _kernel --config baseline --eager --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_r03_baseline agg_halo gcn_halo agg_kernel gcn_fwd linear_x3_bwd > $O/pmc_baseline.txt
  python3 tools/pmc_roofline.py gpurun_out/pmc_r03_baseline baseline 64 agg_halo_loop_kernel gcn_halo_fwd_kernel
  PMC_SCRIPT=bench.py PMC_ITERS="" bash tools/pmc.sh r03_attention _kernel --config attention --eager --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_r03_attention gat_ > $O/pmc_attention.txt
  python3 tools/pmc_roofline.py gpurun_out/pmc_r03_attention attention 64 gat_halo_fwd_kernel gat_halo_bwd_dst_kernel gat_halo_bwd_src_kernel
  PMC_SCRIPT=bench.py PMC_ITERS="" bash tools/pmc.sh r03_wb2 _kernel --config wb2_512x256_19f_ar --eager --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_r03_wb2 agg_ > $O/pmc_wb2_512x256_19f_ar.txt
  python3 tools/pmc_roofline.py gpurun_out/pmc_r03_wb2 wb2_512x256_19f_ar 8 agg_halo_loop_kernel
  cp profiles/pmc_roofline_*.json $O/ ;;
micro)
  timeout -k 10 300 python3 tools/kbench.py > $O/kbench_baseline_b64.txt 2>&1
  timeout -k 10 300 python3 tools/kbench.py --only agg,gcn --ring 4 > $O/kbench_baseline_b64_ring4.txt 2>&1
  timeout -k 10 200 python3 tools/kbench.py --only gat > $O/kbench_attention_gat.txt 2>&1
  timeout -k 10 200 python3 tools/halo_check.py > $O/halo_check_mesh35_f64_b64.txt 2>&1
  timeout -k 10 200 python3 tools/halo_check.py --levels 4,6 --F 128 --B 8 > $O/halo_check_mesh46_f128_b8.txt 2>&1
  timeout -k 10 100 ./tools/probes/copy_probe > $O/copy_probe.txt 2>&1
  timeout -k 10 100 ./tools/probes/tile_copy_probe > $O/tile_copy_probe.txt 2>&1
  GCL_LIB=graphcast-lite_amd/libgcl_hip_stamps.so timeout -k 10 120 python3 tools/stamps_agg.py > $O/stamps_agg_loop.txt 2>&1 ;;
esac
echo "part $1 done"
