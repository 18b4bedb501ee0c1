#!/bin/bash
# Round-2 evidence: bench lines + rocprofv3 kernel stats for the BASELINE.json configs (run on the GPU box from the
# repo root; copies what should be judged from gpurun_out/ into profiles/ afterwards).
#   bash tools/r02_evidence.sh <tag> [configs...]
TAG=$1; shift
CFGS=${@:-"baseline attention wb2_512x256_19f_ar wb2_512x256_sparse_gat"}
R=$PWD
O=$R/gpurun_out/ev_$TAG
mkdir -p $O
for C in $CFGS; do
  python3 bench.py --config $C --steps 20 --warmup 5 > $O/bench_$C.json 2> $O/bench_$C.err
  echo "bench $C rc=$?"; tail -c 600 $O/bench_$C.json
  (cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -- python3 $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_$C.log 2>&1)
  echo "prof $C rc=$?"
  f=$(ls $O/prof_$C/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$C.csv && head -12 $f | cut -c1-160
done
