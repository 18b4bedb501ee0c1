#!/bin/bash
# HBM traffic and L2 hit rate of the aggregation kernels (separate --pmc passes, as the MI355X guide
# prescribes).  Run on the GPU box from the repo root:  bash tools/pmc_agg.sh [extra kbench args]
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$T -- python3 $R/tools/kbench.py --only agg --iters 3 "$@" > $R/gpurun_out/pmc/$T.log 2>&1 || echo "pass $T failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmc/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "agg_kernel" in r["Kernel_Name"]:
                acc[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for (c, g), v in sorted(acc.items()):
            print(f"{c:14s} grid={g:>10s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
