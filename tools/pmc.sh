#!/bin/bash
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of the kernels matching a name filter,
# driven by tools/kbench.py.  Run on the GPU box from the repo root:
#   bash tools/pmc.sh <tag> <kernel-name-substring> [kbench args...]
# PMC_SCRIPT=tools/halo_check.py profiles another driver script instead of kbench; PMC_ITERS="" drops the "--iters 3"
# that kbench / halo_check take (e.g. PMC_SCRIPT=bench.py PMC_ITERS="" bash tools/pmc.sh b bench_kernel --eager --steps 3 ...).
# Writes gpurun_out/pmc_<tag>/summary.txt (per kernel + grid: mean of every counter, and the kernel-trace
# duration / VGPR / LDS columns).
TAG=$1; FILT=$2; shift 2
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$T -- python3 $R/${PMC_SCRIPT:-tools/kbench.py} ${PMC_ITERS---iters 3} "$@" > $OUT/$T.log 2>&1 || echo "pass $T failed" >> $OUT/summary.txt
done
python3 - "$OUT" "$FILT" <<'PY' >> $OUT/summary.txt
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            if filt in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
                acc[key + (r["Counter_Name"],)].append(float(r["Counter_Value"]))
                meta[key] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"))
        for k, v in sorted(acc.items()):
            print(f"{k[0]:60s} grid={k[1]:>9s} {k[2]:26s} n={len(v):3d} mean={sum(v)/len(v):.6g}  regs/lds={meta[k[:2]]}")
PY
cat $OUT/summary.txt
