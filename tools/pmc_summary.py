"""Summarise the per-pass rocprofv3 --pmc CSVs written by tools/pmc.sh: per kernel (short name + grid):
mean of every counter, mean dispatch duration of that pass, VGPR / LDS from the dispatch records.
    python tools/pmc_summary.py gpurun_out/pmc_<tag> [name-substring ...]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read
(MI355X_MICROARCH.md, HBM) - the `hbm_MB` line applies the x2."""
import collections
import csv
import glob
import re
import sys


def short(n):
    m = re.search(r"(\w+<[^(]*>|\w+)\(", n)
    return m.group(1) if m else n[:60]


def main():
    out, filts = sys.argv[1], sys.argv[2:]
    res, meta, dur = collections.defaultdict(dict), {}, collections.defaultdict(list)
    for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
        acc, d = collections.defaultdict(list), collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if filts and not any(x in r["Kernel_Name"] for x in filts):
                continue
            key = (short(r["Kernel_Name"]), r["Grid_Size"])
            acc[key + (r["Counter_Name"],)].append(float(r["Counter_Value"]))
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[key] = f"vgpr {r['VGPR_Count']} agpr {r['Accum_VGPR_Count']} sgpr {r['SGPR_Count']} lds {r['LDS_Block_Size']} wg {r['Workgroup_Size']}"
        for k, v in acc.items():
            res[k[:2]][k[2]] = sum(v) / len(v)
        for k, v in d.items():
            dur[k].append(sum(v) / len(v))
    for k, c in sorted(res.items()):
        print(f"{k[0]}  grid={k[1]}  [{meta[k]}]  dur_us/pass={[round(x, 1) for x in dur[k]]}")
        for n, v in sorted(c.items()):
            print(f"   {n:28s} {v:.6g}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            print(f"   hbm_MB (2*FETCH + WRITE)      {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6:.1f}"
                  f"  (read {2 * c['FETCH_SIZE'] * 1024 / 1e6:.1f} + write {c['WRITE_SIZE'] * 1024 / 1e6:.1f})")
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            print("   shares of wave cycles: wait_any %.2f wait_inst %.2f active %.2f | lds conflict/active %.3f" % (
                c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w,
                c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))


if __name__ == "__main__":
    main()
