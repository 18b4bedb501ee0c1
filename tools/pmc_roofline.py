"""Write profiles/pmc_roofline_<config>.json - HBM-side bytes per launch of the roofline kernels - from the PMC passes
of tools/pmc.sh (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE).

    python tools/pmc_roofline.py gpurun_out/pmc_<tag> <config> <batch> <kernel-substring> [<kernel-substring> ...]

bench.py reads the file for `roofline.traffic` when config and batch match."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, config, batch, tags = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    acc = {t: collections.defaultdict(list) for t in tags}
    for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            for t in tags:
                if t in r["Kernel_Name"]:
                    acc[t][(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    kernels, detail = {}, {}
    for t in tags:
        # the most frequent grid of that kernel = the steady-state launch of the step
        grids = collections.Counter(g for (_, g), v in acc[t].items() for _ in v)
        if not grids:
            continue
        grid = grids.most_common(1)[0][0]
        f = acc[t].get(("FETCH_SIZE", grid))
        w = acc[t].get(("WRITE_SIZE", grid))
        if not f or not w:
            continue
        fetch, write = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
        kernels[t] = 2 * fetch + write
        detail[t] = {"grid": grid, "read_bytes": 2 * fetch, "write_bytes": write, "launches": [len(f), len(w)]}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", f"pmc_roofline_{config}.json")
    json.dump({"config": config, "batch": batch, "kernels": kernels, "detail": detail,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 (gfx950), per launch"},
              open(path, "w"), indent=1)
    print(path, kernels)


if __name__ == "__main__":
    main()
