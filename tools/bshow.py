"""One line per bench.py output file: value, ms/step, the roofline kernel's launch time and fraction.

    python tools/bshow.py gpurun_out/a.json gpurun_out/b.json ...

Development helper for A/B runs (not part of the product path, the tests or bench.py)."""
import json
import sys


def main():
    for path in sys.argv[1:]:
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
        except Exception as e:  # an empty or truncated file is reported, not fatal
            print(f"{path}: unreadable ({e})")
            continue
        r = d.get("roofline") or {}
        sub = r.get("gcn_layer_one_kernel") or r.get("gat_bwd") or {}
        print(f"{path}: {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  "
              f"roofline {r.get('avg_launch_us', float('nan')):.1f} us frac {r.get('frac', float('nan')):.3f} "
              f"traffic {r.get('traffic')}  sub {sub.get('avg_launch_us', float('nan')):.1f} us frac {sub.get('frac', float('nan')):.3f}  "
              f"sustained {d.get('config', {}).get('sustained')}")


if __name__ == "__main__":
    main()
