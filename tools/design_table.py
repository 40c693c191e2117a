#!/usr/bin/env python3
"""tools/design_table.py <round tag> -- the "Measured" table of DESIGN.md from profiles/<tag>_*_summary.json."""
import json
import sys
from pathlib import Path

ORDER = ["products_d128", "arxiv_d128", "products_d64", "products_d256", "products_d512", "products_d100", "products_d130",
         "products_d200", "arxiv_d1433", "arxiv_d3703", "lowdeg_d128", "lowdeg_d64", "hbm16m_d128", "products_x6_d128",
         "h2gcn_like_d128", "h2gcn_like_d64", "products_tail_d128"]
NOTE = {"products_d128": "**configs[3], headline**", "arxiv_d128": "configs[2]", "products_d256": "scratch copy incl.",
        "products_d512": "scratch copy incl.", "products_d100": "400-B rows: 4 lines/edge, floor 0.78", "products_d130": "d % 4 ≠ 0",
        "products_d200": "scratch copy incl.", "arxiv_d1433": "raw Cora width, in place", "arxiv_d3703": "citeseer width",
        "lowdeg_d128": "N = 8M, mean degree 4", "hbm16m_d128": "N = 16M, mean degree 7.5, X = 8.2 GB",
        "products_x6_d128": "N = 16M, mean degree 50, X = 8.2 GB (DRAM-resident)", "h2gcn_like_d128": "A1 mean 8 next to A2 mean 100",
        "products_tail_d128": "degrees from 1, 43 % of rows < 16"}


def main(tag):
    root = Path(__file__).resolve().parents[1] / "profiles"
    print("| workload | walk, slice | fwd ms (avg / median) | fwd frac (avg / median) | traffic / B_alg | adjoint frac |")
    print("|---|---|---|---|---|---|")
    for name in ORDER:
        f = root / f"{tag}_{name}_summary.json"
        if not f.exists():
            continue
        d = json.loads(f.read_text())
        sch = d.get("schedule") or {}
        walk = {"wave per segment": "wave", "lane group per segment (short rows)": "in-tile lane groups"}.get(sch.get("segment_walk"), sch.get("segment_walk", "?"))
        if walk and walk.startswith("lane group per segment (binned"):
            walk = "list-driven"
        ms = d["forward_ns_per_step_from_rocprof"] / 1e6
        med = f"{ms:.2f}"
        if "frac_from_rocprof_median" in d:      # per step, scratch copy included: back out of the fraction
            med += f" / {d['algorithmic_bytes_per_launch'] / (d['frac_from_rocprof_median'] * 8e12) * 1e3:.2f}"
        fr = f"{d['frac_from_rocprof_avg']:.3f}" + (f" / {d['frac_from_rocprof_median']:.3f}" if "frac_from_rocprof_median" in d else "")
        adj = d.get("adjoint", {}).get("frac_from_rocprof_avg")
        label = name.replace("_d", ", d = ") + (f" ({NOTE[name]})" if name in NOTE else "")
        print(f"| {label} | {walk}, {sch.get('slice_cols', '?')} | {med} | {fr} | {d.get('traffic_over_algorithmic', float('nan')):.2f} | "
              f"{'' if adj is None else f'{adj:.3f}'} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
