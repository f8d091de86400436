#!/usr/bin/env python3
"""Per (kernel family, grid) mean times of several kernel traces side by side (profiles/compare_gs.sh)."""
import csv, glob, re, sys
from collections import defaultdict
def short(name):
    m = re.search(r"namespace\)::([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0]
runs = []
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))[-1]
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("gs_"):
            continue
        key = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        agg[(key, k)][0] += 1
        agg[(key, k)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    runs.append(agg)
grids = sorted({k[0] for a in runs for k in a}, reverse=True)
tot = [0.0] * len(runs)
for g in grids:
    cells = []
    for i, a in enumerate(runs):
        items = [(k[1], v) for k, v in a.items() if k[0] == g]
        if items:
            name, (n, us) = items[0]
            cells.append(f"{name:18s} {us / n:8.1f} us x{n:4d}")
            tot[i] += us
        else:
            cells.append(" " * 34)
    print(f"grid {g:<11d} " + " | ".join(cells))
print("total GS ms: " + " | ".join(f"{t / 1e3:8.1f}" for t in tot))
