#!/usr/bin/env python3
"""Where the time between kernels goes ("ms_per_step - sum of kernel time").
Reads a rocprofv3 --kernel-trace CSV of `bench.py` and splits it into solves: a solve of the GMRES loop starts
with the gather_k launches of enter_level_order and ends with the scatter_k of leave_level_order (krylov.cpp).
Per solve: span (first kernel start .. last kernel end), busy time (sum of kernel durations), idle time, and the
idle time grouped by the kernel that follows the gap.
   usage: gap_analysis.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0][:60]))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("mi::k::scatter_k")]
solves = []
for e in ends:
    # walk back to the pair of gather_k launches that opened this solve
    b = e
    while b > 0 and not (rows[b][2].startswith("mi::k::gather_k") and rows[b - 1][2].startswith("mi::k::gather_k")):
        b -= 1
    if b > 0 and (not solves or b - 1 > solves[-1][1]):
        solves.append((b - 1, e))
print(f"{len(rows)} kernels in the trace, {len(solves)} solves found")
for si, (b, e) in enumerate(solves):
    seg = rows[b:e + 1]
    busy = sum(x[1] - x[0] for x in seg)
    span = seg[-1][1] - seg[0][0]
    gaps = defaultdict(lambda: [0, 0])
    small = defaultdict(lambda: [0, 0])
    prev_end = seg[0][1]
    for s, en, name in seg[1:]:
        g = max(0, s - prev_end)
        gaps[name][0] += g
        gaps[name][1] += 1
        prev_end = max(prev_end, en)
    for s, en, name in seg:
        if en - s < 25000:
            small[name][0] += en - s
            small[name][1] += 1
    print(f"\nsolve {si}: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, "
          f"idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.2f} % of the span)")
    print("  idle time by the kernel that follows the gap (top 6):")
    for name, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"     {name:44s} gaps {c:5d}  total {t/1e6:7.3f} ms  mean {t/max(c,1)/1e3:7.1f} us")
    tot_small = sum(v[0] for v in small.values())
    n_small = sum(v[1] for v in small.values())
    print(f"  launches shorter than 25 us (coarse levels, scalars): {n_small} launches, {tot_small/1e6:.2f} ms "
          f"({100*tot_small/span:.2f} % of the span)")
