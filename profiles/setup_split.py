#!/usr/bin/env python3
"""Host-vs-device split of HYPRE_BoomerAMGSetup (the reference prints this phase as
"Preconditioner setup", /root/reference/src/HypreSystem.cpp:685-696,:731).
Reads a rocprofv3 trace directory of `bench.py` taken with --kernel-trace --memory-copy-trace --marker-trace and
reports, for the window of the library's two setup ranges (roctx: "mi_hypre BoomerAMGSetup (hierarchy)" and
"... (solve-phase format)"): the union of the intervals in which the device runs a kernel or a copy (busy), the
rest (idle = the device waits for the host: host loops, allocation, synchronisation), the idle time grouped by
the kernel that follows the gap, and the largest single gaps with their neighbours.
   usage: setup_split.py <dir with *_kernel_trace.csv ...> [min gap in ms to list, default 20]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
min_gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6


def find(pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    return nm.split("(")[0][:58]


ev = []  # (start, end, name)
with open(find("*kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
nk = len(ev)
mc = find("*memory_copy_trace.csv")
if mc:
    with open(mc) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
windows = []
marks = []  # every roctx range: (start, end, name)
mk = find("*marker_api_trace.csv")
if mk:
    with open(mk) as f:
        for r in csv.DictReader(f):
            marks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
            if any("BoomerAMGSetup" in str(v) for v in r.values()):
                windows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), [v for v in r.values() if "BoomerAMGSetup" in str(v)][0]))
if not windows:
    # no marker trace: from the first setup kernel to the first solve (two gather_k in a row open it, krylov.cpp)
    first = next(i for i, e in enumerate(ev) if e[2].startswith("mi::sk::"))
    last = next((i for i in range(first, len(ev) - 1) if ev[i][2].startswith("mi::k::gather_k") and ev[i + 1][2].startswith("mi::k::gather_k")), len(ev) - 1)
    windows = [(ev[first][0], ev[last][0], "first mi::sk:: kernel .. first solve (no marker trace)")]
windows.sort()
print(f"{nk} kernels, {len(ev) - nk} copies in the trace; {len(windows)} setup range(s)")
tot_span = tot_busy = 0
for w0, w1, name in windows:
    seg = [e for e in ev if e[1] > w0 and e[0] < w1]
    busy = 0
    gaps = defaultdict(lambda: [0, 0])
    big = []
    cur = w0
    prev = "(range start)"
    for s, e, nm in seg:
        s2 = max(s, w0)
        if s2 > cur:
            g = s2 - cur
            gaps[nm][0] += g
            gaps[nm][1] += 1
            if g >= min_gap:
                big.append((g, cur - w0, prev, nm))
        if min(e, w1) > cur:
            busy += min(e, w1) - max(cur, s2)
            cur = min(e, w1)
            prev = nm
    if w1 > cur:
        gaps["(range end)"][0] += w1 - cur
        gaps["(range end)"][1] += 1
        if w1 - cur >= min_gap:
            big.append((w1 - cur, cur - w0, prev, "(range end)"))
    span = w1 - w0
    tot_span += span
    tot_busy += busy
    kt = defaultdict(lambda: [0, 0])
    for s, e, nm in seg:
        kt[nm][0] += e - s
        kt[nm][1] += 1
    print(f"\n== {name}: span {span/1e9:.3f} s, device busy {busy/1e9:.3f} s, device idle (host) {(span-busy)/1e9:.3f} s, "
          f"{len(seg)} kernels/copies")
    print("  device time by kernel (top 12):")
    for nm, (t, c) in sorted(kt.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"     {nm:58s} {c:7d} x  {t/1e9:7.3f} s")
    print("  idle time by what follows the gap (top 12):")
    for nm, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"     {nm:58s} gaps {c:6d}  total {t/1e9:7.3f} s")
    print(f"  gaps of at least {min_gap/1e6:.0f} ms (offset into the range, length, after -> before):")
    for g, off, a, b in sorted(big, key=lambda x: x[1]):
        print(f"     +{off/1e9:7.3f} s  {g/1e6:8.1f} ms   {a}  ->  {b}")


def busy_in(w0, w1):
    b = 0
    cur = w0
    for s_, e_, _ in ev:
        if e_ <= w0:
            continue
        if s_ >= w1:
            break
        a = max(s_, cur)
        z = min(e_, w1)
        if z > a:
            b += z - a
            cur = z
    return b


if windows and marks:
    lo, hi = windows[0][0], windows[-1][1]
    sub = sorted(m for m in marks if m[0] >= lo and m[1] <= hi)
    print("\nevery roctx range inside the setup, in time order (indent = nesting): span, device busy, device idle = host")
    stack = []
    tot_idle = {}
    for s_, e_, nm in sub:
        while stack and s_ >= stack[-1]:
            stack.pop()
        b = busy_in(s_, e_)
        print(f"   {'  ' * len(stack)}{nm:{66 - 2 * len(stack)}s} {(e_-s_)/1e9:7.3f} s   busy {b/1e9:7.3f}   idle {(e_-s_-b)/1e9:7.3f}")
        stack.append(e_)
print(f"\nsetup ranges together: span {tot_span/1e9:.3f} s, device busy {tot_busy/1e9:.3f} s, device idle (host) {(tot_span-tot_busy)/1e9:.3f} s")
