#!/bin/bash
# SQ counters of the level-0 kernels (one --pmc pass, kernel trace only):  bash profiles/collect_sq.sh [n]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
n=${1:-256}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_sq" -- python3 "$REPO/bench.py" --n $n --steps 1 --warmup 0 --no-cpu > "$REPO/gpurun_out/prof_sq.log" 2>&1
python3 - "$REPO/gpurun_out/prof_sq" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (nm, int(r["Grid_Size"]))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[key] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
print("%-44s %10s %6s  %8s %8s %8s  %8s %8s %8s %8s" % ("kernel", "grid", "calls", "wait_any", "wait_ins", "active", "valu/wv", "lds/wv", "w_lds", "bankconf"))
for (nm, grid), c in rows:
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    waves = grid / 64.0 * cnt[(nm, grid)]
    print("%-44s %10d %6d  %7.1f%% %7.1f%% %7.1f%%  %8.0f %8.0f %7.1f%% %7.1f%%" % (nm[:44], grid, cnt[(nm, grid)], 100 * c["SQ_WAIT_ANY"] / wc,
          100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_INSTS_VALU"] / waves, c["SQ_INSTS_LDS"] / waves,
          100 * c["SQ_WAIT_INST_LDS"] / wc, 100 * c["SQ_LDS_BANK_CONFLICT"] / wc))
PY
