#!/bin/bash
# Level-0 interpolation kernel by phase (builds with -DMI_INTERP_STOP=<phase>, see setup_kernels.hip): kernel-trace of one
# setup at n^3, first call of interp_group_k<8, 32, 256>.     bash profiles/debug/interp_phase_times.sh [n]
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
n=${1:-256}
export TMPDIR=/tmp
cd /tmp
for v in 1 3 4 5 0; do
  if [ $v = 0 ]; then L=$REPO/hypre-mini-app_amd/libmi_hypre.so; else L=$REPO/hypre-mini-app_amd/build/libmi_hypre_stop$v.so; fi
  rm -rf /tmp/ipt_$v
  MI_HYPRE_LIB=$L MI_HYPRE_DENSE_TAIL_ROWS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ipt_$v -- \
    python3 $REPO/profiles/debug/setup_only.py $n > /tmp/ipt_$v.log 2>&1
  python3 - /tmp/ipt_$v $v <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
if not f:
    print("stop after phase", sys.argv[2], ": no trace"); sys.exit(0)
calls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f[-1]))
         if "interp_group_k<8, 32, 256>" in r["Kernel_Name"]]
calls.sort()
print("stop after phase %s: interp_group_k<8,32,256> calls (ms): %s" % (sys.argv[2], " ".join("%.1f" % (d / 1e6) for _, d in calls[:4])))
PY
done
