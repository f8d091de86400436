#!/bin/bash
# Exploratory counter passes on the 256^3 run (one rocprofv3 --pmc run per counter group).
#   gpurun --timeout 1100 -- 'bash profiles/collect_pmc_explore.sh'
set -e
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
export TMPDIR=/tmp
cd /tmp
i=0
GROUPS_DEFAULT=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
                "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
                "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY")
if [ -n "$MI_PMC_GROUPS" ]; then IFS=';' read -ra GROUPS_SEL <<< "$MI_PMC_GROUPS"; else GROUPS_SEL=("${GROUPS_DEFAULT[@]}"); fi
for grp in "${GROUPS_SEL[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmcx_$i" -- \
    python3 "$REPO/bench.py" --grid 256 --steps 1 --warmup 0 --no-cpu > "$REPO/gpurun_out/pmcx_$i.log" 2>&1
  echo "group $i done: $grp"
done
find "$REPO/gpurun_out" -name "*.csv" -size +60M -delete
