#!/bin/bash
# PMC passes for one bench.py configuration (run on the GPU box via gpurun).
# Counters are collected in their own rocprofv3 runs, one --pmc set per run,
# never combined with tracing options.  usage: tools/pmc_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOTDIR/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
 "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
i=0
NP=${PMC_PASSES:-99}
for P in "${PASSES[@]}"; do
  [ $i -ge $NP ] && break
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOTDIR/bench.py" --no-newton --cpu-sample 0 --steps 3 --warmup 1 "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
  echo "pass $i done"
done
python3 "$ROOTDIR/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
