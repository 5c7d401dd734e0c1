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
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE"
)
i=0
NP=${PMC_PASSES:-99}
FIRST=${PMC_FIRST:-1}
for P in "${PASSES[@]}"; do
  [ $i -ge $NP ] && break
  if [ $((i+1)) -lt $FIRST ]; then i=$((i+1)); continue; fi
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOTDIR/bench.py" --no-newton --no-tet10 --no-off-lattice --cpu-sample 0 --steps 3 --warmup 1 "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
  echo "pass $i done"
done
python3 "$ROOTDIR/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
