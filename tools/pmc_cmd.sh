#!/bin/bash
# PMC passes over an arbitrary python command (separate runs, kernel-trace only, as the pool requires).
#   gpurun -- 'bash tools/pmc_cmd.sh <tag> tools/time_taf.py --only mpx --no-general --steps 3'
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/frlw_pmc_$TAG
KEEP=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$KEEP"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 "$SCRIPT" ${ARGS} > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
ARGS="$*"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
if [ "${TRAFFIC:-1}" = "1" ]; then
run fetch FETCH_SIZE
run write WRITE_SIZE
fi
python3 "$R/tools/pmc_summary.py" "$OUT" | tee "$KEEP/summary.txt"
