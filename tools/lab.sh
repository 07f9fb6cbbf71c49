#!/bin/bash
# One gpurun call: differential + timing run of build/enc_lab (new library vs the reference build), then per-kernel
# rocprofv3 stats of the new library on the configurations named in $PROF.
#   gpurun -- 'bash tools/lab.sh <tag> "<cfgs>" ["<cfgs to profile>"]'
set -u
TAG=$1; CFGS=$2; PROF=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
KEEP=$R/gpurun_out/lab_$TAG
mkdir -p "$KEEP"
NEW=$R/frlw-evd_amd/csrc/libfrlw_evd.so
BASE=${BASE:-$R/build/libfrlw_base.so}
"$R/build/enc_lab" "$NEW" "$BASE" --cfg "$CFGS" --reps ${REPS:-20} 2>&1 | tee "$KEEP/lab.txt"
cd /tmp && export TMPDIR=/tmp
for c in $PROF; do
  OUT=/tmp/frlw_lab_${TAG}_$c; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- "$R/build/enc_lab" "$NEW" --cfg "$c" --reps 10 > "$OUT/run.log" 2>&1
  F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  if [ -n "$F" ]; then cp "$F" "$KEEP/${c}_kernel_stats.csv"; echo "== $c"; python3 "$R/tools/kstats.py" "$F" | head -14; fi
done
# PMC passes (separate runs, kernel-trace only) of the new library on the configurations named in $PMC
for c in ${PMC:-}; do
  OUT=/tmp/frlw_labpmc_${TAG}_$c; rm -rf "$OUT"; mkdir -p "$OUT"
  run() { local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- "$R/build/enc_lab" "$NEW" --cfg "$c" --reps 3 > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"; }
  run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  if [ "${TRAFFIC:-0}" = "1" ]; then run fetch FETCH_SIZE; run write WRITE_SIZE; fi
  echo "== pmc $c"; python3 "$R/tools/pmc_summary.py" "$OUT" | tee "$KEEP/${c}_pmc.txt"
done
