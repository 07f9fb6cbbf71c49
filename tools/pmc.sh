#!/bin/bash
# PMC passes over the default bench workload (separate runs, kernel-trace only, as the pool requires).
#   gpurun -- 'bash tools/pmc.sh r01'   -> gpurun_out/pmc_<tag>/pass*/
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/frlw_pmc_$TAG   # raw counter dumps stay on the box
KEEP=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$KEEP"
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- \
      python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline ${EXTRA:-} > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
python3 "$R/tools/pmc_summary.py" "$OUT" | tee "$KEEP/summary.txt"
