#!/bin/bash
# All profile passes of round 2 in one gpurun call; summaries land under gpurun_out/refresh_r02/ (then
# tools/collect_r02.py copies them into profiles/).      gpurun --timeout 1500 -- 'bash tools/refresh_r02.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=$R/gpurun_out/refresh_r02; rm -rf $K; mkdir -p $K
cd /tmp && export TMPDIR=/tmp
stats() { # tag script args...
  local tag=$1; shift; local O=/tmp/frlw_r02_$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 "$@" > $O/run.log 2>&1; echo "$tag rc=$?"
  cp "$(find $O -name '*kernel_stats.csv' | head -1)" $K/${tag}_kernel_stats.csv
}
stats bench $R/bench.py --steps 20 --warmup 3 --no-also --no-detector --no-train --no-cpu-baseline
stats hot $R/bench.py --steps 10 --warmup 2 --hotspot --no-also --no-detector --no-train --no-cpu-baseline
stats gen1 $R/tools/time_taf.py --only gen1 --steps 20
stats gen1_b64 $R/tools/time_taf.py --only gen1_b64 --steps 10
stats det $R/tools/time_detector.py
B=64 stats train $R/tools/train_breakdown.py
# PMC passes (separate runs, kernel-trace only)
pmc() { # tag counters... (script args in PMC_ARGS)
  local tag=$1; shift; local O=/tmp/frlw_r02_pmc_$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o p -- python3 $R/tools/time_taf.py --only ${PMC_WORK:-mpx} --no-general --steps 3 > $O/run.log 2>&1; echo "pmc $tag rc=$?"
}
for W in mpx mpx_hot; do
  export PMC_WORK=$W
  pmc ${W}_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  pmc ${W}_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  pmc ${W}_fetch FETCH_SIZE
  pmc ${W}_write WRITE_SIZE
  mkdir -p /tmp/frlw_r02_pmcsum_$W; rm -rf /tmp/frlw_r02_pmcsum_$W/*
  for p in sq1 sq2 fetch write; do cp -r /tmp/frlw_r02_pmc_${W}_$p /tmp/frlw_r02_pmcsum_$W/; done
  python3 $R/tools/pmc_summary.py /tmp/frlw_r02_pmcsum_$W > $K/taf_${W}_pmc_summary.txt
done
cd $R && python3 bench.py > $K/bench.json 2> $K/bench.err; echo "bench rc=$?"
python3 bench.py --hotspot --no-detector --no-train --no-also > $K/bench_hotspot.json 2>> $K/bench.err; echo "bench hot rc=$?"
ls -la $K
