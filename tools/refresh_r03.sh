#!/bin/bash
# All profile passes of round 3 in one gpurun call; summaries land under gpurun_out/refresh_r03/ (then
# tools/collect_r03.py copies them into profiles/).      gpurun --timeout 1800 -- 'bash tools/refresh_r03.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=$R/gpurun_out/refresh_r03; rm -rf $K; mkdir -p $K
LIB=$R/frlw-evd_amd/csrc/libfrlw_evd.so
cd /tmp && export TMPDIR=/tmp
stats() { # tag program args...   (the program itself behind `--`: no shell, no env wrapper)
  local tag=$1; shift; local O=/tmp/frlw_r03_$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- "$@" > $O/run.log 2>&1; echo "$tag rc=$?"
  cp "$(find $O -name '*kernel_stats.csv' | head -1)" $K/${tag}_kernel_stats.csv
}
stats bench python3 $R/bench.py --steps 20 --warmup 3 --no-also --no-detector --no-train --no-cpu-baseline
stats hot python3 $R/bench.py --steps 10 --warmup 2 --hotspot --no-also --no-detector --no-train --no-cpu-baseline
for c in gen1 gen1x64 ev1 evb64; do stats lab_$c $R/build/enc_lab $LIB --cfg $c --reps 20; done
stats det python3 $R/tools/time_detector.py
GRAPH=1 B=64 stats train python3 $R/tools/train_gaps.py run   # Trainer(graph=True): 3 eager warm-up steps, the capture, 8 replays
# PMC passes (separate runs, kernel-trace only) of the encoder workloads through build/enc_lab (same kernels, same shapes,
# no Python in the profiled process)
pmc() { # cfg tag counters...
  local cfg=$1 tag=$2; shift 2; local O=/tmp/frlw_r03_pmcsum_$cfg/$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o p -- $R/build/enc_lab $LIB --cfg $cfg --reps 3 > $O/run.log 2>&1; echo "pmc $cfg $tag rc=$?"
}
# the opt-in tile walk (kf_taf_tile) on the headline shape: time and traffic, for DESIGN.md section 3.4
stats lab_mpx_tilewalk $R/build/enc_lab $LIB --cfg mpx --reps 20 --tile-walk
EXTRA=--tile-walk
pmcx() { local cfg=$1 tag=$2; shift 2; local O=/tmp/frlw_r03_pmcsum_tw/$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o p -- $R/build/enc_lab $LIB --cfg $cfg --reps 3 --tile-walk > $O/run.log 2>&1; echo "pmc tw $tag rc=$?"; }
rm -rf /tmp/frlw_r03_pmcsum_tw; pmcx mpx fetch FETCH_SIZE; pmcx mpx write WRITE_SIZE
python3 $R/tools/pmc_summary.py /tmp/frlw_r03_pmcsum_tw > $K/mpx_tilewalk_pmc_summary.txt
for W in mpx mpx_hot gen1 gen1x64 ev1 evb64; do
  rm -rf /tmp/frlw_r03_pmcsum_$W
  if [ $W = mpx ] || [ $W = mpx_hot ]; then
    pmc $W sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
    pmc $W sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  fi
  pmc $W fetch FETCH_SIZE
  pmc $W write WRITE_SIZE
  python3 $R/tools/pmc_summary.py /tmp/frlw_r03_pmcsum_$W > $K/${W}_pmc_summary.txt
done
cd $R && python3 bench.py > $K/bench.json 2> $K/bench.err; echo "bench rc=$?"
python3 bench.py --hotspot --no-detector --no-train --no-also > $K/bench_hotspot.json 2>> $K/bench.err; echo "bench hot rc=$?"
ls -la $K
