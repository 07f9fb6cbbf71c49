#!/bin/bash
# All profile passes of round 6 in one gpurun call; summaries land under gpurun_out/refresh_r06/ (then
# tools/collect_r06.py copies them into profiles/).      gpurun --timeout 2400 -- 'bash tools/refresh_r06.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=$R/gpurun_out/refresh_r06; rm -rf $K; mkdir -p $K
LIB=$R/frlw-evd_amd/csrc/libfrlw_evd.so
cd /tmp && export TMPDIR=/tmp
stats() { # tag program args...   (the program itself behind `--`: no shell, no env wrapper)
  local tag=$1; shift; local O=/tmp/frlw_r06_$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- "$@" > $O/run.log 2>&1; echo "$tag rc=$?"
  cp "$(find $O -name '*kernel_stats.csv' | head -1)" $K/${tag}_kernel_stats.csv
}
stats bench python3 $R/bench.py --steps 20 --warmup 3 --no-also --no-detector --no-train --no-cpu-baseline
stats hot python3 $R/bench.py --steps 10 --warmup 2 --hotspot --no-also --no-detector --no-train --no-cpu-baseline
for c in gen1 gen1x64 ev1 evb1 evb64 small; do stats lab_$c $R/build/enc_lab $LIB --cfg $c --reps 20; done
stats lab_gen1_nocm $R/build/enc_lab $LIB --cfg gen1 --reps 20 --no-cm # ... and the histogram partition on one GEN1 stream
stats lab_evb64_nocm $R/build/enc_lab $LIB --cfg evb64 --reps 20 --no-cm
stats sae python3 $R/tools/run_small_encoders.py sae 20
stats eci python3 $R/tools/run_small_encoders.py eci 20
stats det python3 $R/tools/time_detector.py                    # default arithmetic: float32 MFMA
GRAPH=1 B=64 stats train python3 $R/tools/train_gaps.py run   # Trainer(graph=True): the capture, then replays
(cd $R && bash tools/det_profile.sh > $K/det_layers.txt 2>&1); cd /tmp
(GRAPH=1 bash $R/tools/train_gaps.sh > /dev/null 2>&1; cp $R/gpurun_out/train_gaps/sequence.txt $K/train_sequence.txt; cp $R/gpurun_out/train_gaps/gaps.txt $K/train_gaps.txt); cd /tmp
# PMC passes (separate runs, kernel-trace only)
pmc() { # out-dir tag counters -- program args...
  local O=$1 tag=$2; shift 2; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $O/$tag; mkdir -p $O/$tag
  rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $O/$tag -o p -- "$@" > $O/$tag/run.log 2>&1; echo "pmc $O $tag rc=$?"
}
for W in mpx mpx_hot gen1 gen1x64 evb1 evb64; do
  O=/tmp/frlw_r06_pmcsum_$W; rm -rf $O
  if [ $W = mpx ] || [ $W = gen1 ]; then
    pmc $O sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $R/build/enc_lab $LIB --cfg $W --reps 3
    pmc $O sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -- $R/build/enc_lab $LIB --cfg $W --reps 3
  fi
  pmc $O fetch FETCH_SIZE -- $R/build/enc_lab $LIB --cfg $W --reps 3
  pmc $O write WRITE_SIZE -- $R/build/enc_lab $LIB --cfg $W --reps 3
  python3 $R/tools/pmc_summary.py $O > $K/${W}_pmc_summary.txt
done
for W in sae eci; do
  O=/tmp/frlw_r06_pmcsum_$W; rm -rf $O
  pmc $O fetch FETCH_SIZE -- python3 $R/tools/run_small_encoders.py $W 3
  pmc $O write WRITE_SIZE -- python3 $R/tools/run_small_encoders.py $W 3
  python3 $R/tools/pmc_summary.py $O > $K/${W}_pmc_summary.txt
done
# the NMS launches by candidate count, the window-table A/B (same box), the big 3x3 layer's MFMA-busy share and clock
python3 $R/tools/time_nms.py 2>&1 | grep -v amdgpu.ids > $K/nms_by_candidates.txt
python3 $R/tools/time_wtab.py 2>&1 | grep -v amdgpu.ids > $K/wtab_ab.txt
HOT=1 python3 $R/tools/time_wtab.py 2>&1 | grep -v amdgpu.ids > $K/wtab_ab_hot.txt
CFG=gen1x64 python3 $R/tools/time_wtab.py 2>&1 | grep -v amdgpu.ids > $K/wtab_ab_gen1x64.txt
(cd $R && bash tools/pmc_clock_bin.sh build/conv_lab 32 20 2 > $K/conv_big_clock.txt 2>&1); cd /tmp
cd $R && python3 bench.py > $K/bench.json 2> $K/bench.err; echo "bench rc=$?"; cp bench_detail.json $K/bench_detail.json
python3 bench.py --hotspot --no-detector --no-train --no-also --no-cpu-baseline > $K/bench_hotspot.json 2>> $K/bench.err; echo "bench hot rc=$?"
ls -la $K
