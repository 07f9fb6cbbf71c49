#!/bin/bash
# The train-step files of the round-6 refresh alone (after the block nodes of yolox/train_ops.py changed the step's launch
# sequence; no encoder / detector kernel changed): kernel stats, dispatch sequence, gaps, and the bench line + detail.
#      gpurun --timeout 1500 -- 'bash tools/refresh_r06_train.sh'        -> gpurun_out/refresh_r06_train/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=$R/gpurun_out/refresh_r06_train; rm -rf $K; mkdir -p $K
cd /tmp && export TMPDIR=/tmp
O=/tmp/frlw_r06_train; rm -rf $O; mkdir -p $O
GRAPH=1 B=64 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/train_gaps.py run > $O/run.log 2>&1; echo "train rc=$?"
cp "$(find $O -name '*kernel_stats.csv' | head -1)" $K/train_kernel_stats.csv
(GRAPH=1 bash $R/tools/train_gaps.sh > /dev/null 2>&1; cp $R/gpurun_out/train_gaps/sequence.txt $K/train_sequence.txt; cp $R/gpurun_out/train_gaps/gaps.txt $K/train_gaps.txt)
cd $R
for f in 0 1 0 1; do FRLW_TRAIN_FUSE=$f python3 tools/train_ab.py 2>&1 | tail -1 >> $K/train_fuse_ab.txt; done
for f in 0 1 0 1; do FRLW_TRAIN_STACK=$f python3 tools/train_ab.py 2>&1 | tail -1 >> $K/train_stack_ab.txt; done
python3 bench.py > $K/bench.json 2> $K/bench.err; echo "bench rc=$?"; cp bench_detail.json $K/bench_detail.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $K/smoke.log 2>&1; tail -1 $K/smoke.log
ls -la $K
