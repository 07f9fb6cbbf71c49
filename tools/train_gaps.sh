#!/bin/bash
# Where the GPU idles inside the train step: rocprofv3 kernel trace of Trainer.train_step at batch $B (default 64), then per
# step busy / idle time and the largest gaps with the kernels either side.   gpurun -- 'bash tools/train_gaps.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_train_gaps; KEEP=$R/gpurun_out/train_gaps; rm -rf $OUT; mkdir -p $OUT $KEEP
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/tools/train_gaps.py run > $OUT/run.log 2>&1
tail -3 $OUT/run.log
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/train_gaps.py report "$F" | tee $KEEP/gaps.txt
python3 $R/tools/train_gaps.py sequence "$F" $KEEP/sequence.txt
