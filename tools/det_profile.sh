#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/det_layers; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/tools/det_layers.py > $OUT/run.log 2>&1
cd $R; python3 tools/det_layers.py parse $OUT/t_kernel_trace.csv
