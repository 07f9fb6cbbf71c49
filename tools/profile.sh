#!/bin/bash
# Kernel-trace + stats profile of the default bench workload.  Run on the GPU box:
#   gpurun -- 'bash tools/profile.sh r01'
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv you want judged into profiles/.
set -u
TAG=${1:-r01}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- \
    python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "rc=$?"; tail -3 "$OUT/bench.log"
find "$OUT" -name "*kernel_stats.csv" | head -3
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && column -s, -t < "$F" | cut -c1-200 | head -20
