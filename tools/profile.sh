#!/bin/bash
# Kernel-trace + stats profile of the default bench workload.  Run on the GPU box:
#   gpurun -- 'bash tools/profile.sh r01'
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv you want judged into profiles/.
set -u
TAG=${1:-r01}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/frlw_prof_$TAG   # raw traces stay on the box; only the summaries go to gpurun_out (64 MiB cap)
KEEP=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$KEEP"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- \
    python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "rc=$?"; tail -3 "$OUT/bench.log"
find "$OUT" -name "*kernel_stats.csv" | head -3
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$KEEP/kernel_stats.csv"; cp "$OUT/bench.log" "$KEEP/"
[ -n "$F" ] && head -8 "$F" | cut -c1-200
