#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command line.  gpurun -- 'bash tools/prof_cmd.sh <tag> tools/time_taf.py --only mpx'
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/frlw_prof_$TAG
KEEP=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$KEEP"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1
echo "rc=$?"; tail -5 "$OUT/run.log"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$KEEP/kernel_stats.csv" && python3 "$R/tools/kstats.py" "$F" | head -${TOPN:-14}
cp "$OUT/run.log" "$KEEP/"
