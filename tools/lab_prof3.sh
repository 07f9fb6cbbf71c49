#!/bin/bash
# per-kernel stats of several library builds on one configuration, same box, same call
#   gpurun -- 'bash tools/lab_prof3.sh <cfg> lib1.so lib2.so ...'
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  OUT=/tmp/lp3_$(basename $L .so)_$CFG; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- "$R/build/enc_lab" "$R/$L" --cfg "$CFG" --reps 20 > "$OUT/run.log" 2>&1
  echo "== $L $CFG"; grep -E "^$CFG " "$OUT/run.log" | head -1
  python3 "$R/tools/kstats.py" $(find "$OUT" -name "*kernel_stats.csv" | head -1) | grep -v "selftest\|rocclr" | head -9
done
