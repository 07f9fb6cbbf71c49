#!/bin/bash
# per-kernel stats of ONE library on several configurations with extra enc_lab flags, same box, same call
#   gpurun -- 'bash tools/lab_prof_flags.sh "<cfgs>" <flags...>'
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  OUT=/tmp/lpf_$c; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- "$R/build/enc_lab" "$R/frlw-evd_amd/csrc/libfrlw_evd.so" --cfg "$c" --reps 20 "$@" > "$OUT/run.log" 2>&1
  echo "== $c $*"; grep -E "^$c " "$OUT/run.log" | head -1 | cut -c1-110
  python3 "$R/tools/kstats.py" $(find "$OUT" -name "*kernel_stats.csv" | head -1) | grep -v "selftest\|rocclr\|leaky_fill" | head -8
done
