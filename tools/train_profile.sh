#!/bin/bash
# rocprofv3 kernel stats of the train step (tools/train_breakdown.py); summary -> gpurun_out/train_prof/kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_train_prof; KEEP=$R/gpurun_out/train_prof; rm -rf $OUT; mkdir -p $OUT $KEEP
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/train_breakdown.py > $OUT/run.log 2>&1
tail -2 $OUT/run.log
F=$(find $OUT -name "*kernel_stats.csv" | head -1); cp "$F" $KEEP/kernel_stats.csv
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} total_ms={float(r["TotalDurationNs"])/1e6:8.2f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
