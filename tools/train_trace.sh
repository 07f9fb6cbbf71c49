#!/bin/bash
# per-dispatch durations of the train-step kernels (one step's worth), grouped by grid size
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_train_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B=${B:-32} rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/tools/train_breakdown.py > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$1" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
pat=sys.argv[2]
sel=[r for r in rows if pat in r["Kernel_Name"]]
n=len(sel)//8  # 8 steps in train_breakdown
last=sel[-n:]
agg=collections.OrderedDict()
for r in last:
    key=(r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"])
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg.setdefault(key,[]).append(d)
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(f"grid={k} calls={len(v)} avg_us={sum(v)/len(v):8.1f} total_us={sum(v):9.1f}")
    tot+=sum(v)
print("total ms per step", tot/1e3, "dispatches", n)
PY
