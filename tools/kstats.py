#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: avg us, calls, share."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|frlw::|void ", "", r["Name"])
    name = name.split("(")[0][:60]
    print(f"{float(r['AverageNs'])/1e3:10.1f} us  x{int(r['Calls']):4d}  {float(r['Percentage']):6.2f}%  {name}")
