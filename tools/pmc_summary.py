#!/usr/bin/env python3
"""Average every PMC counter per kernel name over all dispatches found under a pmc_* directory."""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|frlw::|void ", "", r["Kernel_Name"]).split("(")[0][:28]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    if name.startswith("__amd") or name.startswith("at::"):
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:24s} {sum(v)/len(v):16.0f}   (n={len(v)})")
