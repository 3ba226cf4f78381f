#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output (one dir per pass) into per-kernel averages per dispatch."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void tq::", "").replace("tq::", "")
            c = agg[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
for k in sorted(agg):
    if not (k.startswith("k_") or "persp" in k):
        continue
    print(k)
    for name in sorted(agg[k]):
        tot, n = agg[k][name]
        print(f"    {name:34s} {tot / n:18.1f}   (avg over {n} dispatches)")
