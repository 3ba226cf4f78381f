#!/usr/bin/env python3
"""print Calls / AverageNs of the kernels whose name contains one of the given substrings, from a rocprofv3 --stats dir"""
import csv, glob, sys
for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            print("   %-40s calls %4s avg %9.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
