"""Per-kernel durations and the gaps in front of them from a rocprofv3 kernel trace (csv):
    python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_rows]
Groups by (short) kernel name: calls, mean duration, mean gap to the END of the previous kernel on the device."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "")
    if "k_persp_stream" in short:
        short = "k_persp_stream" + ("<SCAN>" if short.rstrip(">").endswith("true") else "")
    short = short[:60]
    if i >= skip and prev_end is not None:
        a = agg[short]
        a[0] += 1
        a[1] += (e - s) / 1e3
        gap = (s - prev_end) / 1e3
        a[2] += gap if gap < 200 else 0.0
    prev_end = e
print("%-62s %6s %10s %10s" % ("kernel", "calls", "dur us", "gap us"))
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %6d %10.2f %10.2f" % (k, n, d / n, g / n))
