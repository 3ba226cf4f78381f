#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace CSV, for runs whose set-up launches the same
kernel many times (bench.py's buffer probe): the timed region of bench.py is the LAST `steps` launches of the stack
write before the two launches of `stack_verified`.
usage: trace_tail.py <dir with *kernel_trace.csv> <kernel name substring> <steps> [launches after the timed region = 2]"""
import csv, glob, statistics, sys

d, name, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
after = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur = [x[1] for x in rows]
print(f"kernel: {rows[-1][2].split('(')[0] if rows else name}")
print(f"dispatches in the trace: {len(dur)}; all: mean {statistics.mean(dur) / 1e3:.2f} us, min {min(dur) / 1e3:.2f}, max {max(dur) / 1e3:.2f}")
tail = dur[-(steps + after):-after] if after else dur[-steps:]
print(f"timed region (the last {steps} before the final {after}): mean {statistics.mean(tail) / 1e3:.2f} us, median {statistics.median(tail) / 1e3:.2f}, "
      f"min {min(tail) / 1e3:.2f}, max {max(tail) / 1e3:.2f}, first {tail[0] / 1e3:.2f}")
