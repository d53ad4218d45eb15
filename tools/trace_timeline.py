#!/usr/bin/env python3
"""Print the launch timeline (start offset, duration, stream/queue) of the last N dispatches of the step kernel out of a
rocprofv3 --kernel-trace csv.  usage: trace_timeline.py <dir> <kernel substring> [N=75]"""
import csv
import glob
import sys

d, kern = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 75
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r.get('Queue_Id', '?'):>3} start {s / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  end {e / 1e3:9.1f}")
