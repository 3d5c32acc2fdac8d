#!/usr/bin/env python3
"""Per-launch durations of a rocprofv3 --kernel-trace run: for every kernel whose name contains argv[2], the launches'
durations in microseconds (sorted), so that one kernel's large and small jobs can be told apart."""
import csv, glob, sys
from collections import defaultdict
rows = csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]))
d = defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, len(v), "launches; us:", " ".join(f"{x:.0f}" for x in v))
