#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"][:48]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(agg, key=lambda k: -sum(dur[k])):
    if sum(dur[k]) < 1e6:
        continue
    print(f"{k}  calls={len(dur[k])}  avg_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}")
    for c, v in sorted(agg[k].items()):
        print(f"    {c:32s} {sum(v) / len(v):.4g}")
