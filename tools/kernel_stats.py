#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), "%10.2f ms" % (float(r["TotalDurationNs"]) / 1e6),
          "avg %9.1f us" % (float(r["AverageNs"]) / 1e3), "%6.2f%%" % float(r["Percentage"]))
