#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats CSV: python tools/kstats.py <dir or csv> [launches-per-unit divisor]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{p}: kernel time {tot / div / 1e6:.3f} ms per unit")
for r in rows[:30]:
    print(f"{r['Name'][:88]:88s} {int(r['Calls']) / div:8.1f} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / div / 1e6:8.3f} ms")
