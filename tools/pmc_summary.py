#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv per (kernel, counter): usage pmc_summary.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
acc = defaultdict(lambda: [0.0, 0])
for f in files:
	for row in csv.DictReader(open(f)):
		name = row["Kernel_Name"]
		if flt and flt not in name:
			continue
		key = (name[:70], row["Counter_Name"])
		acc[key][0] += float(row["Counter_Value"])
		acc[key][1] += 1
for (name, ctr), (v, n) in sorted(acc.items()):
	print("%-70s %-28s per-launch %.4g (n=%d)" % (name, ctr, v / n, n))
