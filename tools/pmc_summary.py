#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter."""
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("caster::", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print("   %-32s %18.0f  (%d dispatches, avg %.1f)" % (c, tot[k][c], n[k][c], tot[k][c] / n[k][c]))
