#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter.

usage: pmc_summary.py [--json OUT.json] CSV [CSV ...]
Prints a text table; with --json also writes {kernel: {counter: {"sum":, "dispatches":, "avg":}}}.
"""
import collections
import csv
import json
import sys

args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json = args[1]
    args = args[2:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for path in args:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("caster::", "").replace("void ", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print("   %-32s %18.0f  (%d dispatches, avg %.1f)" % (c, tot[k][c], n[k][c], tot[k][c] / n[k][c]))
    t = tot[k]
    if "SQ_THREAD_CYCLES_VALU" in t and t.get("SQ_ACTIVE_INST_VALU"):
        print("   -> VALU lane utilisation        %17.1f%%" % (100 * t["SQ_THREAD_CYCLES_VALU"] / (64 * t["SQ_ACTIVE_INST_VALU"])))
    if "TCC_HIT_sum" in t and (t["TCC_HIT_sum"] + t.get("TCC_MISS_sum", 0)) > 0:
        print("   -> L2 hit rate                  %17.1f%%" % (100 * t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])))
    if "SQ_WAVE_CYCLES" in t and t["SQ_WAVE_CYCLES"] > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in t:
                print("   -> %-28s %17.1f%% of wave cycles" % (c, 100 * t[c] / t["SQ_WAVE_CYCLES"]))
if out_json:
    with open(out_json, "w") as f:
        json.dump({k: {c: {"sum": tot[k][c], "dispatches": n[k][c], "avg": tot[k][c] / n[k][c]} for c in tot[k]}
                   for k in tot}, f, indent=1, sort_keys=True)
