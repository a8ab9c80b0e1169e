#!/usr/bin/env python3
"""Fold a rocprofv3 counter_collection.csv per (kernel, counter): dispatch count, sum and the largest single dispatch.
usage: python tools/pmc_fold.py <counter_collection.csv> [kernel substring]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").strip()
    if flt in k:
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-44s %-22s dispatches %4d  sum %.6g  max %.6g" % (k, c, len(v), sum(v), max(v)))
