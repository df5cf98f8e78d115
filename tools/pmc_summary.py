#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch for
each (kernel, counter).  Usage: tools/pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    print("%-40s %-24s n=%-3d mean=%.6g" % (k[:40], c, len(v), sum(v) / len(v)))
