"""Development aid: mean of every counter per kernel over the dispatches of a rocprofv3 --pmc
output directory.  Usage: tools/pmc_kernel_rows.py <dir> [kernel substring]"""
import collections
import csv
import glob
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print("   %-28s n=%d mean=%.4g" % (name, len(v), sum(v) / len(v)))
