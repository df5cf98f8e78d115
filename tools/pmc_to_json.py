#!/usr/bin/env python
"""profiles/pmc_latest.json from a counter summary written by tools/pmc_summary.py.
Usage: tools/pmc_to_json.py profiles/<tag>_pmc_summary.txt"""
import json
import os
import re
import sys

KERNELS = ("sl_plan_kernel", "st_dp_kernel", "sl_dp_kernel", "sl_edge_cost_kernel")
src = sys.argv[1]
k = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
    name = m.group(1).split("<")[0] if m else None   # template instances count under the kernel's name
    if name in KERNELS:
        k.setdefault(name, {})[m.group(2)] = float(m.group(3))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_latest.json")
json.dump({"source": src, "kernels": k}, open(out, "w"), indent=1)
print({a: sorted(b) for a, b in k.items()})
