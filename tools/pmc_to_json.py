#!/usr/bin/env python
"""profiles/pmc_latest.json (lattice workload) or profiles/pmc_pipeline_latest.json (whole
replan, with --pipeline) from a counter summary written by tools/pmc_summary.py: per-kernel
counter means per launch, with the batch size the passes ran at.
Usage: tools/pmc_to_json.py [--pipeline] profiles/<tag>_pmc_summary.txt [scenarios]"""
import json
import os
import re
import sys

args = [a for a in sys.argv[1:] if a != "--pipeline"]
pipeline = "--pipeline" in sys.argv
src = args[0]
scenarios = int(args[1]) if len(args) > 1 else 4096
KERNELS = (("st_dp_kernel", "plan_frontend_kernel", "sl_plan_narrow_kernel", "refline_tables_kernel",
            "st_regions_once_kernel", "plan_path_kernel", "plan_tail_kernel") if pipeline else
           ("sl_plan_kernel", "st_dp_kernel", "sl_dp_kernel", "sl_edge_cost_kernel"))
k = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
    name = m.group(1).split("<")[0] if m else None   # template instances count under the kernel's name
    if name in ("st_dp_r_kernel", "st_dp_plain_kernel"):
        name = "st_dp_kernel"   # the instances per interval bound and the five-wave build are the same kernel body
    if name in KERNELS:
        if m.group(1).startswith("sl_plan_kernel<"):
            targs = [t.strip() for t in m.group(1)[len("sl_plan_kernel<"):].rstrip(">").split(",")]
            if "true" in targs[2:]:
                continue   # the counting instance (third argument) and the second launch of the compact
                           # form (fourth) are not the product kernel's main launch
        k.setdefault(name, {})[m.group(2)] = float(m.group(3))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles",
                   "pmc_pipeline_latest.json" if pipeline else "pmc_latest.json")
json.dump({"source": src, "scenarios": scenarios, "kernels": k}, open(out, "w"), indent=1)
print({a: sorted(b) for a, b in k.items()})
