"""Collapse rocprofv3 counter_collection CSVs (one row per dispatch and counter) into per-kernel sums
so that only small files travel back from the GPU box.   python tools/pmc_aggregate.py gpurun_out/<tag>"""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
for path in glob.glob(os.path.join(src, "pmc_*", "*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    out = {k: {c: {"dispatches": v[0], "sum": v[1]} for c, v in d.items()} for k, d in agg.items()}
    with open(os.path.join(os.path.dirname(path), "per_kernel.json"), "w") as f:
        json.dump(out, f, indent=1)
    if os.path.getsize(path) > 8 << 20:
        os.remove(path)
for path in glob.glob(os.path.join(src, "*", "*_kernel_trace.csv")):
    if os.path.getsize(path) > 8 << 20:
        os.remove(path)
