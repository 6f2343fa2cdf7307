#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc CSVs written by tools/profile_pmc.sh: per-counter mean over the render-kernel dispatches."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
kernel = sys.argv[2] if len(sys.argv) > 2 else "renderKernel<false"      # every variant of the plain (uninstrumented) render kernel
acc = defaultdict(list)
for path in glob.glob(os.path.join(out_dir, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
summary["_dispatches"] = {k: len(v) for k, v in acc.items()}
print(json.dumps(summary, indent=1))
