#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection CSVs per kernel and counter -> JSON on stdout."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if "mulut" not in k:
                    continue
                # one row per dispatch and counter; sum over dimensions of the same dispatch
                acc[k][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
    out = {}
    for k, cs in acc.items():
        short = k.split("(")[0].replace("void mulut::", "")
        out[short] = {}
        for c, vals in sorted(cs.items()):
            per = defaultdict(float)
            for d, v in vals:
                per[d] += v
            xs = list(per.values())
            out[short][c] = {"mean_per_dispatch": sum(xs) / len(xs), "dispatches": len(xs)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
