"""Summarises rocprofv3 --pmc csv output (one row per dispatch and counter) for the fdc:: kernels:
per kernel name, the mean counter value per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "fdc::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean/dispatch %.6g   (n=%d)" % (c, sum(v) / len(v), len(v)))
