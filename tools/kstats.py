#!/usr/bin/env python3
"""Prints the fdc:: rows of a rocprofv3 --stats kernel_stats.csv found under DIR: calls, average us, name.  Usage: tools/kstats.py DIR"""
import csv
import glob
import os
import sys

for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fdc::" in r["Name"]:
            print("%6s %10.1f  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0][-60:]))
