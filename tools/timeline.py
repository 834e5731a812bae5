#!/usr/bin/env python3
"""Timeline of the last steps of a run traced with
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 bench.py ...
Prints, for a window at the end of the run, every kernel and copy with start / duration in microseconds relative to the window
start, and the busy fractions of the compute queue and of each copy direction.  Usage: tools/timeline.py DIR [window_ms]"""
import csv
import glob
import os
import sys


def rows(pattern):
    out = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    d = sys.argv[1]
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 6e6
    ev = []
    for r in rows(os.path.join(d, "**", "*kernel_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-48:]))
    for r in rows(os.path.join(d, "**", "*memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?"))))
    ev.sort()
    if not ev:
        print("no events under", d)
        return
    end = ev[-1][1]
    t0 = end - win
    sel = [e for e in ev if e[0] >= t0]
    busy = {}
    for s, e, k, name in sel:
        key = k if k == "K" else name.split()[0]
        busy[key] = busy.get(key, 0) + (e - s)
        print("%10.1f %9.1f %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, k, name))
    span = sel[-1][1] - sel[0][0]
    print("window %.3f ms:" % (span / 1e6), ", ".join("%s busy %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(busy.items())))


if __name__ == "__main__":
    main()
