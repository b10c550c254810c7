#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh.
Usage: tools/pmc_summary.py <outdir> [kernel-substring]"""
import collections
import csv
import glob
import sys


def main():
    out = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "scan_tile"
    for f in sorted(glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k in sorted(acc):
            print("%-24s %16.0f per launch (%d launches)" % (k, acc[k] / len(disp[k]), len(disp[k])))


if __name__ == "__main__":
    main()
