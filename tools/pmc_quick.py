#!/usr/bin/env python3
"""Per-launch averages of the counters tools/pmc_quick.sh collected, for the scan kernel."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
pat = sys.argv[2] if len(sys.argv) > 2 else "scan_ranges"
pmc = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in acc:
        pmc[k] = acc[k] / len(disp[k])
json.dump(pmc, open(os.path.join(src, "pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(pmc, indent=1, sort_keys=True))
