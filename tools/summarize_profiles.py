#!/usr/bin/env python3
"""Copies the judged summaries of a tools/refresh_profiles.sh run into profiles/:
   <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc.json (per-launch counter averages of the
   scan kernel and the HBM traffic derived from them as MI355X_MICROARCH.md prescribes:
   2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB).
Usage: tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, tag + "_bench.json"), "w").write(line + "\n")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:  # (a tag refreshed twice holds two traces: the newest one)
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, tag + "_kernel_stats.csv"))
    kernel = json.loads(line).get("roofline", {}).get("kernel", "scan_tile_kernel")
    pmc = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime):
        acc = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        for k in acc:
            pmc[k] = acc[k] / len(disp[k])
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        pmc["hbm_read_bytes_per_launch"] = 2 * pmc["FETCH_SIZE"] * 1024   # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        pmc["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
        pmc["traffic_bytes_per_launch"] = pmc["hbm_read_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"]
    # the acceptance figures of the scan's instruction diet, per corpus byte of one launch
    alg = json.loads(line).get("roofline", {}).get("alg_bytes_per_launch")
    if alg and "SQ_INSTS_VALU" in pmc:
        pmc["valu_instructions_per_byte"] = pmc["SQ_INSTS_VALU"] / alg
        pmc["salu_instructions_per_byte"] = pmc.get("SQ_INSTS_SALU", 0.0) / alg
        pmc["lds_instructions_per_byte"] = pmc.get("SQ_INSTS_LDS", 0.0) / alg
    if "SQ_THREAD_CYCLES_VALU" in pmc and pmc.get("SQ_ACTIVE_INST_VALU"):
        # lanes switched on, averaged over all vector instructions (the 5 of 17 per lookup that log a
        # finished token run with only the finishing lanes on)
        pmc["valu_lane_utilization"] = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
    if "SQ_INSTS_VMEM_WR" in pmc and alg:
        # one store instruction per lookup step of a wave: lane-steps offered per corpus byte
        pmc["lookup_steps_per_wave_launch"] = pmc["SQ_INSTS_VMEM_WR"]
        pmc["lane_steps_per_byte"] = 64.0 * pmc["SQ_INSTS_VMEM_WR"] / alg
    if alg and "traffic_bytes_per_launch" in pmc:
        pmc["traffic_over_algorithmic_bytes"] = pmc["traffic_bytes_per_launch"] / alg
    pmc["kernel"] = kernel
    json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
