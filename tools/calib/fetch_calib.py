#!/usr/bin/env python3
"""FETCH_SIZE per byte actually read, from the PMC pass of tools/calib/fetch_calib.hip:  python tools/calib/fetch_calib.py <dir> <bytes_per_launch>
Writes profiles/fetch_calibration.json (read by tools/round_summary.py)."""
import csv, glob, json, os, sys, collections
root, nbytes = sys.argv[1], float(sys.argv[2])
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for fn in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
out = {}
for k in tot:
    counted = tot[k] / n[k] * 1024.0            # KiB -> bytes per launch
    out[k] = {"fetch_size_bytes_per_launch": counted, "bytes_read_per_launch": nbytes, "true_bytes_per_counted_byte": nbytes / counted}
    print("%-16s counted %.4g B, read %.4g B: factor %.3f" % (k, counted, nbytes, nbytes / counted))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "fetch_calibration.json")
json.dump({"kernels": out, "method": "tools/calib/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE on MI355X; 3.2 GB read once per launch"}, open(path, "w"), indent=1)
