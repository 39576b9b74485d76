#!/usr/bin/env python3
"""Sums the rocprofv3 --pmc CSVs written by tools/pmc_round.sh (or pmc_split.sh) per kernel and prints derived ratios.
    python tools/pmc_summary.py [gpurun_out/pmc] [nsteps]"""
import csv, collections, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for fn in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k].add((fn, r.get("Dispatch_Id")))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "vic" not in k:
        continue
    print(k)
    for c, x in sorted(v.items()):
        print("   %-28s %.4g   (%.4g per step)" % (c, x, x / nsteps))
    if v.get("SQ_ACTIVE_INST_VALU"):
        print("   lanes active per VALU inst        %.1f / 64" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]))
        print("   VALU-active share of wave cycles  %.3f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]))
        print("   wave parked (WAIT_ANY) share      %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
        print("   VALU insts per wave               %.3g" % (v["SQ_INSTS_VALU"] / v["SQ_WAVES"]))
        print("   quad-cycles per VALU inst issued  %.2f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"]))
