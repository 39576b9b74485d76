#!/usr/bin/env python3
"""Sums the rocprofv3 --pmc CSVs written by tools/pmc_split.sh per kernel and prints derived ratios."""
import csv, collections, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in sorted(glob.glob("gpurun_out/pmc_split*/p_counter_collection.csv")):
    with open(fn) as f:
        for r in csv.DictReader(f):
            tot[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    if "vic" not in k:
        continue
    print(k)
    for c, x in sorted(v.items()):
        print("   %-26s %.4g" % (c, x))
    if "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"]:
        print("   lanes active per VALU inst  %.1f / 64" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]))
        print("   VALU-active share of wave cycles %.2f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]))
        print("   VALU insts per wave %.3g" % (v["SQ_INSTS_VALU"] / v["SQ_WAVES"]))
