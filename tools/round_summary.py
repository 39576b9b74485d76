#!/usr/bin/env python3
"""Turns the output of tools/profile_round.sh (gpurun_out/round/) into the committed round profile:

  profiles/rNN_<config>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/traffic_<config>.json           HBM bytes per step from the FETCH_SIZE / WRITE_SIZE passes, stamped with the digest of
                                           the device sources it was measured on (bench.py reads it and refuses a stale one);
  profiles/rNN_traffic_<config>.json       the same file, kept per round

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a streaming read, so reads are doubled (the guide calibrates this for 16-byte-per-lane streams only; the 8-byte
and scattered accesses of these kernels are uncalibrated, the figure is an upper-bound estimate for them).

    python tools/round_summary.py r02 cfg3 100000 8 [_compat]   # round tag, config, cells per GPU, steps launched (warmup + timed)
"""
import collections, csv, json, os, shutil, sys
tag, config, ncell, nsteps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
suffix = sys.argv[5] if len(sys.argv) > 5 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
src = os.path.join(ROOT, "gpurun_out", os.environ.get("ROUND_DIR", "round"))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, "%s_%s%s_kernel_stats.csv" % (tag, config, suffix)))


def per_kernel(which):
    tot = collections.defaultdict(float)
    with open(os.path.join(src, which, "p_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            tot[r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"])
    return tot


fetch, write = per_kernel("fetch"), per_kernel("write")
kern = {}
for k in sorted(set(fetch) | set(write)):
    rd = 2.0 * 1024.0 * fetch.get(k, 0.0) / nsteps
    wr = 1024.0 * write.get(k, 0.0) / nsteps
    if rd + wr > 0:
        kern[k] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr}
total = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in kern.values())
out = {"config": config, "cells_per_gpu": ncell, "steps_profiled": nsteps, "round": tag, "csrc_digest": bench.csrc_digest(),
       "hbm_bytes_per_step": total,
       "per_kernel": kern,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); KiB -> bytes; "
                 "reads x2 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)"}
for name in ("traffic_%s%s.json" % (config, suffix), "%s_traffic_%s%s.json" % (tag, config, suffix)):
    with open(os.path.join(dst, name), "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
