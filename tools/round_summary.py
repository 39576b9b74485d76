#!/usr/bin/env python3
"""Turns the output of tools/profile_round.sh (gpurun_out/round/) into the committed round profile:

  profiles/rNN_<config>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/traffic_<config>.json           HBM bytes per step from the FETCH_SIZE / WRITE_SIZE passes, stamped with the digest of
                                           the device sources it was measured on (bench.py reads it and refuses a stale one);
  profiles/rNN_traffic_<config>.json       the same file, kept per round

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a 16-byte-per-lane streaming read.  The factor applied to the reads of these kernels is the one MEASURED for their
access patterns by tools/calib/fetch_calib.hip in the same gpurun call (gpurun_out/<round>/calib -> profiles/fetch_calibration.json):
wave slabs read 16 bytes per lane (parked context: stage, evaluation and put_data kernels) and per-lane contiguous blocks
(item blocks: profile kernels); 2.0 where no calibration is at hand.  Also written: profiles/pmc_<config>.json, lanes active
per vector instruction and the VALU-active share of the wave cycles over the pipeline's kernels (SQ pass).

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


# calibration of this call, else the committed one, else the guide's 2.0
calib = {}
cdir = os.path.join(src, "calib")
if os.path.isdir(cdir):
    import subprocess
    nb = 65536 * 96 * 64 * 8
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "calib", "fetch_calib.py"), cdir, str(nb)])
cpath = os.path.join(dst, "fetch_calibration.json")
if os.path.exists(cpath):
    with open(cpath) as f:
        calib = {k: v["true_bytes_per_counted_byte"] for k, v in json.load(f)["kernels"].items()}
f_slab = calib.get("read_slab16", 2.0)
f_block = calib.get("read_block16", 2.0)


def read_factor(kernel):
    return f_block if "profile_solve" in kernel else f_slab


fetch, write = per_kernel("fetch"), per_kernel("write")
kern = {}
for k in sorted(set(fetch) | set(write)):
    rd = read_factor(k) * 1024.0 * fetch.get(k, 0.0) / nsteps
    wr = 1024.0 * write.get(k, 0.0) / nsteps
    if rd + wr > 0:
        kern[k] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr}
total = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in kern.values())
out = {"config": config, "cells_per_gpu": ncell, "steps_profiled": nsteps, "round": tag, "csrc_digest": bench.csrc_digest(),
       "hbm_bytes_per_step": total,
       "per_kernel": kern,
       "read_factor": {"slab (stage / evaluation / put_data kernels)": f_slab, "per-lane blocks (profile kernels)": f_block,
                       "source": "profiles/fetch_calibration.json" if calib else "MI355X_MICROARCH.md (uncalibrated for these patterns)"},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); KiB -> bytes; "
                 "reads x the factor measured for the kernel's access pattern (tools/calib/fetch_calib.hip)"}
for name in ("traffic_%s%s.json" % (config, suffix), "%s_traffic_%s%s.json" % (tag, config, suffix)):
    with open(os.path.join(dst, name), "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
# SQ pass -> lanes active / VALU share over the pipeline's kernels
sqf = os.path.join(src, "sq", "p_counter_collection.csv")
import glob
sqs = glob.glob(os.path.join(src, "sq", "**", "*counter_collection.csv"), recursive=True)
if sqs:
    tot = collections.defaultdict(float)
    perk = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in sqs:
        with open(fn) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "vic" not in k:
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); perk[k][r["Counter_Name"]] += float(r["Counter_Value"])
    pj = {"config": config, "round": tag, "csrc_digest": bench.csrc_digest(),
          "lanes_active_per_valu_inst": tot["SQ_THREAD_CYCLES_VALU"] / max(tot["SQ_ACTIVE_INST_VALU"], 1),
          "valu_active_share_of_wave_cycles": tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1),
          "wave_parked_share_of_wave_cycles": tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1),
          "valu_insts_per_step": tot["SQ_INSTS_VALU"] / nsteps,
          "per_kernel": {k: {"lanes_active_per_valu_inst": v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"], 1),
                             "valu_active_share_of_wave_cycles": v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_WAVE_CYCLES"], 1),
                             "valu_insts_per_step": v["SQ_INSTS_VALU"] / nsteps} for k, v in perk.items()},
          "method": "rocprofv3 --pmc SQ_* (tools/profile_round.sh), summed over the dispatches of the bench command"}
    for name in ("pmc_%s%s.json" % (config, suffix), "%s_pmc_%s%s.json" % (tag, config, suffix)):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(pj, f, indent=1)
    print(json.dumps({k: v for k, v in pj.items() if k != "per_kernel"}, indent=1))
