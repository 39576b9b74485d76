#!/usr/bin/env python3
"""Per-kernel register, scratch and spill figures of the shipped build flags (hipcc -Rpass-analysis=kernel-resource-usage on
the device side of vic_amd/csrc/vicgpu_api.hip): the table the register-pressure notes in DESIGN.md refer to.

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vic_amd import build as vb

flags = [f for f in vb.FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as td:
    cmd = [vb.HIPCC] + flags + ["--cuda-device-only", "-c", vb.SRC, "-o", os.path.join(td, "x.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
kern, cur = {}, None
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = m.group(1); kern[cur] = {}; continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur:
        kern[cur][m.group(1).strip()] = m.group(2)
demangle = subprocess.run(["c++filt"] + list(kern), stdout=subprocess.PIPE, text=True).stdout.splitlines()
print("build flags:", " ".join(flags))
print("%-62s %5s %5s %5s %8s %6s %6s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratchB", "vspill", "sspill", "occ"))
for (name, v), dm in sorted(zip(kern.items(), demangle), key=lambda kv: kv[1]):
    short = re.sub(r"\(.*", "", dm).replace("void ", "")
    print("%-62s %5s %5s %5s %8s %6s %6s %5s" % (short[:62], v.get("VGPRs", "?"), v.get("AGPRs", "?"), v.get("TotalSGPRs", "?"),
                                                 v.get("ScratchSize", "?"), v.get("VGPRs Spill", "?"), v.get("SGPRs Spill", "?"), v.get("Occupancy", "?")))
