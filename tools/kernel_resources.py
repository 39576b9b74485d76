#!/usr/bin/env python3
"""Per-kernel register, scratch and spill figures of the shipped build (the compiler's kernel-resource-usage remarks, which
vic_amd/build.py records at every build and checks against RESOURCE_LIMITS): prints vic_amd/libvicgpu.resources.txt, building
first if needed.

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vic_amd import build as vb

path = os.path.splitext(vb.OUT)[0] + ".resources.txt"
if vb.needs_build() or not os.path.exists(path):
    vb.build(force=True)
print("build flags:", " ".join(f for f in vb.FLAGS if f not in ("-shared", "-fPIC")))
sys.stdout.write(open(path).read())
