"""Builds the HIP extension vic_amd/libvicgpu.so for gfx950 (in-tree, so it travels with the repo snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "vicgpu_api.hip")
OUT = os.path.join(HERE, "libvicgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ftrivial-auto-var-init=zero: every automatic variable has a defined value.  The big per-lane structs of the step
# (StepConst, SurfEB, SurfSolve ...) are filled member by member and copied as a whole (parked context, sub-step hand-over),
# so some copies move bytes that are still indeterminate (padding, members a branch has not set yet).  LLVM carries those as
# `undef`; in the 512-register + scratch kernels (vic_hru_step, vic_fd_stage) hipcc 7.2 then emits scratch stores whose
# source registers are partly undefined (the machine verifier reports them) and several builds computed
# wrong numbers -- every one of them is correct, and identical for =zero and =pattern, once nothing is indeterminate.
# -mllvm -sgpr-regalloc=basic: a later source (one more struct member live to the end of vic_hru_step) failed again WITH
# zero-initialisation and with a clean machine verifier: one lane-private double (snow.depth, loaded, never changed on
# that path, stored) came back as another value.  vic_hru_step spills ~450 SGPRs and ~1000 VGPRs; of 20 code-generation
# switches tried on the failing source only -O1 and the non-splitting register allocators (-sgpr-regalloc=basic or
# -vgpr-regalloc=basic; with and without zero-init) gave correct code, the SGPR one at no cost in step time (the VGPR one
# costs 30 %).  The source is clean under ASan + UBSan (tools/hostemu).  Evidence and numbers: DESIGN.md (c).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-ftrivial-auto-var-init=zero",
         "-mllvm", "-sgpr-regalloc=basic",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=(), verbose=False, out=OUT):
    if not force and out == OUT and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + [SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-f"]
    out = OUT
    if "-o" in args:                       # tuning variant: python vic_amd/build.py -f -o vic_amd/libvicgpu_b.so -DX
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    build(force="-f" in sys.argv, extra=args, verbose=True, out=out)
