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


# Build-time resource check (hipcc -Rpass-analysis=kernel-resource-usage, written next to the library): the kernels whose
# register budget is part of their design must not spill behind one's back -- a few hundred bytes of scratch in the evaluation
# kernel cost 5 ms per step twice this round.  Limits are per kernel name prefix: (max VGPRs, max scratch bytes).
RESOURCE_LIMITS = {
    # the 512-register kernels (one wave per SIMD, spills into AGPRs and scratch by design): ceilings a little above today's
    # figures, so that a change that inflates them -- the state in which round 2 saw hipcc produce wrong code -- stops the build
    "vic_hru_step<3, false>": (256, 3200),
    "vic_hru_step<3, true>": (256, 1800),
    "vic_hru_step<10, true>": (256, 2200),
    "vic_fd_stage<10, true, false>": (256, 800),
    "vic_fd_stage<10, false, false>": (256, 600),
    "vic_surf_eval": (256, 0),
    "vic::vic_profile_solve_reg<10": (256, 128),
    "vic::vic_profile_solve_lockstep": (256, 0),
    "vic::vic_put_": (256, 0),
    "vic_cell_reduce": (128, 0),
}


def parse_resources(remarks):
    """{kernel symbol: {field: value}} from the compiler's kernel-resource-usage remarks."""
    import re
    kern, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = m.group(1); kern[cur] = {}; continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            kern[cur][m.group(1).strip()] = m.group(2)
    return kern


def resource_table(kern):
    import re
    names = list(kern)
    dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.splitlines() if names else []
    rows = []
    for name, dm in zip(names, dem):
        v = kern[name]
        short = re.sub(r"\(.*", "", dm).replace("void ", "")
        rows.append((short, v.get("VGPRs", "?"), v.get("AGPRs", "?"), v.get("TotalSGPRs", "?"), v.get("ScratchSize", "?"),
                     v.get("VGPRs Spill", "?"), v.get("SGPRs Spill", "?"), v.get("Occupancy", "?")))
    rows.sort()
    return rows


def check_resources(rows):
    """List of violations of RESOURCE_LIMITS."""
    bad = []
    for r in rows:
        for prefix, (max_vgpr, max_scratch) in RESOURCE_LIMITS.items():
            if r[0].startswith(prefix):
                try:
                    if int(r[1]) > max_vgpr or int(r[4]) > max_scratch:
                        bad.append("%s: %s VGPRs, %s B scratch (limit %d / %d)" % (r[0], r[1], r[4], max_vgpr, max_scratch))
                except ValueError:
                    pass
    return bad


def build(force=False, extra=(), verbose=False, out=OUT):
    if not force and out == OUT and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise subprocess.CalledProcessError(p.returncode, cmd)
    rows = resource_table(parse_resources(p.stdout))
    with open(os.path.splitext(out)[0] + ".resources.txt", "w") as f:
        f.write("%-62s %5s %5s %5s %8s %6s %6s %5s\n" % ("kernel", "VGPR", "AGPR", "SGPR", "scratchB", "vspill", "sspill", "occ"))
        for r in rows:
            f.write("%-62s %5s %5s %5s %8s %6s %6s %5s\n" % ((r[0][:62],) + r[1:]))
    bad = check_resources(rows)
    if bad and not extra:            # tuning variants (extra flags) are measured, not policed
        raise RuntimeError("kernel resource limits exceeded (vic_amd/build.py RESOURCE_LIMITS):\n  " + "\n  ".join(bad))
    return OUT


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-f"]
    out = OUT
    if "-o" in args:                       # tuning variant: python vic_amd/build.py -f -o vic_amd/libvicgpu_b.so -DX
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    build(force="-f" in sys.argv, extra=args, verbose=True, out=out)
