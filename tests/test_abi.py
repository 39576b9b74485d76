"""The Python mirror of include/vicgpu.h, and the C-ABI library's exports (no GPU needed)."""
import os
import re
import subprocess
import tempfile

import pytest

from vic_amd import abi
from vic_amd.abi import C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_enum_parse_matches_c_compiler():
    names = sorted(k for k in C if re.match(r"^(SD_|SI_|SDN_|SIN_|FX_|CP_|CPL_|CPN_|CPB_|HPI_|HPD_|VL_|VIC_F_|VIC_DMY_|CO_|CA_|VIC_N)", k))
    src = ['#include <stdio.h>', '#include "vicgpu.h"', 'int main(void){']
    for n in names:
        src.append('printf("%s %%d\\n", (int)%s);' % (n, n))
    src.append('printf("CPNROW %d\\n", VICGPU_CP_NROW(10,5)); printf("SDNROW %d\\n", VICGPU_SD_NROW(10)); printf("SINROW %d\\n", VICGPU_SI_NROW(10));')
    src.append('printf("CPZM %d\\n", VICGPU_CP_ZWT_MOIST(4,10,10,5)); printf("OPTSZ %d\\n", (int)sizeof(vicgpu_options)); return 0;}')
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "t.c")
        open(cfile, "w").write("\n".join(src))
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    got = dict(zip(out[0::2], map(int, out[1::2])))
    for n in names:
        assert got[n] == C[n], n
    assert got["CPNROW"] == abi.cp_nrow(10, 5)
    assert got["SDNROW"] == abi.sd_nrow(10)
    assert got["SINROW"] == abi.si_nrow(10)
    assert got["CPZM"] == abi.cp_zwt_moist(4, 10, 10, 5)
    import ctypes
    assert got["OPTSZ"] == ctypes.sizeof(abi.Options)


def test_library_exports_every_declared_symbol():
    """libvicgpu.so (hipcc cross-compiles it without a GPU) exports every entry point include/*.h declares."""
    from vic_amd import build as vb
    from vic_amd import api
    lib = vb.build(force=False)
    import glob, os
    hdr = "".join(open(h).read() for h in sorted(glob.glob(os.path.join(os.path.dirname(abi.HEADER), "*.h"))))
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(vicgpu_[a-z_]+)\s*\(", hdr)))
    declared = [d for d in declared if d not in ("vicgpu_ctx", "vicgpu_options")]
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    for d in declared:
        assert re.search(r"\bT %s\b" % d, syms), "missing export " + d
    assert sorted(api.EXPORTED_SYMBOLS) == declared
    # loading the library (no compute call) works without a GPU
    api.load_library()


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: without a visible GPU vicgpu_create must fail (VICGPU_ERR_HIP)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from vic_amd import domain, api
    opt = abi.default_options(FULL_ENERGY=1)
    d = domain.make_domain(4, opt)
    with pytest.raises(api.VicGpuError):
        api.Model(d)


def test_kernel_resource_limits_hold():
    """vic_amd/build.py records the compiler's per-kernel register / scratch figures at every build and refuses a build in
    which a kernel with a designed register budget spills (RESOURCE_LIMITS): the table exists, names every kernel family and
    passes the check."""
    import os
    from vic_amd import build as vb
    vb.build(force=False)
    path = os.path.splitext(vb.OUT)[0] + ".resources.txt"
    if not os.path.exists(path):
        vb.build(force=True)
    rows = [l.split() for l in open(path).read().splitlines()[1:]]
    names = " ".join(r[0] for r in rows)
    for k in ("vic_surf_eval", "vic::vic_profile_solve_reg<10,", "vic_fd_stage<10,", "vic_hru_step<3,", "vic::vic_put_sum", "vic_profile_solve_implicit"):
        assert k in names, k
    table = [(" ".join(r[:-7]),) + tuple(r[-7:]) for r in rows]
    assert vb.check_resources(table) == []
    bad = [("vic_surf_eval", "256", "0", "106", "548", "152", "55", "2")]
    assert vb.check_resources(bad), "the check must catch a spilling evaluation kernel"
