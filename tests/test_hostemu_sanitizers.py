"""The device code under AddressSanitizer + UBSan.

GPU sanitizers are not available on the MI355X pool, so tools/hostemu compiles vic_amd/csrc/vicgpu_api.hip as host C++
(kernels as functions, one fiber per work-item, wave operations as rendezvous) with -fsanitize=address,undefined and the
parity scenarios run through it against the oracle.  A clean run means: no out-of-bounds access to the state tables or
to a kernel-local array, no signed overflow / bad shift / misaligned access, and -- with fresh device allocations
poisoned -- no read of a word the kernels never wrote.  The build is a checker only; see tools/hostemu/hip/hip_runtime.h.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tools", "hostemu", "build.sh")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def hostemu_lib(oracle_lib):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ with sanitizer runtimes")
    subprocess.check_call(["bash", BUILD], stdout=subprocess.DEVNULL)
    rt = subprocess.check_output(["bash", BUILD, "--asan-runtime"], text=True).strip()
    return os.path.join(ROOT, "tools", "hostemu", "libvicgpu_hostemu.so"), rt


def _run(lib, rt, args, poison, script="check.py", **extra_env):
    env = dict(os.environ, **extra_env, LD_PRELOAD=rt, VICGPU_LIB=lib, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", HOSTEMU_POISON="1" if poison else "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hostemu", script)] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    report = [l for l in p.stderr.splitlines() if "runtime error" in l or "ERROR: AddressSanitizer" in l]
    assert not report, "\n".join(report[:10])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("poison", [False, True])
def test_monolithic_kernel_clean(hostemu_lib, poison):
    """vic_hru_step (QUICK_FLUX, water balance, glacier HRUs, gauge correction) + vic_cell_reduce."""
    out = _run(*hostemu_lib, ["6", "4", "quickflux_melt", "bands", "waterbalance_daily", "glacier_summer", "corrprec_glacier"], poison)
    assert out.count("worst rel diff") == 5


@pytest.mark.parametrize("poison", [True])          # (the poisoned run checks everything the plain one does; one of them keeps the CPU suite short)
def test_fd_pipeline_clean(hostemu_lib, poison):
    """vic_fd_stage -> { vic_profile_solve_reg ; vic_surf_eval } rounds -> vic_fd_stage, work lists included."""
    out = _run(*hostemu_lib, ["3", "2", "frozen_fixed", "frozen_compat", "glacier_frozen"], poison)
    assert out.count("worst rel diff") == 3


def test_fd_pipeline_newton_and_generic_kernel_clean(hostemu_lib):
    """The Newton node solver (VICGPU_NODE_SOLVER=newton) in the register-resident 10-node kernel and in the generic one
    (8 nodes), and the sub-stepped water-balance case."""
    out = _run(*hostemu_lib, ["3", "2", "frozen_fixed", "frozen_wb_daily", "frozen_fixed_n8"], False, VICGPU_NODE_SOLVER="newton")
    assert out.count("worst rel diff") == 3


@pytest.mark.parametrize("poison", [True])
def test_round2_entry_points_clean(hostemu_lib, poison):
    """put_data (three kernels), state-file records (gather and scatter), forcing prefetch / swap with the on-device
    derivation of atmos[rec], and the IMPLICIT profile kernel with its explicit fall-back."""
    out = _run(*hostemu_lib, ["3"], poison, script="check_round2.py")
    assert out.count("worst rel diff") == 5
