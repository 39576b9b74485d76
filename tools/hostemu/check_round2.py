"""The round-2 entry points through the sanitizer build of the device code: put_data (vic_put_sum / _finish / _aggregate),
state records, forcing prefetch / swap with the on-device derivation, the IMPLICIT profile kernel with its explicit
fall-back, and QUICK_SOLVE (per-lane column end in the profile kernel).  Run by tests/test_hostemu_sanitizers.py (ASan runtime preloaded, VICGPU_LIB = the host build)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from tests import scenarios
from tests.util import rel_diff
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref


def put_data_and_records(kw, glacier, nsteps):
    opt = abi.default_options(**kw)
    d = domain.make_domain(6, opt, ntile=2, glacier_top_band=glacier)
    f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=100)
    sd0, si0 = init_state.initial_state(d, f[0])
    orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
    dev = Model(d); dev.set_state(sd0, si0); dev.push_forcing(f, sf, dmy)
    dev.put_data_config(2)
    names = [t[0] for t in dev.output_list()]
    orc.put_data(-1); dev.put_data_init()
    worst = 0.0
    for s in range(nsteps):
        sd_in, si_in = orc.get_state()
        fo, co, eo = orc.step(f[s], sf[s], dmy[s]); orc.put_data(s, f[s], co, 2)
        dev.set_state(sd_in, si_in); dev.dist_prec(s, 1)
        oo = np.concatenate([orc.get_output(n, True) for n in names])
        worst = max(worst, rel_diff(oo, dev.get_output_data(names, aggregated=True), 1e-3).max())
        if s % 2 == 1:
            dev.get_outputs(["OUT_RUNOFF", "OUT_SWE_BAND"], reset=True); orc.reset_agg()
    dev.get_balance()
    rec = dev.get_state_records()
    assert rec.shape == orc.get_state_records().shape
    dev.set_state_records(rec)
    return worst


def streaming():
    opt = abi.default_options(FULL_ENERGY=1, dt=3, snow_step=1, Nband=2)
    d = domain.make_domain(5, opt, ntile=2)
    f, sf, dmy = domain.make_forcing(d, 0, 8, start_doy=90)
    raw = np.zeros((8, C["VIC_NRAW"], opt.dt, d.ncell))
    for name, src, scale in (("VIC_RAW_AIR_TEMP", "VIC_F_AIR_TEMP", 1.0), ("VIC_RAW_PREC", "VIC_F_PREC", 1.0), ("VIC_RAW_PRESSURE_KPA", "VIC_F_PRESSURE", 1e-3),
                             ("VIC_RAW_VP_KPA", "VIC_F_VP", 1e-3), ("VIC_RAW_SHORTWAVE", "VIC_F_SHORTWAVE", 1.0), ("VIC_RAW_LONGWAVE", "VIC_F_LONGWAVE", 1.0),
                             ("VIC_RAW_WIND", "VIC_F_WIND", 1.0)):
        raw[:, C[name]] = f[:, C[src], :opt.NF] * scale
    dev = Model(d); dev.set_state(*init_state.initial_state(d, f[0]))
    dev.prefetch_forcing_raw(raw[:4], dmy[:4], 0.1, True); dev.swap_forcing()
    dev.prefetch_forcing(f[4:], sf[4:], dmy[4:])
    dev.dist_prec(0, 4)
    fg, sg = dev.get_forcing(3)
    dev.swap_forcing()
    dev.dist_prec(0, 4)
    return rel_diff(f[3], fg, 1e-9).max()


def implicit(nsteps, name="implicit_spring"):
    sp, d, f, sf, dmy = scenarios.build(name, nsteps=nsteps)
    sd0, si0 = init_state.initial_state(d, f[0])
    orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
    dev = Model(d); dev.push_forcing(f, sf, dmy)
    worst = 0.0
    for s in range(nsteps):
        sd_in, si_in = orc.get_state()
        orc.step(f[s], sf[s], dmy[s])
        dev.set_state(sd_in, si_in); dev.dist_prec(s, 1)
        a, b = orc.get_state()[0], dev.get_state()[0]
        a[C["SD_ERROR"]] = 0; b[C["SD_ERROR"]] = 0
        worst = max(worst, rel_diff(a, b, 1e-2).max())
    return worst


def main():
    frozen = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=2, frozen_compat=0)
    w = put_data_and_records(dict(FULL_ENERGY=1, Nband=3), True, 4)
    print("hostemu put_data quickflux_glacier: worst rel diff %.3e" % w, flush=True); ok = w < 1e-6
    w = put_data_and_records(frozen, True, 2)
    print("hostemu put_data frozen_glacier: worst rel diff %.3e" % w, flush=True); ok = ok and w < 1e-6
    w = streaming()
    print("hostemu streaming: worst rel diff %.3e" % w, flush=True); ok = ok and w < 1e-12
    w = implicit(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    print("hostemu implicit: worst rel diff %.3e" % w, flush=True); ok = ok and w < 1e-3
    w = implicit(3, "quick_solve")
    print("hostemu quick_solve: worst rel diff %.3e" % w, flush=True); ok = ok and w < 1e-6
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
