"""Diagnostic for tests/test_forcing_stream.py::test_device_derivation_against_oracle[daily_3h_substeps]: where the device and the
oracle part on the device-derived table, and whether the cell sits on a saturated-air branch point."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
from tests.util import rel_diff
kw = dict(FULL_ENERGY=0, dt=24, snow_step=3)
opt = abi.default_options(**kw)
d = domain.make_domain(70, opt, ntile=2)
nsteps = 12
rng = np.random.default_rng(3)
f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=80)
raw = np.zeros((nsteps, C["VIC_NRAW"], opt.dt, d.ncell))
hours = lambda v: np.repeat(f[:, C[v], :opt.NF], opt.snow_step, axis=1)
raw[:, C["VIC_RAW_AIR_TEMP"]] = hours("VIC_F_AIR_TEMP") + rng.normal(0, 1.0, raw[:, 0].shape)
raw[:, C["VIC_RAW_PREC"]] = hours("VIC_F_PREC") / opt.snow_step
raw[:, C["VIC_RAW_PRESSURE_KPA"]] = hours("VIC_F_PRESSURE") * 1e-3
raw[:, C["VIC_RAW_VP_KPA"]] = hours("VIC_F_VP") * 1e-3 * rng.uniform(0.5, 1.6, raw[:, 0].shape)
raw[:, C["VIC_RAW_SHORTWAVE"]] = hours("VIC_F_SHORTWAVE"); raw[:, C["VIC_RAW_LONGWAVE"]] = hours("VIC_F_LONGWAVE")
raw[:, C["VIC_RAW_WIND"]] = hours("VIC_F_WIND") * rng.uniform(0.0, 1.2, raw[:, 0].shape)
orc = pyref.OracleModel(d)
fo, so = orc.derive_forcing(raw, min_wind=0.4, plapse=1)
gpu = Model(d)
gpu.prefetch_forcing_raw(raw, dmy, min_wind_speed=0.4, plapse=True); gpu.swap_forcing()
dev_f = [gpu.get_forcing(s)[0] for s in range(nsteps)]
sd0, si0 = init_state.initial_state(d, fo[0])
orc.set_state(sd0, si0); gpu.set_state(sd0, si0)
names = {v: k for k, v in C.items() if k.startswith("SD_")}
cell = d.hru_iparams[C["HPI_CELL"]]
for s in range(nsteps):
    sd_in, si_in = orc.get_state()
    orc.step(dev_f[s], so[s], dmy[s])
    gpu.set_state(sd_in, si_in); gpu.dist_prec(s, 1)
    a, b = orc.get_state()[0], gpu.get_state()[0]
    a[C["SD_ERROR"]] = 0; b[C["SD_ERROR"]] = 0
    r = rel_diff(a, b, 1e-6)
    bad = np.argwhere(r > 1e-6)
    if len(bad):
        hrus = sorted(set(bad[:, 1]))
        print("step", s, "HRUs that differ:", hrus)
        for g in hrus[:3]:
            c = cell[g]
            print("  hru", g, "cell", c, "vpd sub-steps", dev_f[s][C["VIC_F_VPD"], :, c], "T", dev_f[s][C["VIC_F_AIR_TEMP"], :, c], "prec", dev_f[s][C["VIC_F_PREC"], :, c])
            for row in sorted(set(bad[bad[:, 1] == g][:, 0])):
                print("    ", names.get(int(row), row), "in", sd_in[row, g], "oracle", a[row, g], "gpu", b[row, g])
print("done")
