"""Which state entries of a QUICK_FLUX step differ between the device and the oracle, with the snow rows of the first few."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
from tests.util import rel_diff
opt = abi.default_options(FULL_ENERGY=1)
d = domain.make_domain(32, opt, ntile=3)
f, sf, dmy = domain.make_forcing(d, 0, 24, start_doy=100)
sd0, si0 = init_state.initial_state(d, f[0])
orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
gpu = Model(d); gpu.push_forcing(f, sf, dmy)
names = {v: k for k, v in C.items() if k.startswith("SD_")}
shown = 0
for s in range(24):
    sd_in, si_in = orc.get_state()
    orc.step(f[s], sf[s], dmy[s])
    so, io = orc.get_state()
    gpu.set_state(sd_in, si_in); gpu.dist_prec(s, 1)
    sg, ig = gpu.get_state()
    so[C["SD_ERROR"]] = 0; sg[C["SD_ERROR"]] = 0
    dd = rel_diff(so, sg, 1e-6)
    bad = np.argwhere(dd > 1e-6)
    if len(bad):
        print("step", s, "bad entries", len(bad), "rows", sorted(set(names.get(int(r), int(r)) for r in bad[:, 0])), "hrus", sorted(set(int(c) for c in bad[:, 1]))[:10])
        for g in sorted(set(int(c) for c in bad[:, 1]))[:3]:
            if shown >= 6: break
            shown += 1
            for r in ("SD_SNOW_SWQ", "SD_SNOW_DEPTH", "SD_SNOW_DENSITY", "SD_SNOW_COVERAGE", "SD_SNOW_SURF_WATER", "SD_SNOW_PACK_WATER", "SD_SNOW_SURF_TEMP", "SD_SNOW_ALBEDO", "SD_SNOW_CANOPY", "SD_SNOW_STORE_SWQ"):
                print("   hru %d %-22s in %-22r oracle %-22r gpu %-22r" % (g, r, sd_in[C[r], g], so[C[r], g], sg[C[r], g]))
            for r in ("SI_SNOW_SNOW", "SI_SNOW_MELTING", "SI_SNOW_STORE_SNOW", "SI_SNOW_LAST_SNOW"):
                print("   hru %d %-22s in %-22r oracle %-22r gpu %-22r" % (g, r, si_in[C[r], g], io[C[r], g], ig[C[r], g]))
print("done")
