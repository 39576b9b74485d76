import numpy as np, sys
sys.path.insert(0, '.')
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
opt = abi.default_options(FULL_ENERGY=1)
d = domain.make_domain(64, opt, ntile=3)
nsteps = 48
f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=1)
sd0, si0 = init_state.initial_state(d, f[0])
orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
gpu = Model(d); gpu.push_forcing(f, sf, dmy)
names = {v: k for k, v in C.items() if k.startswith("SD_")}
import collections
tot = collections.Counter(); nondet = 0
for s in range(30):
    sd_in, si_in = orc.get_state()
    orc.step(f[s], sf[s], dmy[s])
    so, io = orc.get_state()
    prev = None
    for rep in range(2):
        gpu.set_state(sd_in, si_in); gpu.dist_prec(s, 1)
        sg, ig = gpu.get_state()
        if prev is not None and not np.array_equal(prev, sg, equal_nan=True):
            nondet += 1
        prev = sg.copy()
        bad = np.argwhere(~(np.isclose(so, sg, rtol=1e-6, atol=1e-9) | (np.isnan(so) & np.isnan(sg))))
        for r, c in bad:
            tot[(names.get(int(r), int(r)), rep)] += 1
        if len(bad) and rep == 0 and sum(tot.values()) < 40:
            for r, c in bad[:3]:
                print("step", s, names.get(int(r), int(r)), "hru", c, "in", sd_in[r, c], "oracle", so[r, c], "gpu", sg[r, c])
print("bad by row:", dict(tot), "steps non-deterministic:", nondet)
