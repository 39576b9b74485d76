# Runs the same teacher-forced steps through several builds of the library in one process each and saves the states.
import numpy as np, sys, os
sys.path.insert(0, '.')
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
opt = abi.default_options(FULL_ENERGY=1)
d = domain.make_domain(64, opt, ntile=3)
f, sf, dmy = domain.make_forcing(d, 0, 48, start_doy=1)
sd0, si0 = init_state.initial_state(d, f[0])
orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
gpu = Model(d); gpu.push_forcing(f, sf, dmy)
out = []
for s in range(14):
    sd_in, si_in = orc.get_state()
    orc.step(f[s], sf[s], dmy[s])
    gpu.set_state(sd_in, si_in); gpu.dist_prec(s, 1)
    sg, ig = gpu.get_state()
    out.append((sg.copy(), ig.copy(), gpu.get_fluxes().copy()))
np.savez(sys.argv[1], sd=np.stack([o[0] for o in out]), si=np.stack([o[1] for o in out]), fl=np.stack([o[2] for o in out]), so=orc.get_state()[0])
