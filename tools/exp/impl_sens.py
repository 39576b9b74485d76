import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import scenarios
from vic_amd import init_state
from vic_amd.abi import C
from oracle import pyref
for name in ("implicit", "frozen_noflux"):
    sp, d, f, sf, dmy = scenarios.build(name, nsteps=4)
    sd0, si0 = init_state.initial_state(d, f[0])
    outs = []
    for eps in (0.0, 1e-15, -1e-15, 3e-15):
        orc = pyref.OracleModel(d)
        sd = sd0.copy()
        T0 = C["SD_NSCALAR"]; Nn = d.opt.Nnode
        sd[T0:T0 + Nn] *= (1 + eps)
        sd[C["SD_MOIST0"]:C["SD_MOIST0"] + 3] *= (1 + eps)
        orc.set_state(sd, si0)
        orc.step(f[0], sf[0], dmy[0])
        outs.append(orc.get_state()[0])
    T0 = C["SD_NSCALAR"]
    for k in range(1, 4):
        dd = np.abs(outs[k][T0:T0 + Nn] - outs[0][T0:T0 + Nn])
        print(name, "perturbation", k, "max |dT| per hru:", dd.max(axis=0)[:8])
