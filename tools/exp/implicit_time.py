"""Step time of the IMPLICIT soil-heat solution on the cfg3 domain (100k cells x 25 HRUs), next to the explicit solver: tuning
information, not a bench line.  usage: python tools/exp/implicit_time.py [ncell] [nsteps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from vic_amd import domain, init_state
from vic_amd.api import Model
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = bench.config("cfg3")
for implicit in (0, 1):
    opt = cfg["opt"]
    opt.IMPLICIT = implicit
    d = domain.make_domain(ncell, opt, ntile=cfg["ntile"])
    f, sf, dmy = domain.make_forcing(d, 0, nsteps + 2, start_doy=cfg["start_doy"])
    sd0, si0 = init_state.initial_state(d, f[0])
    m = Model(d)
    m.set_state(sd0, si0)
    m.push_forcing(f, sf, dmy)
    m.dist_prec(0, 2, sync=True)
    t0 = time.perf_counter()
    m.dist_prec(2, nsteps, sync=True)
    dt = (time.perf_counter() - t0) / nsteps
    si = m.get_state()[1]
    print("IMPLICIT=%d: %.2f ms per step at %d cells, cells with error flags %d" % (implicit, dt * 1e3, ncell, int((m.get_cell_errors() != 0).sum())), flush=True)
    m.close()
