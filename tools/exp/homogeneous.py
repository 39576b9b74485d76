#!/usr/bin/env python3
"""Experiment: how much of the profile kernel's lane under-utilisation is heterogeneity between the HRUs of a wave?
Runs the cfg3 bench workload with every cell replaced by a copy of one of K template cells (K = 1: all waves perfectly
homogeneous; K large: as heterogeneous as the bench).  Prints ms/step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from vic_amd import domain, init_state
from vic_amd.api import Model

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
block = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # consecutive cells sharing a template
cfg = bench.config("cfg3")
d = domain.make_domain(ncell, cfg["opt"], ntile=cfg["ntile"])
f, sf, dmy = domain.make_forcing(d, 0, 5, start_doy=cfg["start_doy"])
sd0, si0 = init_state.initial_state(d, f[0])
tmpl = ((np.arange(ncell) // block) % K)
d.cell_params = np.ascontiguousarray(d.cell_params[:, tmpl])
f = np.ascontiguousarray(f[..., tmpl]); sf = np.ascontiguousarray(sf[..., tmpl])
nslot = d.nhru // ncell
idx = (np.arange(nslot)[:, None] * ncell + tmpl[None, :]).reshape(-1)
sd0 = np.ascontiguousarray(sd0[:, idx]); si0 = np.ascontiguousarray(si0[:, idx])
d.hru_dparams = np.ascontiguousarray(d.hru_dparams[:, idx])
m = Model(d)
m.set_state(sd0, si0); m.set_write_fluxes(False); m.push_forcing(f, sf, dmy)
m.dist_prec(0, 1)
t0 = time.perf_counter(); m.dist_prec(1, 4); t1 = time.perf_counter()
print("K=%d block=%d: %.2f ms/step" % (K, block, (t1 - t0) / 4 * 1e3))
