"""Tuning / debugging aid: teacher-forced steps of one GPU parity case, printing the rows that differ most from the oracle.
    python tools/exp/rowdiff.py <case> [nsteps]"""
import os, sys, numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as tg
from tests.util import rel_diff
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
name = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw, ncell, ntile, doy = tg.CASES[name]
d, f, sf, dmy, sd0, si0 = tg._setup(kw, ncell, ntile, nsteps, doy)
orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
dev = Model(d); dev.push_forcing(f, sf, dmy)
names = {v: k for k, v in C.items() if k.startswith("SD_")}
fnames = {v: k for k, v in C.items() if k.startswith("FX_")}
for s in range(nsteps):
    sd_in, si_in = orc.get_state()
    fo, co, eo = orc.step(f[s], sf[s], dmy[s])
    so, io = orc.get_state()
    dev.set_state(sd_in, si_in); dev.dist_prec(s, 1)
    sg, ig = dev.get_state(); fg = dev.get_fluxes()
    dd = rel_diff(so, sg, 1e-6)
    bad = np.argwhere(dd > 1e-6)
    print("step", s, "bad state entries", len(bad), "bad hrus", sorted(set(bad[:, 1].tolist()))[:20])
    rows = np.argsort(-dd.max(axis=1))[:10]
    for r in rows:
        h = int(dd[r].argmax())
        print("   %-28s nbad %3d worst hru %3d orc %.10g gpu %.10g" % (names.get(int(r), "node row %d" % (r - C["SD_NSCALAR"])), int((dd[r] > 1e-6).sum()), h, so[r, h], sg[r, h]))
    df = rel_diff(fo, fg, 1e-6); df[~np.isfinite(df)] = 0
    rows = np.argsort(-df.max(axis=1))[:8]
    for r in rows:
        h = int(df[r].argmax())
        print("   %-28s nbad %3d worst hru %3d orc %.10g gpu %.10g" % (fnames.get(int(r), r), int((df[r] > 1e-6).sum()), h, fo[r, h], fg[r, h]))
    slot = np.arange(d.nhru) // d.ncell
    print("   bad hru slots:", sorted(set(slot[sorted(set(bad[:, 1].tolist()))].tolist())), " int diffs", np.argwhere(io != ig)[:6].tolist())
