import numpy as np, sys
sys.path.insert(0, '/root/repo' if False else '.')
from vic_amd import abi, domain, init_state
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref
from tests.util import rel_diff
kw = dict(FULL_ENERGY=1, FROZEN_SOIL=1, Nnode=10, Nband=2, frozen_compat=0)
opt = abi.default_options(**kw)
d = domain.make_domain(24, opt, ntile=3)
nsteps = 720
f, sf, dmy = domain.make_forcing(d, 0, nsteps, start_doy=80)
sd0, si0 = init_state.initial_state(d, f[0])
orc = pyref.OracleModel(d); orc.set_state(sd0, si0)
gpu = Model(d); gpu.set_state(sd0, si0); gpu.push_forcing(f, sf, dmy)
first = None
for s in range(nsteps):
    orc.step(f[s], sf[s], dmy[s])
    gpu.dist_prec(s, 1)
    so, io = orc.get_state(); sg, ig = gpu.get_state()
    so[C["SD_ERROR"]] = 0; sg[C["SD_ERROR"]] = 0
    dd = rel_diff(so, sg, 1e-6)
    w = dd.max()
    if w > 1e-9 and (first is None or s % 48 == 0 or w > 1e-5):
        r, c = np.unravel_index(np.argmax(dd), dd.shape)
        names = {v: k for k, v in C.items() if k.startswith("SD_")}
        print("step", s, "worst %.3e" % w, "row", names.get(int(r), int(r)), "hru", c, so[r, c], sg[r, c], "ints differ", int((io != ig).sum()))
        if first is None: first = s
    if w > 1e-3: break

# teacher-forced replay of the diverging step
orc2 = pyref.OracleModel(d); orc2.set_state(sd0, si0)
for s in range(563):
    orc2.step(f[s], sf[s], dmy[s])
sd_in, si_in = orc2.get_state()
fo, co, eo = orc2.step(f[563], sf[563], dmy[563])
so, io = orc2.get_state()
gpu.set_state(sd_in, si_in); gpu.dist_prec(563, 1)
sg, ig = gpu.get_state()
h = 76
names = {v: k for k, v in C.items() if k.startswith("SD_")}
inames = {v: k for k, v in C.items() if k.startswith("SI_")}
print("teacher-forced step 563, hru", h)
for r in range(so.shape[0]):
    if not (so[r, h] == sg[r, h] or (np.isnan(so[r, h]) and np.isnan(sg[r, h]))) and abs(so[r, h] - sg[r, h]) > 1e-9 * max(1, abs(so[r, h])):
        print("  SD", names.get(r, r), "in", sd_in[r, h], "oracle", so[r, h], "gpu", sg[r, h])
for r in range(io.shape[0]):
    if io[r, h] != ig[r, h]:
        print("  SI", inames.get(r, r), "in", si_in[r, h], "oracle", io[r, h], "gpu", ig[r, h])
print("forcing air_temp", f[563][C["VIC_F_AIR_TEMP"], :, d.hru_iparams[C["HPI_CELL"], h]], "swq in", sd_in[C["SD_SNOW_SWQ"], h], "surf_temp in", sd_in[C["SD_SNOW_SURF_TEMP"], h])

# the other way round: the oracle started from the GPU's own state before the diverging step
gpu3 = Model(d); gpu3.set_state(sd0, si0); gpu3.push_forcing(f, sf, dmy)
gpu3.dist_prec(0, 563)
sg_in, ig_in = gpu3.get_state()
orc3 = pyref.OracleModel(d); orc3.set_state(sg_in, ig_in)
orc3.step(f[563], sf[563], dmy[563])
so3, io3 = orc3.get_state()
gpu3.dist_prec(563, 1)
sg3, ig3 = gpu3.get_state()
so3[C["SD_ERROR"]] = 0; sg3[C["SD_ERROR"]] = 0
dd = rel_diff(so3, sg3, 1e-6)
print("oracle from the GPU state at 562 vs GPU at 563: worst %.3e, ints differ %d" % (dd.max(), int((io3 != ig3).sum())))
print("  hru 76 NETLONGUNDER oracle-from-gpu-state", so3[C["SD_NETLONGUNDER"], 76], "gpu", sg3[C["SD_NETLONGUNDER"], 76], "tsurf fbcount", io3[C["SI_TSURF_FBCOUNT"], 76], ig3[C["SI_TSURF_FBCOUNT"], 76], "in", ig_in[C["SI_TSURF_FBCOUNT"], 76])
print("  state difference going in (gpu vs oracle free run) max", rel_diff(sg_in, sd_in, 1e-6).max())
