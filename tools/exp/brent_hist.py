# Histogram of Brent evaluations per node solve on a small cfg3 domain, through the host build of the device code:
#   SAN=none EXTRA='-DVIC_HOSTEMU_HIST -O2' bash tools/hostemu/build.sh
#   VICGPU_LIB=$PWD/tools/hostemu/libvicgpu_hostemu_plain.so python tools/exp/brent_hist.py
import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from vic_amd import domain, init_state
from vic_amd.api import Model
cfg = bench.config("cfg3")
d = domain.make_domain(24, cfg["opt"], ntile=cfg["ntile"])
f, sf, dmy = domain.make_forcing(d, 0, 4, start_doy=cfg["start_doy"])
sd0, si0 = init_state.initial_state(d, f[0])
m = Model(d, device=0); m.set_state(sd0, si0); m.push_forcing(f, sf, dmy)
m.dist_prec(0, 3)
print("done")
