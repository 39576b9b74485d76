"""Teacher-forced steps of the GPU parity scenarios through the sanitizer build of the device code, against the oracle.
Run by tests/test_hostemu_sanitizers.py with the ASan runtime preloaded and VICGPU_LIB pointing at the host build:
    python tools/hostemu/check.py <ncell> <nsteps> <case> [<case> ...]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as tg
from tests.util import worst
from vic_amd.abi import C
from vic_amd.api import Model
from oracle import pyref


def main():
    ncell, nsteps = int(sys.argv[1]), int(sys.argv[2])
    rc = 0
    for name in sys.argv[3:]:
        kw, _, ntile, doy = tg.CASES[name]
        d, f, sf, dmy, sd0, si0 = tg._setup(kw, ncell, ntile, nsteps, doy)
        # VICGPU_NODE_SOLVER=newton: the Newton node solver against the oracle with converged node roots (tests/test_gpu_parity.py)
        orc = pyref.OracleModel(d, converged_nodes=os.environ.get("VICGPU_NODE_SOLVER") == "newton"); orc.set_state(sd0, si0)
        dev = Model(d); dev.push_forcing(f, sf, dmy)
        w_all = 0.0
        for s in range(nsteps):
            sd_in, si_in = orc.get_state()
            fo, co, eo = orc.step(f[s], sf[s], dmy[s])
            so, io = orc.get_state()
            dev.set_state(sd_in, si_in); dev.dist_prec(s, 1)
            sg, ig = dev.get_state()
            so[C["SD_ERROR"]] = 0; sg[C["SD_ERROR"]] = 0
            w1, m1 = worst(so, sg, "SD_", floor=1e-6)
            w3, m3 = worst(co, dev.get_cell_outputs(), "CO_", floor=1e-6)
            w_all = max(w_all, w1, w3)
            if not np.array_equal(io, ig): w_all = max(w_all, 1.0)
        print("hostemu %s: worst rel diff %.3e" % (name, w_all), flush=True)
        if not w_all < 1e-6: rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
