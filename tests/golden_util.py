import glob
import os

import numpy as np

from vic_amd import abi, domain
from vic_amd.abi import C

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("pure_", "gmb_", "forcing_derive_"))]      # pure_*.npz: per-function vectors, tests/pure_inputs.py


def load_golden(name):
    """Returns (Domain rebuilt from the stored tables, npz dict)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    opt = abi.default_options()
    for k, v in zip(z["opt_names"], z["opt_values"]):
        k = str(k)
        if k == "abi_version":
            continue                      # the fixture stores data, not the ABI revision it was written through
        setattr(opt, k, float(v) if k == "wind_h" else int(v))
    d = domain.Domain()
    d.opt = opt
    d.veglib = np.ascontiguousarray(z["veglib"])
    d.cell_params = np.ascontiguousarray(z["cell_params"])
    d.hru_iparams = np.ascontiguousarray(z["hru_iparams"].astype(np.int32))
    hpd = z["hru_dparams"]
    if hpd.shape[0] < C["HPD_NROW"]:      # fixtures written before the blowing-snow rows existed: the values the harness used then
        pad = np.zeros((C["HPD_NROW"], hpd.shape[1])); pad[:hpd.shape[0]] = hpd
        pad[C["HPD_SIGMA_SLOPE"]] = np.float32(0.08); pad[C["HPD_LAG_ONE"]] = np.float32(0.95); pad[C["HPD_FETCH"]] = np.float32(1000.0)
        hpd = pad
    d.hru_dparams = np.ascontiguousarray(hpd)
    d.cell_hru_offset = np.ascontiguousarray(z["cell_hru_offset"].astype(np.int32))
    d.cell_hru_list = np.ascontiguousarray(z["cell_hru_list"].astype(np.int32))
    d.init_moist = np.ascontiguousarray(z["init_moist"])
    d.ncell = d.cell_params.shape[1]
    d.nhru = d.hru_iparams.shape[1]
    return d, z
