import numpy as np

from vic_amd.abi import C


def rel_diff(a, b, floor=1e-9):
    """Element-wise relative difference with an absolute floor; NaN == NaN counts as equal."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(floor, np.maximum(np.abs(a), np.abs(b)))
    d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
    d = np.where(a == b, 0.0, d)
    d = np.where(np.isnan(d), np.inf, d)
    return d


def row_names(prefix):
    return {v: k for k, v in C.items() if k.startswith(prefix)}


def worst(a, b, prefix, floor=1e-9):
    d = rel_diff(a, b, floor)
    r, c = np.unravel_index(np.argmax(d), d.shape)
    return float(d[r, c]), "%s[%s] col %d: %r vs %r" % (prefix, row_names(prefix).get(int(r), int(r)), c, a[r, c], b[r, c])


# rows of the flux table that both implementations define for non-glacier HRUs
FLUX_ROWS_COMMON = [C[k] for k in (
    "FX_RUNOFF", "FX_BASEFLOW", "FX_ASAT", "FX_INFLOW", "FX_EVAP0", "FX_EVAP1", "FX_EVAP2", "FX_CANOPYEVAP", "FX_THROUGHFALL",
    "FX_SNOW_VAPOR_FLUX", "FX_SNOW_CANOPY_VAPOR_FLUX", "FX_SNOW_BLOWING_FLUX", "FX_SNOW_SURFACE_FLUX", "FX_SNOW_MELT",
    "FX_POT_EVAP0", "FX_POT_EVAP1", "FX_POT_EVAP2", "FX_POT_EVAP3", "FX_POT_EVAP4", "FX_POT_EVAP5",
    "FX_AERO_RESIST_SURFACE", "FX_AERO_RESIST_OVERSTORY", "FX_ROOTMOIST", "FX_WETNESS", "FX_ZWT", "FX_ZWT2", "FX_ZWT3",
    "FX_ATMOS_LATENT", "FX_ATMOS_LATENT_SUB", "FX_ATMOS_SENSIBLE", "FX_LONG_UNDER_IN", "FX_NET_LONG_ATMOS", "FX_NET_LONG_UNDER",
    "FX_NET_SHORT_ATMOS", "FX_NET_SHORT_GRND", "FX_NET_SHORT_UNDER", "FX_SHORT_UNDER_IN")]
