"""Python mirror of include/vicgpu.h: the enum row indices and the options struct.

The header is the single source of truth: the enums are parsed from it at import
time (tests/test_abi.py cross-checks the parse against a C program compiled from the
same header), so the Python host layer, the tests and the C/HIP code can never
disagree about a row number.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vicgpu.h")


def _parse_header(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    consts = {}
    # simple object-like integer macros
    for m in re.finditer(r"^#define\s+([A-Z_0-9a-z]+)\s+(-?\d+)\s*$", txt, flags=re.M):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", txt, flags=re.S):
        if "(" in m.group(1) and "VICGPU_OUT_VARS" in m.group(1):
            continue                                  # generated from the X-macro list
        val = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [s.strip() for s in item.split("=", 1)]
                val = int(eval(expr, {}, consts))  # expressions only reference earlier constants
            else:
                name = item
                val += 1
            consts[name] = val
    return consts


C = _parse_header(HEADER)
# put_data's tables (include/vicgpu_out.h): the plain enums; the variable list itself is an X-macro, read from the library
C.update(_parse_header(os.path.join(os.path.dirname(HEADER), "vicgpu_out.h")))
globals().update(C)

VIC_NLAYER = C["VIC_NLAYER"]
VIC_MAX_NODES = C["VIC_MAX_NODES"]
VIC_MAX_ZWTVMOIST = C["VIC_MAX_ZWTVMOIST"]
VIC_NZWT_ROWS = (VIC_NLAYER + 2) * VIC_MAX_ZWTVMOIST


def cp_layer(f, l):
    return C["CP_NSCALAR"] + f * VIC_NLAYER + l


def cp_node0():
    return C["CP_NSCALAR"] + C["CPL_NFIELD"] * VIC_NLAYER


def cp_node(f, n, Nn):
    return cp_node0() + f * Nn + n


def cp_band0(Nn):
    return cp_node0() + C["CPN_NFIELD"] * Nn


def cp_band(f, b, Nn, Nb):
    return cp_band0(Nn) + f * Nb + b


def cp_zwt0(Nn, Nb):
    return cp_band0(Nn) + C["CPB_NFIELD"] * Nb


def cp_zwt_zwt(l, i, Nn, Nb):
    return cp_zwt0(Nn, Nb) + l * VIC_MAX_ZWTVMOIST + i


def cp_zwt_moist(l, i, Nn, Nb):
    return cp_zwt0(Nn, Nb) + VIC_NZWT_ROWS + l * VIC_MAX_ZWTVMOIST + i


def cp_nrow(Nn, Nb):
    return cp_zwt0(Nn, Nb) + 2 * VIC_NZWT_ROWS


def sd_node(f, n, Nn):
    return C["SD_NSCALAR"] + f * Nn + n


def sd_nrow(Nn):
    return C["SD_NSCALAR"] + C["SDN_NFIELD"] * Nn


def si_node(f, n, Nn):
    return C["SI_NSCALAR"] + f * Nn + n


def si_nrow(Nn):
    return C["SI_NSCALAR"] + C["SIN_NFIELD"] * Nn


def sr_t(f, Nn):
    """Position of an SRT_* field of a state-file record (VICGPU_SR_T)."""
    return C["SR_ENERGY_T"] + Nn + f


def sr_u(f, Nn):
    return C["SR_ENERGY_T"] + Nn + C["SRT_T_FBCOUNT"] + Nn + f


def sr_len(Nn):
    return sr_u(C["SRU_NFIELD"], Nn)


class Options(ctypes.Structure):
    """struct vicgpu_options (include/vicgpu.h)."""
    _fields_ = [
        ("abi_version", ctypes.c_int), ("Nlayer", ctypes.c_int), ("Nnode", ctypes.c_int), ("Nband", ctypes.c_int),
        ("dt", ctypes.c_int), ("snow_step", ctypes.c_int), ("FULL_ENERGY", ctypes.c_int), ("FROZEN_SOIL", ctypes.c_int),
        ("QUICK_FLUX", ctypes.c_int), ("NOFLUX", ctypes.c_int), ("EXP_TRANS", ctypes.c_int),
        ("GRND_FLUX_TYPE", ctypes.c_int), ("TFALLBACK", ctypes.c_int), ("AERO_RESIST_CANSNOW", ctypes.c_int),
        ("SNOW_ALBEDO", ctypes.c_int), ("SNOW_DENSITY", ctypes.c_int), ("TEMP_TH_TYPE", ctypes.c_int),
        ("GLACIER_ID", ctypes.c_int), ("GLACIER_DYNAMICS", ctypes.c_int), ("frozen_compat", ctypes.c_int),
        ("nveg_types", ctypes.c_int), ("CORRPREC", ctypes.c_int), ("IMPLICIT", ctypes.c_int), ("BLOWING", ctypes.c_int),
        ("QUICK_SOLVE", ctypes.c_int), ("NODE_SOLVER", ctypes.c_int),
        ("wind_h", ctypes.c_double), ("reserved_d", ctypes.c_double * 3),
    ]

    @property
    def NF(self):
        r = self.dt // self.snow_step
        return 1 if r == 1 else r

    @property
    def NR(self):
        r = self.dt // self.snow_step
        return 0 if r == 1 else r


def default_options(**kw):
    """Defaults of initialize_global.c:127-182 for the fields the path reads."""
    o = Options()
    o.abi_version = C["VICGPU_ABI_VERSION"]
    o.Nlayer = 3
    o.Nnode = 3
    o.Nband = 1
    o.dt = 1
    o.snow_step = 1
    o.FULL_ENERGY = 0
    o.FROZEN_SOIL = 0
    o.QUICK_FLUX = 1
    o.NOFLUX = 0
    o.EXP_TRANS = 0
    o.GRND_FLUX_TYPE = C["VIC_GF_410"]
    o.TFALLBACK = 1
    o.AERO_RESIST_CANSNOW = C["VIC_AR_406_FULL"]
    o.SNOW_ALBEDO = C["VIC_SNOW_ALBEDO_USACE"]
    o.SNOW_DENSITY = C["VIC_DENS_BRAS"]
    o.TEMP_TH_TYPE = C["VIC_TEMP_TH_KIENZLE"]
    o.GLACIER_ID = -1
    o.GLACIER_DYNAMICS = 0
    o.frozen_compat = 0
    o.nveg_types = 0
    o.wind_h = 10.0
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    # get_global_param.c:376-381: FROZEN_SOIL forces QUICK_FLUX FALSE; :1151-1155 QUICK_FLUX forces Nnode 3
    if o.FROZEN_SOIL and "QUICK_FLUX" not in kw:
        o.QUICK_FLUX = 0
    if o.QUICK_FLUX:
        o.Nnode = 3
    return o
