"""Initial HRU state tables without a state file: host-side restatement of
initialize_model_state.c:8-760 (+ initialize_snow.c, initialize_soil.c, initialize_veg.c)
for the supported option subset, vectorised over HRUs with numpy.

This is ingest logic that runs once before the time loop (reference layer L3, outside
the hot path); tests/test_domain.py checks it against the reference build.
"""
import numpy as np

from . import abi
from .abi import C

STEFAN_B = 5.6696e-8
KELVIN = 273.15
LF = 3.337e5
INVALID_INT = -2147483648


def maximum_unfrozen_water(T, max_moist, bubble, expt):
    """soil_conduction.c:830-863."""
    with np.errstate(all="ignore"):
        u = max_moist * np.power((-LF * T) / 273.16 / (9.81 * bubble / 100.), -(2.0 / (expt - 3.0)))
    u = np.where(u > max_moist, max_moist, u)
    u = np.where(u < 0, 0, u)
    return np.where(T <= 0, u, max_moist)


def soil_conductivity(moist, Wu, soil_dens_min, bulk_dens_min, quartz, soil_density, bulk_density, organic):
    """soil_conduction.c:7-105."""
    Ki, Kw, Kdry_org, Ks_org = 2.2, 0.57, 0.05, 0.25
    Kdry_min = (0.135 * bulk_dens_min + 64.7) / (soil_dens_min - 0.947 * bulk_dens_min)
    Kdry = (1 - organic) * Kdry_min + organic * Kdry_org
    porosity = 1.0 - bulk_density / soil_density
    with np.errstate(all="ignore"):
        Sr = moist / porosity
        Ks_min = np.where(quartz < .2, np.power(7.7, quartz) * np.power(3.0, 1.0 - quartz),
                          np.power(7.7, quartz) * np.power(2.2, 1.0 - quartz))
        Ks = (1 - organic) * Ks_min + organic * Ks_org
        unfrozen = Wu == moist
        Ksat_u = np.power(Ks, 1.0 - porosity) * np.power(Kw, porosity)
        Ke_u = 0.7 * np.log10(Sr) + 1.0
        Ksat_f = np.power(Ks, 1.0 - porosity) * np.power(Ki, porosity - Wu) * np.power(Kw, Wu)
        Ksat = np.where(unfrozen, Ksat_u, Ksat_f)
        Ke = np.where(unfrozen, Ke_u, Sr)
        K = (Ksat - Kdry) * Ke + Kdry
    K = np.where(K < Kdry, Kdry, K)
    return np.where(moist > 0., K, Kdry)


def volumetric_heat_capacity(soil_fract, water_fract, ice_fract, organic_fract):
    """soil_conduction.c:108-139."""
    Cs = 2.0e6 * soil_fract * (1 - organic_fract)
    Cs = Cs + 2.7e6 * soil_fract * organic_fract
    Cs = Cs + 4.2e6 * water_fract
    Cs = Cs + 1.9e6 * ice_fract
    Cs = Cs + 1.3e3 * (1. - (soil_fract + water_fract + ice_fract))
    return Cs


def _layer(cp, f, cell):
    return np.stack([cp[abi.cp_layer(C[f], l)][cell] for l in range(3)])


def _node(cp, f, cell, Nn):
    return np.stack([cp[abi.cp_node(C[f], n, Nn)][cell] for n in range(Nn)])


def distribute_node_moisture_properties(opt, cp, cell, T, moist):
    """soil_conduction.c:304-440 for every HRU. T [Nn][nhru], moist [3][nhru] -> (moist_n, ice_n, kappa_n, Cs_n)."""
    Nn = opt.Nnode
    nh = T.shape[1]
    depth = _layer(cp, "CPL_DEPTH", cell)
    bd = _layer(cp, "CPL_BULK_DENSITY", cell); sd = _layer(cp, "CPL_SOIL_DENSITY", cell); org = _layer(cp, "CPL_ORGANIC", cell)
    sdm = _layer(cp, "CPL_SOIL_DENS_MIN", cell); bdm = _layer(cp, "CPL_BULK_DENS_MIN", cell); qz = _layer(cp, "CPL_QUARTZ", cell)
    Z = _node(cp, "CPN_ZSUM", cell, Nn); mmn = _node(cp, "CPN_MAX_MOIST", cell, Nn)
    bub = _node(cp, "CPN_BUBBLE", cell, Nn); ex = _node(cp, "CPN_EXPT", cell, Nn)
    fs = (cp[C["CP_FS_ACTIVE"]][cell] != 0) & bool(opt.FROZEN_SOIL)
    hid = np.arange(nh)
    l = np.zeros(nh, dtype=int); Lsum = np.zeros(nh); past = np.zeros(nh, dtype=bool)
    mo = np.zeros((Nn, nh)); ic = np.zeros((Nn, nh)); ka = np.zeros((Nn, nh)); cs = np.zeros((Nn, nh))
    for n in range(Nn):
        dl = depth[l, hid]
        nxt = np.minimum(l + 1, 2)
        onb = (Z[n] == Lsum + dl) & (n != 0) & (l != 2)
        m = np.where(onb, (moist[l, hid] / dl + moist[nxt, hid] / depth[nxt, hid]) / 1000 / 2., moist[l, hid] / dl / 1000)
        m = np.where(m - mmn[n] > 0, mmn[n], m)
        frozen = (T[n] < 0) & fs
        i = m - maximum_unfrozen_water(T[n], mmn[n], bub[n], ex[n])
        i = np.where(i < 0, 0, i)
        i = np.where(frozen, i, 0.0)
        k = soil_conductivity(m, m - i, sdm[l, hid], bdm[l, hid], qz[l, hid], sd[l, hid], bd[l, hid], org[l, hid])
        c = volumetric_heat_capacity(bd[l, hid] / sd[l, hid], m - i, i, org[l, hid])
        mo[n], ic[n], ka[n], cs[n] = m, i, k, c
        adv = (Z[n] > Lsum + dl) & ~past
        Lsum = np.where(adv, Lsum + dl, Lsum)
        l2 = np.where(adv, l + 1, l)
        hit = adv & (l2 == 3)
        past = past | hit
        l = np.where(hit, 2, l2)
    return mo, ic, ka, cs


def estimate_layer_ice_content(opt, cp, cell, T, moist):
    """soil_conduction.c:444-614 (one frost area): layer ice (mm) and layer T from node T by trapezoids over the node
    segments clipped to each layer."""
    Nn = opt.Nnode
    nh = T.shape[1]
    depth = _layer(cp, "CPL_DEPTH", cell); mm = _layer(cp, "CPL_MAX_MOIST", cell)
    bub = _layer(cp, "CPL_BUBBLE", cell); ex = _layer(cp, "CPL_EXPT", cell)
    Z = _node(cp, "CPN_ZSUM", cell, Nn)
    fs = (cp[C["CP_FS_ACTIVE"]][cell] != 0) & bool(opt.FROZEN_SOIL)
    Lsum = np.concatenate([np.zeros((1, nh)), np.cumsum(depth, axis=0)])
    Lsum[2] = Lsum[1] + depth[1]; Lsum[3] = Lsum[2] + depth[2]
    ice = np.zeros((3, nh)); LT = np.zeros((3, nh))
    for l in range(3):
        accI = np.zeros(nh); accT = np.zeros(nh)
        for n in range(Nn - 1):
            lo = np.maximum(Z[n], Lsum[l]); hi = np.minimum(Z[n + 1], Lsum[l + 1])
            on = hi > lo
            with np.errstate(all="ignore"):
                Tlo = np.where(Z[n] < Lsum[l], (Lsum[l] - Z[n]) / (Z[n + 1] - Z[n]) * (T[n + 1] - T[n]) + T[n], T[n])
                Thi = np.where(Z[n + 1] > Lsum[l + 1], (Lsum[l + 1] - Z[n]) / (Z[n + 1] - Z[n]) * (T[n + 1] - T[n]) + T[n], T[n + 1])
            Ilo = moist[l] - maximum_unfrozen_water(Tlo, mm[l], bub[l], ex[l]); Ilo = np.where(Ilo < 0, 0, Ilo)
            Ihi = moist[l] - maximum_unfrozen_water(Thi, mm[l], bub[l], ex[l]); Ihi = np.where(Ihi < 0, 0, Ihi)
            Ilo = np.where(fs, Ilo, 0.0); Ihi = np.where(fs, Ihi, 0.0)
            accI = np.where(on, accI + (hi - lo) * (Ihi + Ilo) / 2., accI)
            accT = np.where(on, accT + (hi - lo) * (Thi + Tlo) / 2., accT)
        ice[l] = accI / depth[l]
        LT[l] = accT / depth[l]
    return ice, LT


def estimate_layer_ice_content_quick_flux(opt, cp, cell, Tsurf, T1, moist):
    """soil_conduction.c:617-723."""
    depth = _layer(cp, "CPL_DEPTH", cell); mm = _layer(cp, "CPL_MAX_MOIST", cell)
    bub = _layer(cp, "CPL_BUBBLE", cell); ex = _layer(cp, "CPL_EXPT", cell)
    avg_temp = cp[C["CP_AVG_TEMP"]][cell]; dp = cp[C["CP_DP"]][cell]
    fs = (cp[C["CP_FS_ACTIVE"]][cell] != 0) & bool(opt.FROZEN_SOIL)
    L = [np.zeros_like(dp)]
    for l in range(3):
        L.append(depth[l] + L[-1])
    LT = np.zeros((3, len(dp)))
    LT[0] = 0.5 * (Tsurf + T1)
    for l in (1, 2):
        LT[l] = avg_temp - dp / depth[l] * (T1 - avg_temp) * (np.exp(-(L[l + 1] - L[1]) / dp) - np.exp(-(L[l] - L[1]) / dp))
    ice = np.zeros((3, len(dp)))
    for l in range(3):
        i = moist[l] - maximum_unfrozen_water(LT[l], mm[l], bub[l], ex[l])
        i = np.clip(i, 0, moist[l])
        ice[l] = np.where(fs, i, 0.0)
    return ice, LT


def initial_state(dom, forcing0):
    """State tables after initialize_model_state.  forcing0 = first record [VIC_NFORCE][NF+1][ncell]."""
    opt = dom.opt
    Nn, Nb = opt.Nnode, opt.Nband
    cp = dom.cell_params
    nh = dom.nhru
    cell = dom.hru_iparams[C["HPI_CELL"]]
    band = dom.hru_iparams[C["HPI_BAND"]]
    sd = np.zeros((abi.sd_nrow(Nn), nh))
    si = np.zeros((abi.si_nrow(Nn), nh), dtype=np.int32)

    Tair = forcing0[C["VIC_F_AIR_TEMP"], opt.NR][cell]
    surf_temp = np.where(Tair < -1., -1., Tair)                      # initialize_model_state.c:147
    depth = _layer(cp, "CPL_DEPTH", cell)
    max_moist = _layer(cp, "CPL_MAX_MOIST", cell)
    moist = np.minimum(dom.init_moist[:, cell], max_moist)           # initialize_soil.c:44-46
    avg_temp = cp[C["CP_AVG_TEMP"]][cell]; dp = cp[C["CP_DP"]][cell]
    Z = _node(cp, "CPN_ZSUM", cell, Nn)
    T = np.zeros((Nn, nh))
    if opt.QUICK_FLUX:
        T[0] = surf_temp; T[1] = surf_temp; T[2] = avg_temp          # :517-523
    else:
        def exp_interp(x, lx, ux, ly, uy):                           # modify_Ksat.c:11-13
            return uy + (ly - uy) * np.exp(-(x - lx))
        T[0] = surf_temp
        T[Nn - 1] = avg_temp
        T[1] = exp_interp(depth[0], 0., dp, surf_temp, avg_temp)
        T[2] = exp_interp(2. * depth[0], 0., dp, surf_temp, avg_temp)
        for n in range(3, Nn - 1):
            T[n] = exp_interp(Z[n], 0., dp, surf_temp, avg_temp)
    mo, ic, ka, cs = distribute_node_moisture_properties(opt, cp, cell, T, moist)
    if opt.QUICK_FLUX:
        lice, lT = estimate_layer_ice_content_quick_flux(opt, cp, cell, T[0], T[1], moist)
    else:
        lice, lT = estimate_layer_ice_content(opt, cp, cell, T, moist)
    for l in range(3):
        sd[C["SD_MOIST0"] + l] = moist[l]
        sd[C["SD_ICE0"] + l] = lice[l]
        sd[C["SD_LAYER_T0"] + l] = lT[l]
    for n in range(Nn):
        sd[abi.sd_node(C["SDN_T"], n, Nn)] = T[n]
        sd[abi.sd_node(C["SDN_MOIST"], n, Nn)] = mo[n]
        sd[abi.sd_node(C["SDN_ICE"], n, Nn)] = ic[n]
        sd[abi.sd_node(C["SDN_KAPPA"], n, Nn)] = ka[n]
        sd[abi.sd_node(C["SDN_CS"], n, Nn)] = cs[n]
    # LongUnderOut is computed from energy.T[0] BEFORE the node temperatures are assigned (:283-284), i.e. from 0 C
    sd[C["SD_LONGUNDEROUT"]] = STEFAN_B * (0.0 + KELVIN) ** 4
    tf = np.stack([cp[abi.cp_band(C["CPB_TFACTOR"], b, Nn, Nb)] for b in range(Nb)])
    sd[C["SD_TFOLIAGE"]] = Tair + tf[band, cell]
    sd[C["SD_GLAC_CUM_MASS_BALANCE"]] = np.nan                       # glac_data_struct ctor, vicNl_def.h:1344-1348
    si[C["SI_SNOW_LAST_SNOW"]] = INVALID_INT                         # initialize_snow.c
    if (not opt.QUICK_FLUX):
        # find_0_degree_fronts counts (soil_conduction.c:775-828) for FS_ACTIVE cells
        fs = (cp[C["CP_FS_ACTIVE"]][cell] != 0)
        nthaw = np.zeros(nh, dtype=np.int32); nfrost = np.zeros(nh, dtype=np.int32)
        for n in range(Nn - 2, -1, -1):
            th = (T[n] > 0) & (T[n + 1] <= 0) & (nthaw < 3)
            fr = ~th & (T[n] < 0) & (T[n + 1] >= 0) & (nfrost < 3)
            nthaw += th; nfrost += fr
        si[C["SI_NTHAW"]] = np.where(fs, nthaw, 0)
        si[C["SI_NFROST"]] = np.where(fs, nfrost, 0)
    return sd, si
