"""Synthetic regular-grid domains and forcings packed into the
struct-of-arrays tables of include/vicgpu.h.

This is host-side ingest logic (the reference's L3, out of the hot path): it
restates what read_soilparam.c / read_veglib.c / read_snowband.c /
initialize_atmos.c derive from their input files for the fields the path reads,
with a citation at each derivation.  tests/test_domain.py checks the node /
water-table tables against the reference build when it is present.
"""
import numpy as np

from . import abi
from .abi import C

LAI_WATER_FACTOR = 0.2           # user_def.h:111
SEED = 20261003

_MONTH_DAYS = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])


def make_veglib(glacier=False):
    """Veg library rows (read_veglib.c:44-137) + the 4 reference PET surfaces (global.h:49-62).

    Classes: 1 = grass (no overstory), 2 = conifer (overstory), 22 = glacier (no overstory, LAI 0).
    Returns (table[nrow][VL_NFIELD], nveg_types).
    """
    def row(veg_class, overstory, rarc, rmin, lai, albedo, rough, displ, wind_h, rgl, rad_atten, wind_atten, trunk):
        r = np.zeros(C["VL_NFIELD"])
        r[C["VL_OVERSTORY"]] = overstory
        r[C["VL_RARC"]] = rarc
        r[C["VL_RMIN"]] = rmin
        r[C["VL_RAD_ATTEN"]] = rad_atten
        r[C["VL_TRUNK_RATIO"]] = trunk
        r[C["VL_WIND_ATTEN"]] = wind_atten
        r[C["VL_WIND_H"]] = wind_h
        r[C["VL_RGL"]] = np.float32(rgl)
        r[C["VL_VEG_CLASS"]] = veg_class
        lai = np.broadcast_to(np.asarray(lai, dtype=float), (12,))
        r[C["VL_LAI"]:C["VL_LAI"] + 12] = lai
        r[C["VL_WDMAX"]:C["VL_WDMAX"] + 12] = LAI_WATER_FACTOR * lai      # read_veglib.c:64
        r[C["VL_ALBEDO"]:C["VL_ALBEDO"] + 12] = albedo
        r[C["VL_DISPLACEMENT"]:C["VL_DISPLACEMENT"] + 12] = displ
        r[C["VL_EMISSIVITY"]:C["VL_EMISSIVITY"] + 12] = 0.0
        r[C["VL_ROUGHNESS"]:C["VL_ROUGHNESS"] + 12] = rough
        return r

    grass_lai = [0.5, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 2.5, 2.0, 1.25, 0.75, 0.5]
    conifer_lai = [3.4, 3.4, 3.5, 3.7, 4.0, 4.4, 4.4, 4.3, 4.2, 3.7, 3.5, 3.4]
    rows = [
        row(1, 0, 2.0, 120.0, grass_lai, 0.20, 0.0738, 0.402, 2.0, 100.0, 0.5, 0.5, 0.2),
        row(2, 1, 60.0, 250.0, conifer_lai, 0.12, 1.476, 8.04, 40.0, 30.0, 0.5, 0.5, 0.2),
    ]
    if glacier:
        rows.append(row(22, 0, 100.0, 0.0, 0.0, 0.30, 0.002, 0.0, 2.0, 0.0, 0.0, 0.0, 0.0))
    nveg = len(rows)
    # reference PET surfaces, global.h:49-62 via read_veglib.c:118-136
    ref = dict(over=[0, 0, 0, 0], rarc=[0.0, 0.0, 25, 25], rmin=[0.0, 0.0, 100, 100], lai=[1.0, 1.0, 2.88, 4.45],
               albedo=[0.2, 0.08, 0.23, 0.23], rough=[0.001, 0.001, 0.0148, 0.0615],
               displ=[0.0054, 0.0054, 0.08, 0.3333], wind_h=[10.0] * 4, RGL=[0.0, 0.0, 100, 100])
    for i in range(4):
        rows.append(row(nveg + i + 1, ref["over"][i], ref["rarc"][i], ref["rmin"][i], ref["lai"][i], ref["albedo"][i],
                        ref["rough"][i], ref["displ"][i], ref["wind_h"][i], ref["RGL"][i], 0.0, 0.0, 0.0))
    return np.ascontiguousarray(np.stack(rows)), nveg


def zwt_tables(depth, expt, bubble, max_moist, resid_moist):
    """zwtvmoist_zwt / zwtvmoist_moist curves, restating read_soilparam.c:1188-1284.

    Inputs are [3][ncell]; returns (zwt[5][11][ncell], moist[5][11][ncell]).
    """
    nl, nc = depth.shape
    NZ = abi.VIC_MAX_ZWTVMOIST
    zwt = np.zeros((nl + 2, NZ, nc))
    mst = np.zeros((nl + 2, NZ, nc))
    with np.errstate(all="ignore"):
        # individual layers (:1189-1206)
        tmp_depth = np.zeros(nc)
        for l in range(nl):
            b = 0.5 * (expt[l] - 3)
            bub = bubble[l]
            resid = resid_moist[l] * depth[l] * 1000
            zp = np.zeros(nc)
            for i in range(NZ):
                zwt[l, i] = -tmp_depth * 100 - zp
                w = (depth[l] * 100 - zp - (b / (b - 1)) * bub * (1 - np.power((zp + bub) / bub, (b - 1) / b))) / (depth[l] * 100)
                w = np.clip(w, 0, 1)
                mst[l, i] = w * (max_moist[l] - resid) + resid
                zp = zp + depth[l] * 100 / (NZ - 1)
            tmp_depth = tmp_depth + depth[l]
        # top N-1 layers lumped (:1208-1233)
        tmp_depth = np.zeros(nc); b = np.zeros(nc); bub = np.zeros(nc); tmm = np.zeros(nc); trm = np.zeros(nc)
        for l in range(nl - 1):
            b = b + 0.5 * (expt[l] - 3) * depth[l]
            bub = bub + bubble[l] * depth[l]
            tmm = tmm + max_moist[l]
            trm = trm + resid_moist[l] * depth[l] * 1000
            tmp_depth = tmp_depth + depth[l]
        b = b / tmp_depth
        bub = bub / tmp_depth
        zp = np.zeros(nc)
        for i in range(NZ):
            zwt[nl, i] = -zp
            w = (tmp_depth * 100 - zp - (b / (b - 1)) * bub * (1 - np.power((zp + bub) / bub, (b - 1) / b))) / (tmp_depth * 100)
            w = np.clip(w, 0, 1)
            mst[nl, i] = w * (tmm - trm) + trm
            zp = zp + tmp_depth * 100 / (NZ - 1)
        # whole column filled from the bottom up (:1235-1284), vectorised over cells with masks for the layer walks
        tot = depth.sum(axis=0)
        cidx = np.arange(nc)
        zp = np.zeros(nc)
        for i in range(NZ):
            zwt[nl + 1, i] = -zp
            if i == 0:
                mst[nl + 1, i] = max_moist.sum(axis=0)
            else:
                tm = np.zeros(nc)
                l = np.full(nc, nl - 1)
                td2 = tot - depth[nl - 1]
                for _ in range(nl - 1):
                    go = (l > 0) & (zp <= td2 * 100)
                    tm = np.where(go, tm + max_moist[l, cidx], tm)
                    l = np.where(go, l - 1, l)
                    td2 = np.where(go, td2 - depth[l, cidx], td2)
                dl = depth[l, cidx]
                w = (td2 * 100 + dl * 100 - zp) / (dl * 100)
                b = 0.5 * (expt[l, cidx] - 3)
                bub = bubble[l, cidx]
                resid = resid_moist[l, cidx] * dl * 1000
                w = w + (-(b / (b - 1)) * bub * (1 - np.power((zp + bub - td2 * 100) / bub, (b - 1) / b)) / (dl * 100))
                tm = tm + (w * (max_moist[l, cidx] - resid) + resid)
                b_save, bub_save, td2_save = b, bub, td2
                for _ in range(nl - 1):
                    go = l > 0
                    l2 = np.where(go, l - 1, l)
                    dl = depth[l2, cidx]
                    td2n = td2 - dl
                    b = 0.5 * (expt[l2, cidx] - 3)
                    bub = bubble[l2, cidx]
                    resid = resid_moist[l2, cidx] * dl * 1000
                    zpe = td2_save * 100 - bub + bub * np.power((zp + bub_save - td2_save * 100) / bub_save, b / b_save)
                    w = -(b / (b - 1)) * bub * (1 - np.power((zpe + bub - td2n * 100) / bub, (b - 1) / b)) / (dl * 100)
                    tm = np.where(go, tm + (w * (max_moist[l2, cidx] - resid) + resid), tm)
                    b_save = np.where(go, b, b_save); bub_save = np.where(go, bub, bub_save)
                    td2_save = np.where(go, td2n, td2_save)
                    td2 = np.where(go, td2n, td2)
                    l = l2
                mst[nl + 1, i] = tm
            zp = zp + tot * 100 / (NZ - 1)
    return zwt, mst


def node_geometry(opt, depth, dp):
    """Thermal node depths/thicknesses: initialize_model_state.c:505-512 (QUICK_FLUX) and
    :545-586 (finite difference, EXP_TRANS FALSE).  depth [3][ncell], dp [ncell] -> Zsum[Nn][ncell], dz[Nn][ncell]."""
    Nn = opt.Nnode
    nc = depth.shape[1]
    Z = np.zeros((Nn, nc)); dz = np.zeros((Nn, nc))
    d0 = depth[0]
    if opt.QUICK_FLUX:
        dz[0] = d0; dz[1] = d0; dz[2] = 2. * (dp - 1.5 * d0)
        Z[0] = 0; Z[1] = d0; Z[2] = dp
    elif not opt.EXP_TRANS:
        dz[0] = d0; dz[1] = d0; dz[2] = d0
        Z[0] = 0; Z[1] = d0
        Zsum = 2. * d0
        Z[2] = Zsum
        tmpdp = dp - d0 * 2.5
        for i in range(3, Nn - 1):
            dz[i] = tmpdp / (float(Nn) - 3.5)
            Zsum = Zsum + (dz[i] + dz[i - 1]) / 2.
            Z[i] = Zsum
        dz[Nn - 1] = (dp - Zsum - dz[Nn - 2] / 2.) * 2.
        Zsum = Zsum + (dz[Nn - 2] + dz[Nn - 1]) / 2.
        Z[Nn - 1] = Zsum
    else:
        Bexp = np.log(dp + 1.) / float(Nn - 1)
        for i in range(Nn):
            Z[i] = np.exp(Bexp * i) - 1.
        dz[0] = Z[1] - Z[0]
        for i in range(1, Nn - 1):
            dz[i] = (Z[i + 1] - Z[i]) / 2. + (Z[i] - Z[i - 1]) / 2.
        dz[Nn - 1] = Z[Nn - 1] - Z[Nn - 2]
    return Z, dz


def node_parameters(opt, Z, depth, max_moist, expt, bubble):
    """set_node_parameters (soil_conduction.c:225-275): per-node max_moist (mm/mm), expt, bubble, alpha/beta/gamma."""
    Nn = opt.Nnode
    nc = depth.shape[1]
    mm = np.zeros((Nn, nc)); ex = np.zeros((Nn, nc)); bu = np.zeros((Nn, nc))
    alpha = np.zeros((Nn, nc)); beta = np.zeros((Nn, nc)); gamma = np.zeros((Nn, nc))
    cidx = np.arange(nc)
    lidx = np.zeros(nc, dtype=int)
    Lsum = np.zeros(nc)
    past = np.zeros(nc, dtype=bool)
    for n in range(Nn):
        dl = depth[lidx, cidx]
        nxt = np.minimum(lidx + 1, 2)
        onb = (Z[n] == Lsum + dl) & (n != 0) & (lidx != 2)
        mm[n] = np.where(onb, (max_moist[lidx, cidx] / dl + max_moist[nxt, cidx] / depth[nxt, cidx]) / 1000 / 2.,
                         max_moist[lidx, cidx] / dl / 1000)
        ex[n] = np.where(onb, (expt[lidx, cidx] + expt[nxt, cidx]) / 2., expt[lidx, cidx])
        bu[n] = np.where(onb, (bubble[lidx, cidx] + bubble[nxt, cidx]) / 2., bubble[lidx, cidx])
        adv = (Z[n] > Lsum + dl) & ~past
        Lsum = np.where(adv, Lsum + dl, Lsum)
        l2 = np.where(adv, lidx + 1, lidx)
        hit = adv & (l2 == 3)
        past = past | hit
        lidx = np.where(hit, 2, l2)
    for n in range(Nn - 2):
        alpha[n] = Z[n + 2] - Z[n]
        beta[n] = Z[n + 1] - Z[n]
        gamma[n] = Z[n + 2] - Z[n + 1]
    if opt.NOFLUX:
        alpha[Nn - 2] = 2. * (Z[Nn - 1] - Z[Nn - 2])
        beta[Nn - 2] = Z[Nn - 1] - Z[Nn - 2]
        gamma[Nn - 2] = Z[Nn - 1] - Z[Nn - 2]
    return mm, ex, bu, alpha, beta, gamma


class Domain:
    """All host-side tables of one synthetic domain."""
    pass


def make_domain(ncell, opt, ntile=1, tile_classes=None, glacier_top_band=False, seed=SEED, band_spread=400.0, bare_fraction=0.0,
                cell_range=None):
    """Regular grid of `ncell` cells, opt.Nband snow bands x ntile veg tiles per band.

    HRU numbering is slot-major: hru = slot * ncell + cell with slot = tile * Nband + band, so a
    wavefront of consecutive HRUs covers consecutive cells of the same (tile, band) — coalesced
    forcing / parameter loads and a uniform vegetation class per wavefront.

    bare_fraction > 0 leaves that fraction of every cell without vegetation, which read_vegparam.c:312-340 fills with one
    "artificial" bare-soil HRU per band (vegIndex = num_veg_types, Cv = (1 - Cv_sum) / Nband, no root zones); those HRUs
    take the last slots.

    cell_range = (c0, c1) builds only cells [c0, c1) of the `ncell`-cell domain (one rank's shard of a multi-GPU run:
    the per-cell random draws are made for the whole domain, everything derived from them only for the shard), with
    exactly the tables shard.shard_domain would cut out of the full domain.
    """
    rng = np.random.default_rng(seed)
    Nn, Nb = opt.Nnode, opt.Nband
    d = Domain()
    d.opt = opt
    ncell_global = ncell
    c0, c1 = (0, ncell) if cell_range is None else (int(cell_range[0]), int(cell_range[1]))
    assert 0 <= c0 < c1 <= ncell_global
    sl = slice(c0, c1)
    d.global_cell0 = c0
    d.ncell_global = ncell_global
    veglib, nveg = make_veglib(glacier=glacier_top_band)
    opt.nveg_types = nveg
    if glacier_top_band:
        opt.GLACIER_ID = 22
    d.veglib = veglib

    # the per-cell draws, in a fixed order, for the whole domain; the shard's slice of each
    ng = ncell_global
    u_d1 = rng.uniform(0.2, 0.5, ng)[sl]; u_d2 = rng.uniform(0.8, 2.0, ng)[sl]
    b_infilt = rng.uniform(0.05, 0.4, ng)[sl]
    Ds = rng.uniform(0.001, 0.3, ng)[sl]
    Dsmax = rng.uniform(2, 30, ng)[sl]
    Ws = rng.uniform(0.5, 0.95, ng)[sl]
    u_expt = rng.uniform(8, 16, ng)[sl]
    u_ksat = rng.uniform(100, 2000, ng)[sl]
    u_quartz = rng.uniform(0.2, 0.8, ng)[sl]
    u_bulk = rng.uniform(1400, 1600, ng)[sl]
    avg_temp = rng.uniform(-3, 8, ng)[sl]
    elevation = np.float32(rng.uniform(500, 2500, ng)).astype(float)[sl]
    cell_offset_T = rng.uniform(-3, 3, ng)[sl]
    ncell = c1 - c0
    d.ncell = ncell

    cp = np.zeros((abi.cp_nrow(Nn, Nb), ncell))
    # layer depths rounded to mm like read_soilparam.c's (float)(int)(x*1000+0.5)/1000
    depth = np.stack([np.full(ncell, 0.1), u_d1, u_d2])
    depth = np.floor(depth * 1000 + 0.5) / 1000
    expt = np.tile(u_expt, (3, 1))
    Ksat = np.tile(u_ksat, (3, 1))
    bubble = 0.32 * expt + 4.3
    quartz = np.tile(u_quartz, (3, 1))
    bulk = np.tile(u_bulk, (3, 1))
    soil_dens = np.full((3, ncell), 2650.0)
    organic = np.zeros((3, ncell))
    porosity = 1.0 - bulk / soil_dens                      # read_soilparam.c:900
    max_moist = depth * porosity * 1000.                   # :902
    Wcr = 0.7 * max_moist                                  # :1029
    Wpwp = 0.3 * max_moist                                 # :1030
    resid = np.full((3, ncell), 0.02)
    dp = np.full(ncell, 4.0)
    nlat = int(np.ceil(np.sqrt(ncell_global)))
    lat = np.float32(45.0 + 0.0625 * (np.arange(c0, c1) // nlat)).astype(float)

    cp[C["CP_DS"]] = Ds; cp[C["CP_DSMAX"]] = Dsmax; cp[C["CP_WS"]] = Ws; cp[C["CP_C"]] = 2.0
    cp[C["CP_B_INFILT"]] = b_infilt; cp[C["CP_DP"]] = dp; cp[C["CP_AVG_TEMP"]] = avg_temp
    cp[C["CP_ROUGH"]] = 0.01; cp[C["CP_SNOW_ROUGH"]] = 0.0005
    cp[C["CP_ELEVATION"]] = elevation; cp[C["CP_LAT"]] = lat
    cp[C["CP_FS_ACTIVE"]] = 1.0 if opt.FROZEN_SOIL else 0.0
    cp[C["CP_NEW_SNOW_ALB"]] = 0.85
    cp[C["CP_SNOW_ALB_ACCUM_A"]] = 0.94; cp[C["CP_SNOW_ALB_ACCUM_B"]] = 0.58
    cp[C["CP_SNOW_ALB_THAW_A"]] = 0.82; cp[C["CP_SNOW_ALB_THAW_B"]] = 0.46
    cp[C["CP_MIN_RAIN_TEMP"]] = 1.0    # KIENZLE TT (calc_rainonly.c:73)
    cp[C["CP_MAX_SNOW_TEMP"]] = 3.0    # KIENZLE TR (calc_rainonly.c:74)
    cp[C["CP_PADJ_R"]] = 1.0; cp[C["CP_PADJ_S"]] = 1.0
    cp[C["CP_GLAC_SURF_THICK"]] = 100.0; cp[C["CP_GLAC_SURF_WE"]] = 91.7
    cp[C["CP_GLAC_KMIN"]] = 0.05; cp[C["CP_GLAC_DK"]] = 0.75; cp[C["CP_GLAC_A"]] = 1.0
    cp[C["CP_GLAC_ALBEDO"]] = 0.3; cp[C["CP_GLAC_ROUGH"]] = 0.002
    for l in range(3):
        cp[abi.cp_layer(C["CPL_KSAT"], l)] = Ksat[l]
        cp[abi.cp_layer(C["CPL_WCR"], l)] = Wcr[l]
        cp[abi.cp_layer(C["CPL_WPWP"], l)] = Wpwp[l]
        cp[abi.cp_layer(C["CPL_EXPT"], l)] = expt[l]
        cp[abi.cp_layer(C["CPL_BUBBLE"], l)] = bubble[l]
        cp[abi.cp_layer(C["CPL_DEPTH"], l)] = depth[l]
        cp[abi.cp_layer(C["CPL_MAX_MOIST"], l)] = max_moist[l]
        cp[abi.cp_layer(C["CPL_RESID_MOIST"], l)] = resid[l]
        cp[abi.cp_layer(C["CPL_POROSITY"], l)] = porosity[l]
        cp[abi.cp_layer(C["CPL_QUARTZ"], l)] = quartz[l]
        cp[abi.cp_layer(C["CPL_ORGANIC"], l)] = organic[l]
        cp[abi.cp_layer(C["CPL_BULK_DENSITY"], l)] = bulk[l]
        cp[abi.cp_layer(C["CPL_SOIL_DENSITY"], l)] = soil_dens[l]
        cp[abi.cp_layer(C["CPL_BULK_DENS_MIN"], l)] = bulk[l]
        cp[abi.cp_layer(C["CPL_SOIL_DENS_MIN"], l)] = soil_dens[l]
    Z, dz = node_geometry(opt, depth, dp)
    mm, ex, bu, alpha, beta, gamma = node_parameters(opt, Z, depth, max_moist, expt, bubble)
    for n in range(Nn):
        cp[abi.cp_node(C["CPN_ZSUM"], n, Nn)] = Z[n]
        cp[abi.cp_node(C["CPN_DZ"], n, Nn)] = dz[n]
        cp[abi.cp_node(C["CPN_ALPHA"], n, Nn)] = alpha[n]
        cp[abi.cp_node(C["CPN_BETA"], n, Nn)] = beta[n]
        cp[abi.cp_node(C["CPN_GAMMA"], n, Nn)] = gamma[n]
        cp[abi.cp_node(C["CPN_MAX_MOIST"], n, Nn)] = mm[n]
        cp[abi.cp_node(C["CPN_EXPT"], n, Nn)] = ex[n]
        cp[abi.cp_node(C["CPN_BUBBLE"], n, Nn)] = bu[n]
    # snow bands: read_snowband.c:91-114 with T_LAPSE 6.5 C/km, PGRAD 0
    T_LAPSE = 6.5
    area = np.full((Nb, ncell), 1.0 / Nb)
    if Nb > 1:
        offs = np.linspace(-band_spread, band_spread, Nb)
    else:
        offs = np.zeros(1)
    for b in range(Nb):
        be = np.float32(elevation + offs[b]).astype(float)
        cp[abi.cp_band(C["CPB_AREAFRACT"], b, Nn, Nb)] = area[b]
        cp[abi.cp_band(C["CPB_TFACTOR"], b, Nn, Nb)] = (elevation - be) / 1000. * T_LAPSE
        cp[abi.cp_band(C["CPB_PFACTOR"], b, Nn, Nb)] = 1.0
        cp[abi.cp_band(C["CPB_BANDELEV"], b, Nn, Nb)] = be
        cp[abi.cp_band(C["CPB_ABOVETREELINE"], b, Nn, Nb)] = 0.0
    zwt, zm = zwt_tables(depth, expt, bubble, max_moist, resid)
    for l in range(5):
        for i in range(abi.VIC_MAX_ZWTVMOIST):
            cp[abi.cp_zwt_zwt(l, i, Nn, Nb)] = zwt[l, i]
            cp[abi.cp_zwt_moist(l, i, Nn, Nb)] = zm[l, i]
    d.cell_params = np.ascontiguousarray(cp)
    d.init_moist = np.ascontiguousarray(0.6 * max_moist)

    # HRUs
    if tile_classes is None:
        tile_classes = [0, 1, 0, 1, 0][:ntile] if ntile > 1 else [0]
    assert len(tile_classes) == ntile
    nbare = 1 if bare_fraction > 0 else 0
    nslot = (ntile + nbare) * Nb
    nhru = nslot * ncell
    d.nhru = nhru
    d.nslot = nslot
    hpi = np.zeros((C["HPI_NROW"], nhru), dtype=np.int32)
    hpd = np.zeros((C["HPD_NROW"], nhru))
    # veg_con.sigma_slope / lag_one / fetch (blowing snow only); float32 values like the reference's vegetation file gives
    hpd[C["HPD_SIGMA_SLOPE"]] = np.float32(0.08); hpd[C["HPD_LAG_ONE"]] = np.float32(0.95); hpd[C["HPD_FETCH"]] = np.float32(1000.0)
    tile_frac = np.full(ntile, (1.0 - bare_fraction) / ntile)
    roots = {0: (0.10, 0.70, 0.20), 1: (0.10, 0.60, 0.30), 2: (0.0, 0.0, 0.0)}
    cells = np.arange(ncell)
    for k in range(ntile):
        for b in range(Nb):
            slot = k * Nb + b
            g = slot * ncell + cells
            vidx = tile_classes[k]
            # glacier_top_band: True = tile 0 of the top band; "all" = tile 0 of every band (and tile 1 of the top band too:
            # two glacier HRUs at one elevation, which GlacierMassBalanceResult.c:40-47 merges into one point)
            if glacier_top_band and k == 0 and (b == Nb - 1 or glacier_top_band == "all"):
                vidx = 2
            if glacier_top_band == "all" and k == 1 and b == Nb - 1:
                vidx = 2
            hpi[C["HPI_CELL"], g] = cells
            hpi[C["HPI_BAND"], g] = b
            hpi[C["HPI_VEG_INDEX"], g] = vidx
            hpi[C["HPI_VEG_CLASS"], g] = int(veglib[vidx, C["VL_VEG_CLASS"]])
            hpi[C["HPI_IS_GLACIER"], g] = 1 if int(veglib[vidx, C["VL_VEG_CLASS"]]) == opt.GLACIER_ID else 0
            hpi[C["HPI_IS_ARTIFICIAL_BARE"], g] = 0
            hpd[C["HPD_CV"], g] = tile_frac[k] * area[b]
            r = roots[vidx]
            for l in range(3):
                hpd[C["HPD_ROOT0"] + l, g] = np.float32(r[l])
    if nbare:
        for b in range(Nb):
            g = (ntile * Nb + b) * ncell + cells
            hpi[C["HPI_CELL"], g] = cells
            hpi[C["HPI_BAND"], g] = b
            hpi[C["HPI_VEG_INDEX"], g] = nveg                  # veg_lib[num_veg_types]: the first appended PET surface
            hpi[C["HPI_VEG_CLASS"], g] = nveg
            hpi[C["HPI_IS_GLACIER"], g] = 0
            hpi[C["HPI_IS_ARTIFICIAL_BARE"], g] = 1
            hpd[C["HPD_CV"], g] = bare_fraction / Nb
    d.hru_iparams = np.ascontiguousarray(hpi)
    d.hru_dparams = np.ascontiguousarray(hpd)
    d.cell_hru_offset = np.ascontiguousarray((np.arange(ncell + 1) * nslot).astype(np.int32))
    d.cell_hru_list = np.ascontiguousarray((np.arange(nslot)[None, :] * ncell + cells[:, None]).reshape(-1).astype(np.int32))
    d.elevation = elevation
    d.cell_offset_T = cell_offset_T
    d.rng_seed = seed
    return d


def drop_hrus(d, drop):
    """Removes the HRUs flagged in the boolean array `drop` (a cell may end up with fewer HRUs than its neighbours, or
    none): ragged cell -> HRU lists as real vegetation parameter files produce them.  Returns the kept HRU indices."""
    keep = np.flatnonzero(~np.asarray(drop, dtype=bool))
    newid = -np.ones(d.nhru, dtype=np.int64)
    newid[keep] = np.arange(keep.size)
    d.hru_iparams = np.ascontiguousarray(d.hru_iparams[:, keep])
    d.hru_dparams = np.ascontiguousarray(d.hru_dparams[:, keep])
    off = [0]
    lst = []
    for c in range(d.ncell):
        g = d.cell_hru_list[d.cell_hru_offset[c]:d.cell_hru_offset[c + 1]]
        g = newid[g]
        g = g[g >= 0]
        lst.append(g)
        off.append(off[-1] + g.size)
    d.cell_hru_offset = np.ascontiguousarray(np.array(off, dtype=np.int32))
    d.cell_hru_list = np.ascontiguousarray(np.concatenate(lst).astype(np.int32)) if keep.size else np.zeros(0, np.int32)
    d.nhru = int(keep.size)
    return keep


def svp(T):
    """svp.c:7-24 (Pa)."""
    T = np.asarray(T, dtype=float)
    s = 0.61078 * np.exp(17.269 * T / (237.3 + T))
    s = np.where(T < 0, s * (1.0 + .00972 * T + .000042 * T * T), s)
    return s * 1000.


def make_dmy(opt, step0, nsteps, start_doy=1, year=2001):
    """dmy_struct per step (make_dmy.c:11), non-leap calendar."""
    out = np.zeros((nsteps, C["VIC_NDMY"]), dtype=np.int32)
    per_day = 24 // opt.dt
    cum = np.concatenate([[0], np.cumsum(_MONTH_DAYS)])
    for i in range(nsteps):
        s = step0 + i
        day_idx = s // per_day + (start_doy - 1)
        yr = year + day_idx // 365
        doy = day_idx % 365 + 1
        month = int(np.searchsorted(cum, doy - 1, side="right"))
        out[i, C["VIC_DMY_MONTH"]] = month
        out[i, C["VIC_DMY_DAY_IN_YEAR"]] = doy
        out[i, C["VIC_DMY_HOUR"]] = (s % per_day) * opt.dt
        out[i, C["VIC_DMY_DAY"]] = doy - cum[month - 1]
        out[i, C["VIC_DMY_YEAR"]] = yr
    return out


def make_forcing(d, step0, nsteps, start_doy=1, cold=0.0):
    """Synthetic forcing chunk: returns (forcing[nsteps][NFORCE][NF+1][ncell],
    snowflag uint8[nsteps][NF+1][ncell], dmy int32[nsteps][NDMY]).

    Sub-step values 0..NF-1 are snow_step-hour means; index NR is the step mean (sum for prec),
    as initialize_atmos.c fills atmos[rec].x[NR].  `cold` shifts air temperature (C).
    """
    opt = d.opt
    NF, NR = opt.NF, opt.NR
    ns = NR + 1
    nc = d.ncell
    f = np.zeros((nsteps, C["VIC_NFORCE"], ns, nc))
    sflag = np.zeros((nsteps, ns, nc), dtype=np.uint8)
    dmy = make_dmy(opt, step0, nsteps, start_doy)
    Nn, Nb = opt.Nnode, opt.Nband
    tf = np.stack([d.cell_params[abi.cp_band(C["CPB_TFACTOR"], b, Nn, Nb)] for b in range(Nb)])
    min_tf = tf.min(axis=0)
    max_snow = d.cell_params[C["CP_MAX_SNOW_TEMP"]]
    min_rain = d.cell_params[C["CP_MIN_RAIN_TEMP"]]
    # global cell numbers: a shard (make_domain(cell_range=...) / shard.shard_domain) draws the precipitation of ITS cells
    cell_id = np.arange(nc, dtype=np.uint64) + np.uint64(getattr(d, "global_cell0", 0))
    for i in range(nsteps):
        s = step0 + i
        for j in range(NF):
            hour_abs = s * opt.dt + j * opt.snow_step + 0.5 * opt.snow_step
            doy = (start_doy - 1) + hour_abs / 24.0
            hod = hour_abs % 24.0
            T = 4.0 + cold + d.cell_offset_T - 14.0 * np.cos(2 * np.pi * (doy - 15) / 365.0) \
                - 5.0 * np.cos(2 * np.pi * (hod - 3.0) / 24.0) - 6.5e-3 * (d.elevation - 1500.0)
            # counter-based hash -> uniform in [0,1): reproducible per (cell, sub-step) without RNG state
            k = (cell_id * np.uint64(2654435761) + np.uint64((s * NF + j) * 40503 + d.rng_seed)) & np.uint64(0xFFFFFFFF)
            k = (k ^ (k >> np.uint64(16))) * np.uint64(0x45d9f3b) & np.uint64(0xFFFFFFFF)
            k = (k ^ (k >> np.uint64(16))) * np.uint64(0x45d9f3b) & np.uint64(0xFFFFFFFF)
            k = k ^ (k >> np.uint64(16))
            u1 = (k & np.uint64(0xFFFF)).astype(float) / 65536.0
            u2 = ((k >> np.uint64(16)) & np.uint64(0xFFFF)).astype(float) / 65536.0
            prec = np.where(u1 < 0.08, 2.0 * u2 * opt.snow_step, 0.0)
            sw = np.maximum(0.0, np.sin(np.pi * (hod - 6.0) / 12.0)) * (450.0 + 300.0 * np.cos(2 * np.pi * (doy - 172) / 365.0))
            lw = 0.75 * 5.6696e-8 * (T + 273.15) ** 4
            P = 85000.0 * np.exp(-(d.elevation - 1500.0) / 8000.0)
            es = svp(T)
            vp = 0.7 * es
            wind = 1.5 + 3.0 * (((k >> np.uint64(8)) & np.uint64(0xFF)).astype(float) / 256.0)
            f[i, C["VIC_F_AIR_TEMP"], j] = T
            f[i, C["VIC_F_PREC"], j] = prec
            f[i, C["VIC_F_PRESSURE"], j] = P
            f[i, C["VIC_F_VP"], j] = vp
            f[i, C["VIC_F_VPD"], j] = es - vp
            f[i, C["VIC_F_DENSITY"], j] = P / (287.0 * (T + 273.15))
            f[i, C["VIC_F_SHORTWAVE"], j] = sw
            f[i, C["VIC_F_LONGWAVE"], j] = lw
            f[i, C["VIC_F_WIND"], j] = wind
            # snowflag: initialize_atmos.c:1275-1303 (KIENZLE / VIC_412 thresholds, coldest band)
            if opt.TEMP_TH_TYPE == C["VIC_TEMP_TH_KIENZLE"]:
                thr = max_snow + min_rain / 2
            else:
                thr = max_snow
            sflag[i, j] = ((T + min_tf) < thr) & (prec > 0)
        if NR > 0:
            for v in range(C["VIC_NFORCE"]):
                if v == C["VIC_F_PREC"]:
                    f[i, v, NR] = f[i, v, :NF].sum(axis=0)
                else:
                    f[i, v, NR] = f[i, v, :NF].mean(axis=0)
            # density[NR] comes from pressure[NR] and air_temp[NR] like every sub-step's (initialize_atmos.c:984-998)
            f[i, C["VIC_F_DENSITY"], NR] = f[i, C["VIC_F_PRESSURE"], NR] / (287.0 * (f[i, C["VIC_F_AIR_TEMP"], NR] + 273.15))
            sflag[i, NR] = sflag[i, :NF].max(axis=0)
    return np.ascontiguousarray(f), np.ascontiguousarray(sflag), dmy
