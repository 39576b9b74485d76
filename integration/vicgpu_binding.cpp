// vicgpu_binding.cpp -- see vicgpu_binding.h
#include <string.h>
#include "vicgpu_binding.h"

void vicgpu_binding_options(const ProgramState *state, vicgpu_options *o) {
  memset(o, 0, sizeof(*o));
  o->abi_version = VICGPU_ABI_VERSION;
  o->Nlayer = state->options.Nlayer; o->Nnode = state->options.Nnode; o->Nband = state->options.SNOW_BAND;
  o->dt = state->global_param.dt; o->snow_step = state->options.SNOW_STEP;
  o->FULL_ENERGY = state->options.FULL_ENERGY; o->FROZEN_SOIL = state->options.FROZEN_SOIL; o->QUICK_FLUX = state->options.QUICK_FLUX;
  o->NOFLUX = state->options.NOFLUX; o->EXP_TRANS = state->options.EXP_TRANS; o->GRND_FLUX_TYPE = state->options.GRND_FLUX_TYPE;
  o->TFALLBACK = state->options.TFALLBACK; o->AERO_RESIST_CANSNOW = state->options.AERO_RESIST_CANSNOW;
  o->SNOW_ALBEDO = state->options.SNOW_ALBEDO; o->SNOW_DENSITY = state->options.SNOW_DENSITY; o->TEMP_TH_TYPE = state->options.TEMP_TH_TYPE;
  o->GLACIER_ID = state->options.GLACIER_ID; o->GLACIER_DYNAMICS = state->options.GLACIER_DYNAMICS ? 1 : 0;
  o->CORRPREC = state->options.CORRPREC; o->IMPLICIT = state->options.IMPLICIT; o->BLOWING = state->options.BLOWING;
  o->QUICK_SOLVE = state->options.QUICK_SOLVE;
  o->NODE_SOLVER = VIC_NODE_SOLVER_BRENT;      // the reference's own node iteration; VIC_NODE_SOLVER_NEWTON is the faster, converged one
  o->frozen_compat = 1;                        // frozen_soil.c:218-221 as shipped (0 = the node arrays)
  o->nveg_types = state->veg_lib[0].NVegLibTypes;
  o->wind_h = state->global_param.wind_h;
}

void vicgpu_binding_number_hrus(const std::vector<cell_info_struct> &cells, VicGpuTables &t) {
  t.ncell = (int)cells.size();
  size_t maxn = 0, total = 0;
  for (size_t c = 0; c < cells.size(); c++) { const size_t n = cells[c].prcp.hruList.size(); total += n; if (n > maxn) maxn = n; }
  t.nhru = (int)total;
  t.hru_cell.clear(); t.hru_pos.clear();
  std::vector<std::vector<int> > ids(cells.size());
  for (size_t k = 0; k < maxn; k++)
    for (size_t c = 0; c < cells.size(); c++)
      if (k < cells[c].prcp.hruList.size()) { ids[c].push_back((int)t.hru_cell.size()); t.hru_cell.push_back((int)c); t.hru_pos.push_back((int)k); }
  t.cell_off.assign(cells.size() + 1, 0); t.cell_list.clear();
  for (size_t c = 0; c < cells.size(); c++) {
    t.cell_list.insert(t.cell_list.end(), ids[c].begin(), ids[c].end());
    t.cell_off[c + 1] = (int)t.cell_list.size();
  }
}

void vicgpu_binding_pack_veglib(const ProgramState *state, VicGpuTables &t) {
  const int nrow = state->veg_lib[0].NVegLibTypes + 4;        // + the four reference PET surfaces (compute_pot_evap.c)
  t.nveg_rows = nrow;
  t.veglib.assign((size_t)nrow * VL_NFIELD, 0.0);
  for (int i = 0; i < nrow; i++) {
    const veg_lib_struct &v = state->veg_lib[i];
    double *r = &t.veglib[(size_t)i * VL_NFIELD];
    r[VL_OVERSTORY] = v.overstory ? 1.0 : 0.0; r[VL_RARC] = v.rarc; r[VL_RMIN] = v.rmin; r[VL_RAD_ATTEN] = v.rad_atten;
    r[VL_TRUNK_RATIO] = v.trunk_ratio; r[VL_WIND_ATTEN] = v.wind_atten; r[VL_WIND_H] = v.wind_h; r[VL_RGL] = v.RGL;
    r[VL_VEG_CLASS] = v.veg_class;
    for (int m = 0; m < 12; m++) {
      r[VL_LAI + m] = v.LAI[m]; r[VL_WDMAX + m] = v.Wdmax[m]; r[VL_ALBEDO + m] = v.albedo[m]; r[VL_DISPLACEMENT + m] = v.displacement[m];
      r[VL_EMISSIVITY + m] = v.emissivity[m]; r[VL_ROUGHNESS + m] = v.roughness[m];
    }
  }
}

void vicgpu_binding_pack_domain(const ProgramState *state, const std::vector<cell_info_struct> &cells, VicGpuTables &t) {
  const int Nn = state->options.Nnode, Nb = state->options.SNOW_BAND;
  const size_t nc = cells.size(), nh = (size_t)t.nhru;
  t.Nnode = Nn; t.Nband = Nb;
  t.cell_params.assign((size_t)VICGPU_CP_NROW(Nn, Nb) * nc, 0.0);
#define CP(row) t.cell_params[(size_t)(row) * nc + c]
  for (size_t c = 0; c < nc; c++) {
    const soil_con_struct &sc = cells[c].soil_con;
    CP(CP_DS) = sc.Ds; CP(CP_DSMAX) = sc.Dsmax; CP(CP_WS) = sc.Ws; CP(CP_C) = sc.c; CP(CP_B_INFILT) = sc.b_infilt; CP(CP_DP) = sc.dp;
    CP(CP_AVG_TEMP) = sc.avg_temp; CP(CP_ROUGH) = sc.rough; CP(CP_SNOW_ROUGH) = sc.snow_rough; CP(CP_ELEVATION) = sc.elevation;
    CP(CP_LAT) = sc.lat; CP(CP_FS_ACTIVE) = sc.FS_ACTIVE;
    CP(CP_NEW_SNOW_ALB) = sc.NEW_SNOW_ALB; CP(CP_SNOW_ALB_ACCUM_A) = sc.SNOW_ALB_ACCUM_A; CP(CP_SNOW_ALB_ACCUM_B) = sc.SNOW_ALB_ACCUM_B;
    CP(CP_SNOW_ALB_THAW_A) = sc.SNOW_ALB_THAW_A; CP(CP_SNOW_ALB_THAW_B) = sc.SNOW_ALB_THAW_B;
    CP(CP_MIN_RAIN_TEMP) = sc.MIN_RAIN_TEMP; CP(CP_MAX_SNOW_TEMP) = sc.MAX_SNOW_TEMP; CP(CP_PADJ_R) = sc.PADJ_R; CP(CP_PADJ_S) = sc.PADJ_S;
    CP(CP_GLAC_SURF_THICK) = sc.GLAC_SURF_THICK; CP(CP_GLAC_SURF_WE) = sc.GLAC_SURF_WE; CP(CP_GLAC_KMIN) = sc.GLAC_KMIN;
    CP(CP_GLAC_DK) = sc.GLAC_DK; CP(CP_GLAC_A) = sc.GLAC_A; CP(CP_GLAC_ALBEDO) = sc.GLAC_ALBEDO; CP(CP_GLAC_ROUGH) = sc.GLAC_ROUGH;
    for (int l = 0; l < 3; l++) {
      CP(VICGPU_CP_LAYER(CPL_KSAT, l)) = sc.Ksat[l]; CP(VICGPU_CP_LAYER(CPL_WCR, l)) = sc.Wcr[l]; CP(VICGPU_CP_LAYER(CPL_WPWP, l)) = sc.Wpwp[l];
      CP(VICGPU_CP_LAYER(CPL_EXPT, l)) = sc.expt[l]; CP(VICGPU_CP_LAYER(CPL_BUBBLE, l)) = sc.bubble[l]; CP(VICGPU_CP_LAYER(CPL_DEPTH, l)) = sc.depth[l];
      CP(VICGPU_CP_LAYER(CPL_MAX_MOIST, l)) = sc.max_moist[l]; CP(VICGPU_CP_LAYER(CPL_RESID_MOIST, l)) = sc.resid_moist[l];
      CP(VICGPU_CP_LAYER(CPL_POROSITY, l)) = sc.porosity[l]; CP(VICGPU_CP_LAYER(CPL_QUARTZ, l)) = sc.quartz[l];
      CP(VICGPU_CP_LAYER(CPL_ORGANIC, l)) = sc.organic[l]; CP(VICGPU_CP_LAYER(CPL_BULK_DENSITY, l)) = sc.bulk_density[l];
      CP(VICGPU_CP_LAYER(CPL_SOIL_DENSITY, l)) = sc.soil_density[l]; CP(VICGPU_CP_LAYER(CPL_BULK_DENS_MIN, l)) = sc.bulk_dens_min[l];
      CP(VICGPU_CP_LAYER(CPL_SOIL_DENS_MIN, l)) = sc.soil_dens_min[l];
    }
    for (int n = 0; n < Nn; n++) {
      CP(VICGPU_CP_NODE(CPN_ZSUM, n, Nn)) = sc.Zsum_node[n]; CP(VICGPU_CP_NODE(CPN_DZ, n, Nn)) = sc.dz_node[n];
      CP(VICGPU_CP_NODE(CPN_ALPHA, n, Nn)) = sc.alpha[n]; CP(VICGPU_CP_NODE(CPN_BETA, n, Nn)) = sc.beta[n];
      CP(VICGPU_CP_NODE(CPN_GAMMA, n, Nn)) = sc.gamma[n]; CP(VICGPU_CP_NODE(CPN_MAX_MOIST, n, Nn)) = sc.max_moist_node[n];
      CP(VICGPU_CP_NODE(CPN_EXPT, n, Nn)) = sc.expt_node[n]; CP(VICGPU_CP_NODE(CPN_BUBBLE, n, Nn)) = sc.bubble_node[n];
    }
    for (int b = 0; b < Nb; b++) {
      CP(VICGPU_CP_BAND(CPB_AREAFRACT, b, Nn, Nb)) = sc.AreaFract[b]; CP(VICGPU_CP_BAND(CPB_TFACTOR, b, Nn, Nb)) = sc.Tfactor[b];
      CP(VICGPU_CP_BAND(CPB_PFACTOR, b, Nn, Nb)) = sc.Pfactor[b]; CP(VICGPU_CP_BAND(CPB_BANDELEV, b, Nn, Nb)) = sc.BandElev[b];
      CP(VICGPU_CP_BAND(CPB_ABOVETREELINE, b, Nn, Nb)) = sc.AboveTreeLine[b];
    }
    for (int l = 0; l < VIC_NLAYER + 2; l++)
      for (int i = 0; i < VIC_MAX_ZWTVMOIST; i++) {
        CP(VICGPU_CP_ZWT_ZWT(l, i, Nn, Nb)) = sc.zwtvmoist_zwt[l][i]; CP(VICGPU_CP_ZWT_MOIST(l, i, Nn, Nb)) = sc.zwtvmoist_moist[l][i];
      }
  }
#undef CP
  t.hpi.assign((size_t)HPI_NROW * nh, 0); t.hpd.assign((size_t)HPD_NROW * nh, 0.0);
  for (int g = 0; g < t.nhru; g++) {
    const HRU &u = cells[t.hru_cell[g]].prcp.hruList[t.hru_pos[g]];
    t.hpi[(size_t)HPI_CELL * nh + g] = t.hru_cell[g]; t.hpi[(size_t)HPI_BAND * nh + g] = u.bandIndex;
    t.hpi[(size_t)HPI_VEG_INDEX * nh + g] = u.veg_con.vegIndex; t.hpi[(size_t)HPI_VEG_CLASS * nh + g] = u.veg_con.vegClass;
    t.hpi[(size_t)HPI_IS_GLACIER * nh + g] = u.isGlacier ? 1 : 0; t.hpi[(size_t)HPI_IS_ARTIFICIAL_BARE * nh + g] = u.isArtificialBareSoil ? 1 : 0;
    t.hpd[(size_t)HPD_CV * nh + g] = u.veg_con.Cv;
    for (int l = 0; l < 3; l++) t.hpd[(size_t)(HPD_ROOT0 + l) * nh + g] = u.veg_con.root[l];
    t.hpd[(size_t)HPD_SIGMA_SLOPE * nh + g] = u.veg_con.sigma_slope; t.hpd[(size_t)HPD_LAG_ONE * nh + g] = u.veg_con.lag_one;
    t.hpd[(size_t)HPD_FETCH * nh + g] = u.veg_con.fetch;
  }
}

#define SDP(row) sd[(size_t)(row) * nh + g]
#define SIP(row) si[(size_t)(row) * nh + g]
void vicgpu_binding_state_to_tables(const std::vector<cell_info_struct> &cells, const int *hru_cell, const int *hru_pos, int nhru, int Nn,
                                    double *sd, int *si) {
  const size_t nh = nhru;
  for (int g = 0; g < nhru; g++) {
    const HRU &u = cells[hru_cell[g]].prcp.hruList[hru_pos[g]];
    const hru_data_struct &cw = u.cell[WET];
    for (int l = 0; l < 3; l++) {
      SDP(SD_MOIST0 + l) = cw.layer[l].moist; SDP(SD_ICE0 + l) = cw.layer[l].soil_ice; SDP(SD_LAYER_T0 + l) = cw.layer[l].T;
    }
    const energy_bal_struct &e = u.energy;
    SDP(SD_SNOW_FLUX) = e.snow_flux; SDP(SD_GRND_FLUX) = e.grnd_flux; SDP(SD_DELTAH) = e.deltaH; SDP(SD_FUSION) = e.fusion;
    SDP(SD_LONGUNDEROUT) = e.LongUnderOut; SDP(SD_TFOLIAGE) = e.Tfoliage;
    const snow_data_struct &s = u.snow;
    SDP(SD_SNOW_ALBEDO) = s.albedo; SDP(SD_SNOW_COLDCONTENT) = s.coldcontent; SDP(SD_SNOW_COVERAGE) = s.coverage;
    SDP(SD_SNOW_DENSITY) = s.density; SDP(SD_SNOW_DEPTH) = s.depth; SDP(SD_SNOW_PACK_TEMP) = s.pack_temp;
    SDP(SD_SNOW_PACK_WATER) = s.pack_water; SDP(SD_SNOW_CANOPY) = s.snow_canopy; SDP(SD_SNOW_SURF_TEMP) = s.surf_temp;
    SDP(SD_SNOW_SURF_WATER) = s.surf_water; SDP(SD_SNOW_SWQ) = s.swq; SDP(SD_SNOW_TMP_INT_STORAGE) = s.tmp_int_storage;
    SDP(SD_SNOW_STORE_SWQ) = s.store_swq; SDP(SD_SNOW_STORE_COVERAGE) = s.store_coverage; SDP(SD_SNOW_SWQ_SLOPE) = s.swq_slope;
    SDP(SD_SNOW_MAX_SWQ) = s.max_swq;
    SDP(SD_WDEW) = u.veg_var[WET].Wdew;
    SDP(SD_GLAC_SURF_TEMP) = u.glacier.surf_temp; SDP(SD_GLAC_WATER_STORAGE) = u.glacier.water_storage;
    SDP(SD_GLAC_CUM_MASS_BALANCE) = u.glacier.cum_mass_balance;
    SDP(SD_TCANOPY) = e.Tcanopy; SDP(SD_TSURF) = e.Tsurf; SDP(SD_ALBEDO_OVER) = e.AlbedoOver; SDP(SD_ALBEDO_UNDER) = e.AlbedoUnder;
    SDP(SD_CANOPY_ADVECTION) = e.canopy_advection; SDP(SD_CANOPY_LATENT) = e.canopy_latent;
    SDP(SD_CANOPY_LATENT_SUB) = e.canopy_latent_sub; SDP(SD_CANOPY_SENSIBLE) = e.canopy_sensible;
    SDP(SD_CANOPY_REFREEZE) = e.canopy_refreeze; SDP(SD_ADVECTED_SENSIBLE) = e.advected_sensible;
    SDP(SD_ADVECTION) = e.advection; SDP(SD_DELTACC) = e.deltaCC; SDP(SD_REFREEZE_ENERGY) = e.refreeze_energy;
    SDP(SD_MELT_ENERGY) = e.melt_energy; SDP(SD_ERROR) = e.error;
    SDP(SD_LATENT) = e.latent; SDP(SD_LATENT_SUB) = e.latent_sub; SDP(SD_SENSIBLE) = e.sensible;
    SDP(SD_LONGOVERIN) = e.LongOverIn; SDP(SD_NETLONGOVER) = e.NetLongOver; SDP(SD_NETSHORTOVER) = e.NetShortOver;
    SDP(SD_SHORTOVERIN) = e.ShortOverIn; SDP(SD_NETLONGUNDER) = e.NetLongUnder;
    for (int n = 0; n < Nn; n++) {
      SDP(VICGPU_SD_NODE(SDN_T, n, Nn)) = e.T[n]; SDP(VICGPU_SD_NODE(SDN_MOIST, n, Nn)) = e.moist[n];
      SDP(VICGPU_SD_NODE(SDN_ICE, n, Nn)) = e.ice_content[n]; SDP(VICGPU_SD_NODE(SDN_KAPPA, n, Nn)) = e.kappa_node[n];
      SDP(VICGPU_SD_NODE(SDN_CS, n, Nn)) = e.Cs_node[n];
      SIP(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)) = e.T_fbflag[n]; SIP(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn)) = e.T_fbcount[n];
    }
    SIP(SI_SNOW_LAST_SNOW) = s.last_snow; SIP(SI_SNOW_MELTING) = s.MELTING ? 1 : 0; SIP(SI_SNOW_SNOW) = s.snow;
    SIP(SI_SNOW_STORE_SNOW) = s.store_snow; SIP(SI_SNOW_SURF_TEMP_FBCOUNT) = s.surf_temp_fbcount;
    SIP(SI_SNOW_SURF_TEMP_FBFLAG) = s.surf_temp_fbflag ? 1 : 0;
    SIP(SI_TSURF_FBCOUNT) = e.Tsurf_fbcount; SIP(SI_TSURF_FBFLAG) = e.Tsurf_fbflag;
    SIP(SI_TFOLIAGE_FBCOUNT) = e.Tfoliage_fbcount; SIP(SI_TFOLIAGE_FBFLAG) = e.Tfoliage_fbflag;
    SIP(SI_TCANOPY_FBCOUNT) = e.Tcanopy_fbcount; SIP(SI_TCANOPY_FBFLAG) = e.Tcanopy_fbflag;
    SIP(SI_GLAC_SURF_TEMP_FBCOUNT) = u.glacier.surf_temp_fbcount; SIP(SI_GLAC_SURF_TEMP_FBFLAG) = u.glacier.surf_temp_fbflag ? 1 : 0;
    SIP(SI_FROZEN) = e.frozen; SIP(SI_NFROST) = e.Nfrost; SIP(SI_NTHAW) = e.Nthaw;
  }
}

void vicgpu_binding_tables_to_state(std::vector<cell_info_struct> &cells, const int *hru_cell, const int *hru_pos, int nhru, int Nn,
                                    const double *sd, const int *si) {
  const size_t nh = nhru;
  for (int g = 0; g < nhru; g++) {
    HRU &u = cells[hru_cell[g]].prcp.hruList[hru_pos[g]];
    hru_data_struct &cw = u.cell[WET];
    for (int l = 0; l < 3; l++) {
      cw.layer[l].moist = SDP(SD_MOIST0 + l); cw.layer[l].soil_ice = SDP(SD_ICE0 + l); cw.layer[l].T = SDP(SD_LAYER_T0 + l);
    }
    energy_bal_struct &e = u.energy;
    e.snow_flux = SDP(SD_SNOW_FLUX); e.grnd_flux = SDP(SD_GRND_FLUX); e.deltaH = SDP(SD_DELTAH); e.fusion = SDP(SD_FUSION);
    e.LongUnderOut = SDP(SD_LONGUNDEROUT); e.Tfoliage = SDP(SD_TFOLIAGE);
    snow_data_struct &s = u.snow;
    s.albedo = SDP(SD_SNOW_ALBEDO); s.coldcontent = SDP(SD_SNOW_COLDCONTENT); s.coverage = SDP(SD_SNOW_COVERAGE);
    s.density = SDP(SD_SNOW_DENSITY); s.depth = SDP(SD_SNOW_DEPTH); s.pack_temp = SDP(SD_SNOW_PACK_TEMP);
    s.pack_water = SDP(SD_SNOW_PACK_WATER); s.snow_canopy = SDP(SD_SNOW_CANOPY); s.surf_temp = SDP(SD_SNOW_SURF_TEMP);
    s.surf_water = SDP(SD_SNOW_SURF_WATER); s.swq = SDP(SD_SNOW_SWQ); s.tmp_int_storage = SDP(SD_SNOW_TMP_INT_STORAGE);
    s.store_swq = SDP(SD_SNOW_STORE_SWQ); s.store_coverage = SDP(SD_SNOW_STORE_COVERAGE); s.swq_slope = SDP(SD_SNOW_SWQ_SLOPE);
    s.max_swq = SDP(SD_SNOW_MAX_SWQ);
    u.veg_var[WET].Wdew = SDP(SD_WDEW);
    u.glacier.surf_temp = SDP(SD_GLAC_SURF_TEMP); u.glacier.water_storage = SDP(SD_GLAC_WATER_STORAGE);
    u.glacier.cum_mass_balance = SDP(SD_GLAC_CUM_MASS_BALANCE);
    e.Tcanopy = SDP(SD_TCANOPY); e.Tsurf = SDP(SD_TSURF); e.AlbedoOver = SDP(SD_ALBEDO_OVER); e.AlbedoUnder = SDP(SD_ALBEDO_UNDER);
    e.canopy_advection = SDP(SD_CANOPY_ADVECTION); e.canopy_latent = SDP(SD_CANOPY_LATENT);
    e.canopy_latent_sub = SDP(SD_CANOPY_LATENT_SUB); e.canopy_sensible = SDP(SD_CANOPY_SENSIBLE);
    e.canopy_refreeze = SDP(SD_CANOPY_REFREEZE); e.advected_sensible = SDP(SD_ADVECTED_SENSIBLE);
    e.advection = SDP(SD_ADVECTION); e.deltaCC = SDP(SD_DELTACC); e.refreeze_energy = SDP(SD_REFREEZE_ENERGY);
    e.melt_energy = SDP(SD_MELT_ENERGY); e.error = SDP(SD_ERROR);
    e.latent = SDP(SD_LATENT); e.latent_sub = SDP(SD_LATENT_SUB); e.sensible = SDP(SD_SENSIBLE);
    e.LongOverIn = SDP(SD_LONGOVERIN); e.NetLongOver = SDP(SD_NETLONGOVER); e.NetShortOver = SDP(SD_NETSHORTOVER);
    e.ShortOverIn = SDP(SD_SHORTOVERIN); e.NetLongUnder = SDP(SD_NETLONGUNDER);
    for (int n = 0; n < Nn; n++) {
      e.T[n] = SDP(VICGPU_SD_NODE(SDN_T, n, Nn)); e.moist[n] = SDP(VICGPU_SD_NODE(SDN_MOIST, n, Nn));
      e.ice_content[n] = SDP(VICGPU_SD_NODE(SDN_ICE, n, Nn)); e.kappa_node[n] = SDP(VICGPU_SD_NODE(SDN_KAPPA, n, Nn));
      e.Cs_node[n] = SDP(VICGPU_SD_NODE(SDN_CS, n, Nn));
      e.T_fbflag[n] = (char)SIP(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)); e.T_fbcount[n] = SIP(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn));
    }
    s.last_snow = SIP(SI_SNOW_LAST_SNOW); s.MELTING = SIP(SI_SNOW_MELTING) != 0; s.snow = SIP(SI_SNOW_SNOW);
    s.store_snow = SIP(SI_SNOW_STORE_SNOW); s.surf_temp_fbcount = SIP(SI_SNOW_SURF_TEMP_FBCOUNT);
    s.surf_temp_fbflag = SIP(SI_SNOW_SURF_TEMP_FBFLAG) != 0;
    e.Tsurf_fbcount = SIP(SI_TSURF_FBCOUNT); e.Tsurf_fbflag = (char)SIP(SI_TSURF_FBFLAG);
    e.Tfoliage_fbcount = SIP(SI_TFOLIAGE_FBCOUNT); e.Tfoliage_fbflag = (char)SIP(SI_TFOLIAGE_FBFLAG);
    e.Tcanopy_fbcount = SIP(SI_TCANOPY_FBCOUNT); e.Tcanopy_fbflag = (char)SIP(SI_TCANOPY_FBFLAG);
    u.glacier.surf_temp_fbcount = SIP(SI_GLAC_SURF_TEMP_FBCOUNT); u.glacier.surf_temp_fbflag = SIP(SI_GLAC_SURF_TEMP_FBFLAG) != 0;
    e.frozen = (char)SIP(SI_FROZEN); e.Nfrost = SIP(SI_NFROST); e.Nthaw = SIP(SI_NTHAW);
  }
}
#undef SDP
#undef SIP

void vicgpu_binding_pack_forcing(const std::vector<cell_info_struct> &cells, int rec, int NR, double *forcing, unsigned char *snowflag) {
  const int ns = NR + 1;
  const size_t nc = cells.size();
  for (size_t c = 0; c < nc; c++) {
    const atmos_data_struct &a = cells[c].atmos[rec];
    for (int s = 0; s < ns; s++) {
#define FV(v) forcing[((size_t)(v) * ns + s) * nc + c]
      FV(VIC_F_AIR_TEMP) = a.air_temp[s]; FV(VIC_F_PREC) = a.prec[s]; FV(VIC_F_PRESSURE) = a.pressure[s]; FV(VIC_F_VP) = a.vp[s];
      FV(VIC_F_VPD) = a.vpd[s]; FV(VIC_F_DENSITY) = a.density[s]; FV(VIC_F_SHORTWAVE) = a.shortwave[s]; FV(VIC_F_LONGWAVE) = a.longwave[s];
      FV(VIC_F_WIND) = a.wind[s];
#undef FV
      snowflag[(size_t)s * nc + c] = a.snowflag[s] ? 1 : 0;
    }
  }
}

VicGpuBinding::VicGpuBinding(const ProgramState *st, std::vector<cell_info_struct> &cl, int device, int frozen_compat, int node_solver)
    : state(st), cells(cl), ctx(NULL) {
  vicgpu_options opt;
  vicgpu_binding_options(state, &opt);
  opt.frozen_compat = frozen_compat; opt.NODE_SOLVER = node_solver;
  if (vicgpu_create(&opt, device, &ctx) != VICGPU_OK) { ctx = NULL; return; }
  VicGpuTables &t = tables;
  vicgpu_binding_number_hrus(cells, t);
  vicgpu_binding_pack_veglib(state, t);
  vicgpu_binding_pack_domain(state, cells, t);
  t.sd.assign((size_t)VICGPU_SD_NROW(t.Nnode) * t.nhru, 0.0); t.si.assign((size_t)VICGPU_SI_NROW(t.Nnode) * t.nhru, 0);
  vicgpu_binding_state_to_tables(cells, t.hru_cell.data(), t.hru_pos.data(), t.nhru, t.Nnode, t.sd.data(), t.si.data());
  int r = vicgpu_set_veglib(ctx, t.nveg_rows, t.veglib.data());
  if (r == VICGPU_OK) r = vicgpu_set_domain(ctx, t.ncell, t.nhru, t.cell_params.data(), t.hpi.data(), t.hpd.data(), t.cell_off.data(), t.cell_list.data());
  if (r == VICGPU_OK) r = vicgpu_set_state(ctx, t.sd.data(), t.si.data());
  if (r != VICGPU_OK) { vicgpu_destroy(ctx); ctx = NULL; }
}

VicGpuBinding::~VicGpuBinding() { if (ctx) vicgpu_destroy(ctx); }
const char *VicGpuBinding::error() const { return ctx ? vicgpu_last_error(ctx) : "vicgpu_create / set_domain failed"; }

int VicGpuBinding::enable_put_data(int out_step_ratio) {
  if (!ctx) return VICGPU_ERR_STATE;
  // the per-HRU values put_data reads that live outside the state tables: here the frost / thaw fronts of
  // initialize_model_state (energy.fdepth / tdepth); everything else is rewritten by the first step
  VicGpuTables &t = tables;
  std::vector<double> flux((size_t)FX_NROW * t.nhru, 0.0);
  for (int g = 0; g < t.nhru; g++) {
    const HRU &u = cells[t.hru_cell[g]].prcp.hruList[t.hru_pos[g]];
    for (int l = 0; l < VIC_MAX_FRONTS; l++) {
      flux[(size_t)(FX_FDEPTH0 + l) * t.nhru + g] = u.energy.fdepth[l];
      flux[(size_t)(FX_TDEPTH0 + l) * t.nhru + g] = u.energy.tdepth[l];
    }
    for (int l = 0; l < 3; l++) flux[(size_t)(FX_ZWTL0 + l) * t.nhru + g] = u.cell[WET].layer[l].zwt;
    flux[(size_t)FX_AERO_RESIST_SURFACE * t.nhru + g] = u.cell[WET].aero_resist.surface;
    flux[(size_t)FX_AERO_RESIST_OVERSTORY * t.nhru + g] = u.cell[WET].aero_resist.overstory;
  }
  int r = vicgpu_set_fluxes(ctx, flux.data());
  if (r == VICGPU_OK) r = vicgpu_put_data_config(ctx, out_step_ratio);
  if (r == VICGPU_OK) r = vicgpu_put_data_init(ctx);
  return r;
}

int VicGpuBinding::outputs(const std::vector<std::string> &names, std::vector<float> &out, bool reset) {
  if (!ctx) return VICGPU_ERR_STATE;
  vicgpu_options opt;
  vicgpu_binding_options(state, &opt);
  std::vector<int> ids;
  int rows = 0;
  for (size_t i = 0; i < names.size(); i++) {
    const int id = vicgpu_out_var_id(names[i].c_str());
    if (id < 0) return VICGPU_ERR_ARG;
    ids.push_back(id);
    rows += vicgpu_out_var_nelem(&opt, id);
  }
  out.assign((size_t)rows * cells.size(), 0.f);
  const int r = vicgpu_get_outputs(ctx, (int)ids.size(), ids.data(), out.data(), reset ? 1 : 0);
  return r == VICGPU_OK ? rows : r;
}

int VicGpuBinding::run(int rec0, int nrec, const dmy_struct *dmy) {
  if (!ctx) return VICGPU_ERR_STATE;
  const int ns = state->NR + 1;
  const size_t nc = cells.size();
  std::vector<double> forcing((size_t)nrec * VIC_NFORCE * ns * nc);
  std::vector<unsigned char> snowflag((size_t)nrec * ns * nc);
  std::vector<int> d((size_t)nrec * VIC_NDMY);
  for (int r = 0; r < nrec; r++) {
    vicgpu_binding_pack_forcing(cells, rec0 + r, state->NR, &forcing[(size_t)r * VIC_NFORCE * ns * nc], &snowflag[(size_t)r * ns * nc]);
    const dmy_struct &m = dmy[rec0 + r];
    d[(size_t)r * VIC_NDMY + VIC_DMY_MONTH] = m.month; d[(size_t)r * VIC_NDMY + VIC_DMY_DAY_IN_YEAR] = m.day_in_year;
    d[(size_t)r * VIC_NDMY + VIC_DMY_HOUR] = m.hour; d[(size_t)r * VIC_NDMY + VIC_DMY_DAY] = m.day; d[(size_t)r * VIC_NDMY + VIC_DMY_YEAR] = m.year;
  }
  int r = vicgpu_push_forcing(ctx, nrec, forcing.data(), snowflag.data(), d.data());
  if (r == VICGPU_OK) r = vicgpu_step(ctx, 0, nrec);
  if (r == VICGPU_OK) r = vicgpu_synchronize(ctx);
  return r;
}

int VicGpuBinding::finish(int *flags) {
  if (!ctx) return VICGPU_ERR_STATE;
  VicGpuTables &t = tables;
  int r = vicgpu_get_state(ctx, t.sd.data(), t.si.data());
  if (r != VICGPU_OK) return r;
  vicgpu_binding_tables_to_state(cells, t.hru_cell.data(), t.hru_pos.data(), t.nhru, t.Nnode, t.sd.data(), t.si.data());
  if (flags) r = vicgpu_get_cell_errors(ctx, flags);
  return r;
}
