/*
 * orc_putdata.c — TEST INFRASTRUCTURE (CPU oracle): the per-cell output aggregation, put_data.c:7-760 with
 * collect_wb_terms (:762-948) and collect_eb_terms (:950-1232), the balance checks of
 * calc_water_energy_balance_errors.c:7-94 and the temporal aggregation of put_data.c:663-685, for the option
 * subset of this repository (no lakes, Ndist = 1, SPATIAL_FROST / EXCESS_ICE off, MOISTFRACT / ALMA_OUTPUT off).
 * Pinned against the reference's own put_data through oracle/ref_build/vicref_shim.cpp (tests/test_putdata.py).
 */
#include <stdlib.h>
#include "orc.h"
#include "vicgpu_out.h"

#define ORC_OUT_KIND_(name, kind, agg) kind,
#define ORC_OUT_AGG_(name, kind, agg) agg,
#define ORC_OUT_NAME_(name, kind, agg) "OUT_" #name,
static const int out_kind[VOUT_NVAR] = {VICGPU_OUT_VARS(ORC_OUT_KIND_)};
static const int out_agg[VOUT_NVAR] = {VICGPU_OUT_VARS(ORC_OUT_AGG_)};
static const char *const out_name[VOUT_NVAR] = {VICGPU_OUT_VARS(ORC_OUT_NAME_)};

static int kind_nelem(const vicgpu_options *o, int kind) {
  switch (kind) {
    case VOUT_KLAYER: return VIC_NLAYER;
    case VOUT_KNODE: return o->Nnode;
    case VOUT_KBAND: return o->Nband;
    case VOUT_KFRONT: return o->FROZEN_SOIL ? VIC_MAX_FRONTS : 1;      /* output_list_utils.c:298-301 */
    default: return 1;
  }
}

int vicorc_out_nvar(void) { return VOUT_NVAR; }
const char *vicorc_out_var_name(int id) { return (id >= 0 && id < VOUT_NVAR) ? out_name[id] : NULL; }
int vicorc_out_var_id(const char *name) {
  int v;
  for (v = 0; v < VOUT_NVAR; v++)
    if (strcmp(name, out_name[v]) == 0) return v;
  return -1;
}
int vicorc_out_var_nelem(void *hv, int id) { return (id >= 0 && id < VOUT_NVAR) ? kind_nelem(&((vicorc_handle *)hv)->model.opt, out_kind[id]) : -1; }
int vicorc_out_var_agg(int id) { return (id >= 0 && id < VOUT_NVAR) ? out_agg[id] : -1; }

static int row0(const vicgpu_options *o, int id) {
  int v, r = 0;
  for (v = 0; v < id; v++) r += kind_nelem(o, out_kind[v]);
  return r;
}
static int nrow_total(const vicgpu_options *o) { return row0(o, VOUT_NVAR); }

/* put_data for every cell.  rec < 0: the initialisation call (vicNl.c:524-541).  forcing: the step's
 * [VIC_NFORCE][NF+1][ncell] (NULL when rec < 0); cell_out: [CO_NROW][ncell] of the step. */
int vicorc_put_data(void *hv, int rec, const double *forcing, const double *cell_out, int out_step_ratio) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const vicgpu_options *o = &h->model.opt;
  const int nc = h->ncell, Nn = o->Nnode, Nb = o->Nband, NR = h->model.NR, ns = NR + 1;
  const int nrow = nrow_total(o);
  const double dt_sec = (double)o->dt * 3600.;
  int off[VOUT_NVAR], c, v;
  if (!h->out_data) {
    h->out_data = (double *)calloc((size_t)nrow * nc, sizeof(double));
    h->out_agg = (double *)calloc((size_t)nrow * nc, sizeof(double));
    h->pb = (double *)calloc((size_t)PB_NROW * nc, sizeof(double));
  }
  for (v = 0; v < VOUT_NVAR; v++) off[v] = row0(o, v);
#define OD(var, i) od[off[VOUT_##var] + (i)]
#define PB(r) h->pb[(size_t)(r) * nc + c]
  for (c = 0; c < nc; c++) {
    const orc_soil *sc = &h->soil[c];
    double od[1024];
    double bandCv[VIC_MAX_BANDS], TreeAdjust[VIC_MAX_BANDS];
    double cv_baresoil = 0, cv_veg = 0, cv_overstory = 0, cv_snow = 0, cv_glacier = 0;
    int k, b, l, n, r;
    if (nrow > 1024) return -1;
    for (r = 0; r < nrow; r++) od[r] = 0;                                           /* zero_output_list */
    for (b = 0; b < VIC_MAX_BANDS; b++) bandCv[b] = 0;
    for (k = h->cell_off[c]; k < h->cell_off[c + 1]; k++) {                         /* put_data.c:185-196 */
      const orc_hru *u = &h->hru[h->cell_list[k]];
      if (orc_veg(&h->model, u->veg_index)[VL_OVERSTORY] != 0) bandCv[u->band] += u->Cv;
    }
    for (b = 0; b < Nb; b++) TreeAdjust[b] = sc->AboveTreeLine[b] ? 1. / (1. - bandCv[b]) : 1.;          /* :199-208 */
    if (rec >= 0) {                                                                 /* :229-256 */
#define FV(var) forcing[((size_t)(var) * ns + NR) * nc + c]
      OD(AIR_TEMP, 0) = FV(VIC_F_AIR_TEMP); OD(DENSITY, 0) = FV(VIC_F_DENSITY); OD(LONGWAVE, 0) = FV(VIC_F_LONGWAVE);
      OD(PREC, 0) = cell_out[(size_t)CO_OUT_PREC * nc + c];
      OD(PRESSURE, 0) = FV(VIC_F_PRESSURE) / 1000.;
      OD(QAIR, 0) = ORC_EPS * FV(VIC_F_VP) / FV(VIC_F_PRESSURE);
      OD(RAINF, 0) = cell_out[(size_t)CO_OUT_RAIN * nc + c];
      OD(REL_HUMID, 0) = 100. * FV(VIC_F_VP) / (FV(VIC_F_VP) + FV(VIC_F_VPD));
      OD(SHORTWAVE, 0) = FV(VIC_F_SHORTWAVE);
      OD(SNOWF, 0) = cell_out[(size_t)CO_OUT_SNOW * nc + c];
      OD(VP, 0) = FV(VIC_F_VP) / 1000.; OD(VPD, 0) = FV(VIC_F_VPD) / 1000.; OD(WIND, 0) = FV(VIC_F_WIND);
#undef FV
    }
    for (k = h->cell_off[c]; k < h->cell_off[c + 1]; k++) {                         /* :260-545 */
      const orc_hru *u = &h->hru[h->cell_list[k]];
      const orc_energy *e = &u->energy; const orc_snow *s = &u->snow; const orc_glac *g = &u->glac;
      const double Cv = u->Cv;
      const int HasVeg = !(u->is_artificial_bare || u->is_glacier), HasGlac = u->is_glacier;
      const int overstory = orc_veg(&h->model, u->veg_index)[VL_OVERSTORY] != 0;
      const int band = u->band;
      double ThisAreaFract, ThisTreeAdjust, AreaFactor, tmp_evap, tmp_cond1, tmp_cond2, rad_temp, tmp_fract, bandFactor;
      if (!(Cv > 0)) continue;
      ThisAreaFract = sc->AreaFract[band]; ThisTreeAdjust = TreeAdjust[band];
      if (!(ThisAreaFract > 0. && (u->is_artificial_bare || (!sc->AboveTreeLine[band] || (sc->AboveTreeLine[band] && !overstory)))))
        continue;                                                                   /* :289-290 */
      OD(ELEV_BAND, band) = (double)(float)sc->BandElev[band];
      if (HasVeg) cv_veg += Cv * 1. * ThisTreeAdjust; else cv_baresoil += Cv * 1. * ThisTreeAdjust;
      if (overstory) cv_overstory += Cv * 1. * ThisTreeAdjust;
      if (s->swq > 0.0) cv_snow += Cv * 1. * ThisTreeAdjust;
      if (HasGlac) cv_glacier += Cv * 1. * ThisTreeAdjust;

      /* ---- collect_wb_terms, put_data.c:762-948 (mu = 1, lakefactor = 1) */
      AreaFactor = Cv * 1. * ThisTreeAdjust * 1.;
      tmp_evap = 0.0;
      for (l = 0; l < 3; l++) tmp_evap += u->layer[l].evap;
      if (HasVeg) OD(TRANSP_VEG, 0) += tmp_evap * AreaFactor; else OD(EVAP_BARE, 0) += tmp_evap * AreaFactor;
      tmp_evap += s->vapor_flux * 1000.;
      OD(SUB_SNOW, 0) += s->vapor_flux * 1000. * AreaFactor;
      OD(SUB_SURFACE, 0) += s->surface_flux * 1000. * AreaFactor;
      OD(SUB_BLOWING, 0) += s->blowing_flux * 1000. * AreaFactor;
      if (HasVeg) { tmp_evap += s->canopy_vapor_flux * 1000.; OD(SUB_CANOP, 0) += s->canopy_vapor_flux * 1000. * AreaFactor; }
      if (HasVeg) { tmp_evap += u->veg.canopyevap; OD(EVAP_CANOP, 0) += u->veg.canopyevap * AreaFactor; }
      if (HasGlac) tmp_evap += g->vapor_flux * 1000.;
      OD(EVAP, 0) += tmp_evap * AreaFactor;
      OD(PET_SATSOIL, 0) += u->pot_evap[0] * AreaFactor; OD(PET_H2OSURF, 0) += u->pot_evap[1] * AreaFactor;
      OD(PET_SHORT, 0) += u->pot_evap[2] * AreaFactor; OD(PET_TALL, 0) += u->pot_evap[3] * AreaFactor;
      OD(PET_NATVEG, 0) += u->pot_evap[4] * AreaFactor; OD(PET_VEGNOCR, 0) += u->pot_evap[5] * AreaFactor;
      OD(ASAT, 0) += u->asat * AreaFactor;
      OD(RUNOFF, 0) += u->runoff * AreaFactor;
      OD(BASEFLOW, 0) += u->baseflow * AreaFactor;
      OD(INFLOW, 0) += (u->inflow) * AreaFactor;
      if (HasVeg) OD(WDEW, 0) += u->veg.Wdew * AreaFactor;
      if (u->aero_resist_surface > ORC_SMALL) tmp_cond1 = (1 / u->aero_resist_surface) * AreaFactor; else tmp_cond1 = ORC_HUGE_RESIST;
      OD(AERO_COND1, 0) += tmp_cond1;
      if (overstory) {
        if (u->aero_resist_overstory > ORC_SMALL) tmp_cond2 = (1 / u->aero_resist_overstory) * AreaFactor; else tmp_cond2 = ORC_HUGE_RESIST;
      } else tmp_cond2 = ORC_HUGE_RESIST;
      OD(AERO_COND2, 0) += tmp_cond2;
      if (overstory) OD(AERO_COND, 0) += tmp_cond2; else OD(AERO_COND, 0) += tmp_cond1;
      for (l = 0; l < 3; l++) {
        double tmp_moist = u->layer[l].moist, tmp_ice = u->layer[l].ice;
        tmp_moist -= tmp_ice;
        OD(SOIL_LIQ, l) += tmp_moist * AreaFactor;
        OD(SOIL_ICE, l) += tmp_ice * AreaFactor;
      }
      OD(SOIL_WET, 0) += u->wetness * AreaFactor;
      OD(ROOTMOIST, 0) += u->rootmoist * AreaFactor;
      OD(ZWT, 0) += u->zwt * AreaFactor; OD(ZWT2, 0) += u->zwt2 * AreaFactor; OD(ZWT3, 0) += u->zwt3 * AreaFactor;
      for (l = 0; l < 3; l++) OD(ZWTL, l) += u->layer[l].zwt * AreaFactor;
      for (l = 0; l < 3; l++) OD(SOIL_TEMP, l) += u->layer[l].T * AreaFactor;
      OD(SWE, 0) += s->swq * AreaFactor * 1000.;
      OD(SNOW_DEPTH, 0) += s->depth * AreaFactor * 100.;
      if (s->swq > 0.0) {
        OD(SALBEDO, 0) += s->albedo * AreaFactor;
        OD(SNOW_SURF_TEMP, 0) += s->surf_temp * AreaFactor;
        OD(SNOW_PACK_TEMP, 0) += s->pack_temp * AreaFactor;
      }
      if (HasVeg) OD(SNOW_CANOPY, 0) += (s->snow_canopy) * AreaFactor * 1000.;
      OD(SNOW_MELT, 0) += s->melt * AreaFactor * 1000.;                              /* sic */
      OD(SNOW_COVER, 0) += s->coverage * AreaFactor;
      if (HasGlac) {
        OD(GLAC_WAT_STOR, 0) += g->water_storage * AreaFactor * 1000.;
        OD(GLAC_AREA, 0) += AreaFactor;
        OD(GLAC_MBAL, 0) += g->mass_balance * AreaFactor * 1000.;
        OD(GLAC_IMBAL, 0) += g->ice_mass_balance * AreaFactor * 1000.;
        OD(GLAC_ACCUM, 0) += g->accumulation * AreaFactor * 1000.;
        OD(GLAC_MELT, 0) += g->melt * AreaFactor * 1000.;
        OD(GLAC_SUB, 0) += g->vapor_flux * AreaFactor * 1000.;
        OD(GLAC_INFLOW, 0) += g->inflow * AreaFactor * 1000.;
        OD(GLAC_OUTFLOW, 0) += g->outflow * AreaFactor * 1000.;
        OD(GLAC_OUTFLOW_COEF, 0) += g->outflow_coef * AreaFactor;
      }

      /* ---- collect_eb_terms, put_data.c:950-1232 */
      AreaFactor = Cv * ThisTreeAdjust * 1.;
      if (o->FROZEN_SOIL) {
        for (l = 0; l < VIC_MAX_FRONTS; l++) {
          if (!isnan(e->fdepth[l])) OD(FDEPTH, l) += e->fdepth[l] * AreaFactor * 100.;
          if (!isnan(e->tdepth[l])) OD(TDEPTH, l) += e->tdepth[l] * AreaFactor * 100.;
        }
      }
      tmp_fract = 0;
      if (u->layer[0].ice > 0) tmp_fract = 1.;
      OD(SURF_FROST_FRAC, 0) += tmp_fract * AreaFactor;
      if (overstory && s->snow) rad_temp = e->Tcanopy + ORC_KELVIN; else rad_temp = e->Tsurf + ORC_KELVIN;
      if (HasVeg) OD(BARESOILT, 0) += (rad_temp - ORC_KELVIN) * AreaFactor;          /* sic: inverted (Appendix C #10) */
      else {
        if (overstory && !s->snow) OD(VEGT, 0) += e->Tfoliage * AreaFactor;
        else OD(VEGT, 0) += (rad_temp - ORC_KELVIN) * AreaFactor;
      }
      OD(SURF_TEMP, 0) += e->Tsurf * AreaFactor;
      for (n = 0; n < Nn; n++) OD(SOIL_TNODE, n) += e->T[n] * AreaFactor;
      OD(SURFT_FBFLAG, 0) += e->Tsurf_fbflag * AreaFactor;
      PB(PB_FB_TSURF) += e->Tsurf_fbcount;
      for (n = 0; n < Nn; n++) { OD(SOILT_FBFLAG, n) += e->T_fbflag[n] * AreaFactor; PB(PB_FB_TSOIL) += e->T_fbcount[n]; }
      OD(SNOWT_FBFLAG, 0) += s->surf_temp_fbflag * AreaFactor; PB(PB_FB_TSNOWSURF) += s->surf_temp_fbcount;
      OD(TFOL_FBFLAG, 0) += e->Tfoliage_fbflag * AreaFactor; PB(PB_FB_TFOLIAGE) += e->Tfoliage_fbcount;
      OD(TCAN_FBFLAG, 0) += e->Tcanopy_fbflag * AreaFactor; PB(PB_FB_TCANOPY) += e->Tcanopy_fbcount;
      OD(GLAC_TSURF_FBFLAG, 0) += g->surf_temp_fbflag * AreaFactor; PB(PB_FB_TGLACSURF) += g->surf_temp_fbcount;
      OD(NET_SHORT, 0) += e->NetShortAtmos * AreaFactor;
      OD(NET_LONG, 0) += e->NetLongAtmos * AreaFactor;
      if (s->snow && overstory) OD(IN_LONG, 0) += e->LongOverIn * AreaFactor; else OD(IN_LONG, 0) += e->LongUnderIn * AreaFactor;
      if (s->snow && overstory) OD(ALBEDO, 0) += e->AlbedoOver * AreaFactor; else OD(ALBEDO, 0) += e->AlbedoUnder * AreaFactor;
      OD(LATENT, 0) -= e->AtmosLatent * AreaFactor;
      OD(LATENT_SUB, 0) -= e->AtmosLatentSub * AreaFactor;
      OD(SENSIBLE, 0) -= e->AtmosSensible * AreaFactor;
      OD(GRND_FLUX, 0) -= e->grnd_flux * AreaFactor;
      OD(DELTAH, 0) -= e->deltaH * AreaFactor;
      OD(FUSION, 0) -= e->fusion * AreaFactor;
      OD(ENERGY_ERROR, 0) += e->error * AreaFactor;
      OD(RAD_TEMP, 0) += ((rad_temp) * (rad_temp) * (rad_temp) * (rad_temp)) * AreaFactor;
      OD(DELTACC, 0) += e->deltaCC * AreaFactor;
      if (s->snow && overstory) OD(ADVECTION, 0) += e->canopy_advection * AreaFactor;
      OD(ADVECTION, 0) += e->advection * AreaFactor;
      OD(SNOW_FLUX, 0) += e->snow_flux * AreaFactor;
      if (s->snow && overstory) OD(RFRZ_ENERGY, 0) += e->canopy_refreeze * AreaFactor;
      OD(RFRZ_ENERGY, 0) += e->refreeze_energy * AreaFactor;
      OD(MELT_ENERGY, 0) += e->melt_energy * AreaFactor;
      if (!overstory) OD(ADV_SENS, 0) -= e->advected_sensible * AreaFactor;
      if (HasGlac) {
        OD(GLAC_SURF_TEMP, 0) += g->surf_temp * AreaFactor;
        OD(GLAC_DELTACC, 0) += e->deltaCC_glac * AreaFactor;
        OD(GLAC_FLUX, 0) += e->glacier_flux * AreaFactor;
        OD(GLAC_MELT_ENERGY, 0) += e->glacier_melt_energy * AreaFactor;
      }
      bandFactor = Cv * 1. / ThisAreaFract;
      OD(AREA_BAND, band) += (Cv * 1.);
      OD(SWE_BAND, band) += s->swq * bandFactor * 1000.;
      OD(SNOW_DEPTH_BAND, band) += s->depth * bandFactor * 100.;
      if (HasVeg) OD(SNOW_CANOPY_BAND, band) += (s->snow_canopy) * bandFactor * 1000.;
      OD(SNOW_MELT_BAND, band) += s->melt * bandFactor;
      OD(SNOW_COVER_BAND, band) += s->coverage * bandFactor;
      OD(DELTACC_BAND, band) += e->deltaCC * bandFactor;
      OD(ADVECTION_BAND, band) += e->advection * bandFactor;
      OD(SNOW_FLUX_BAND, band) += e->snow_flux * bandFactor;
      OD(RFRZ_ENERGY_BAND, band) += e->refreeze_energy * bandFactor;
      OD(MELT_ENERGY_BAND, band) += e->melt_energy * bandFactor;
      OD(ADV_SENS_BAND, band) -= e->advected_sensible * bandFactor;
      OD(SNOW_SURFT_BAND, band) += s->surf_temp * bandFactor;
      OD(SNOW_PACKT_BAND, band) += s->pack_temp * bandFactor;
      OD(LATENT_SUB_BAND, band) += e->latent_sub * bandFactor;
      OD(NET_SHORT_BAND, band) += e->NetShortAtmos * bandFactor;
      OD(NET_LONG_BAND, band) += e->NetLongAtmos * bandFactor;
      if (s->snow && overstory) OD(ALBEDO_BAND, band) += e->AlbedoOver * bandFactor; else OD(ALBEDO_BAND, band) += e->AlbedoUnder * bandFactor;
      OD(LATENT_BAND, band) -= e->latent * bandFactor;
      OD(SENSIBLE_BAND, band) -= e->sensible * bandFactor;
      OD(GRND_FLUX_BAND, band) -= e->grnd_flux * bandFactor;
      if (HasGlac) {
        OD(GLAC_DELTACC_BAND, band) += e->deltaCC_glac;
        OD(GLAC_FLUX_BAND, band) += e->glacier_flux;
        OD(GLAC_WAT_STOR_BAND, band) += g->water_storage * 1000.;
        OD(GLAC_AREA_BAND, band) += Cv;
        OD(GLAC_MBAL_BAND, band) += g->mass_balance * 1000.;
        OD(GLAC_IMBAL_BAND, band) += g->ice_mass_balance * 1000.;
        OD(GLAC_ACCUM_BAND, band) += g->accumulation * 1000.;
        OD(GLAC_MELT_BAND, band) += g->melt * 1000.;
        OD(GLAC_SUB_BAND, band) += g->vapor_flux * 1000.;
        OD(GLAC_INFLOW_BAND, band) += g->inflow * 1000.;
        OD(GLAC_OUTFLOW_BAND, band) += g->outflow * 1000.;
      }
    }
    /* ---- special cases and derived variables, put_data.c:549-606 */
    if (cv_baresoil > 0) OD(BARESOILT, 0) /= cv_baresoil;
    if (cv_veg > 0) OD(VEGT, 0) /= cv_veg;
    if (cv_overstory > 0) OD(AERO_COND2, 0) /= cv_overstory;
    if (cv_snow > 0) { OD(SALBEDO, 0) /= cv_snow; OD(SNOW_SURF_TEMP, 0) /= cv_snow; OD(SNOW_PACK_TEMP, 0) /= cv_snow; }
    if (cv_glacier > 0) OD(GLAC_SURF_TEMP, 0) /= cv_glacier;
    OD(RAD_TEMP, 0) = pow(OD(RAD_TEMP, 0), 0.25);
    if (OD(AERO_COND1, 0) > ORC_SMALL) OD(AERO_RESIST1, 0) = 1 / OD(AERO_COND1, 0); else OD(AERO_RESIST1, 0) = ORC_HUGE_RESIST;
    if (OD(AERO_COND2, 0) > ORC_SMALL) OD(AERO_RESIST2, 0) = 1 / OD(AERO_COND2, 0); else OD(AERO_RESIST2, 0) = ORC_HUGE_RESIST;
    if (OD(AERO_COND, 0) > ORC_SMALL) OD(AERO_RESIST, 0) = 1 / OD(AERO_COND, 0); else OD(AERO_RESIST, 0) = ORC_HUGE_RESIST;
    OD(DELSOILMOIST, 0) = 0;
    for (l = 0; l < 3; l++) {
      OD(SOIL_LIQ_TOT, 0) += OD(SOIL_LIQ, l);
      OD(SOIL_ICE_TOT, 0) += OD(SOIL_ICE, l);
      OD(SOIL_MOIST, l) = OD(SOIL_LIQ, l) + OD(SOIL_ICE, l);
      OD(DELSOILMOIST, 0) += OD(SOIL_MOIST, l);
      OD(SMLIQFRAC, l) = OD(SOIL_LIQ, l) / OD(SOIL_MOIST, l);
      OD(SMFROZFRAC, l) = 1 - OD(SMLIQFRAC, l);
    }
    if (rec >= 0) {
      OD(DELSOILMOIST, 0) -= PB(PB_SAVE_TOTAL_SOIL_MOIST);
      OD(DELSWE, 0) = OD(SWE, 0) + OD(SNOW_CANOPY, 0) - PB(PB_SAVE_SWE);
      OD(DELINTERCEPT, 0) = OD(WDEW, 0) - PB(PB_SAVE_WDEW);
      OD(DELSURFSTOR, 0) = OD(SURFSTOR, 0) - PB(PB_SAVE_SURFSTOR);
    }
    OD(REFREEZE, 0) = (OD(RFRZ_ENERGY, 0) / ORC_LF) * dt_sec;
    OD(R_NET, 0) = OD(NET_SHORT, 0) + OD(NET_LONG, 0);
    PB(PB_SAVE_TOTAL_SOIL_MOIST) = 0;
    for (l = 0; l < 3; l++) PB(PB_SAVE_TOTAL_SOIL_MOIST) += OD(SOIL_MOIST, l);
    OD(SOIL_MOIST_TOT, 0) = PB(PB_SAVE_TOTAL_SOIL_MOIST);
    PB(PB_SAVE_SURFSTOR) = OD(SURFSTOR, 0);
    PB(PB_SAVE_SWE) = OD(SWE, 0) + OD(SNOW_CANOPY, 0);
    PB(PB_SAVE_WDEW) = OD(WDEW, 0);
    {                                                                               /* water balance, :611-619 + calc_water_balance_error */
      const double inflow = OD(PREC, 0) + 0.;
      const double outflow = OD(EVAP, 0) + OD(RUNOFF, 0) + OD(BASEFLOW, 0);
      const double glac_icebal = OD(GLAC_IMBAL, 0);
      double storage = 0., error;
      for (l = 0; l < 3; l++) storage += OD(SOIL_LIQ, l) + OD(SOIL_ICE, l);
      storage += OD(SWE, 0) + OD(SNOW_CANOPY, 0) + OD(WDEW, 0) + OD(SURFSTOR, 0) + OD(GLAC_WAT_STOR, 0);
      if (rec < 0) { PB(PB_WATER_LAST_STORAGE) = storage; PB(PB_WATER_CUM_ERROR) = 0.; PB(PB_WATER_MAX_ERROR) = 0.; OD(WATER_ERROR, 0) = 0.0; }
      else {
        error = inflow - outflow - (storage - PB(PB_WATER_LAST_STORAGE)) - glac_icebal;
        PB(PB_WATER_CUM_ERROR) += error;
        if (fabs(error) > fabs(PB(PB_WATER_MAX_ERROR)) && fabs(error) > 1e-5) PB(PB_WATER_MAX_ERROR) = error;
        PB(PB_WATER_LAST_STORAGE) = storage;
        OD(WATER_ERROR, 0) = error;
      }
    }
    if (o->FULL_ENERGY) {                                                           /* :624-633 + calc_energy_balance_error */
      if (rec < 0) { PB(PB_ENERGY_CUM_ERROR) = 0; PB(PB_ENERGY_MAX_ERROR) = 0; }
      else {
        const double net_rad = OD(NET_SHORT, 0) + OD(NET_LONG, 0), latent = OD(LATENT, 0) + OD(LATENT_SUB, 0),
                     sensible = OD(SENSIBLE, 0) + OD(ADV_SENS, 0), grnd = OD(GRND_FLUX, 0) + OD(DELTAH, 0) + OD(FUSION, 0),
                     snowf = OD(ADVECTION, 0) - OD(DELTACC, 0) - OD(SNOW_FLUX, 0) + OD(RFRZ_ENERGY, 0),
                     glacf = -OD(GLAC_DELTACC, 0) - OD(GLAC_MELT_ENERGY, 0);
        const double error = net_rad - latent - sensible - grnd + snowf + glacf;
        PB(PB_ENERGY_CUM_ERROR) += error;
        if (fabs(error) > fabs(PB(PB_ENERGY_MAX_ERROR)) && fabs(error) > 0.001) PB(PB_ENERGY_MAX_ERROR) = error;
      }
    }
    for (r = 0; r < nrow; r++) h->out_data[(size_t)r * nc + c] = od[r];
    if (rec < 0) continue;
    for (v = 0; v < VOUT_NVAR; v++) {                                               /* temporal aggregation, :663-682 */
      const int ne = kind_nelem(o, out_kind[v]);
      int i;
      for (i = 0; i < ne; i++) {
        double *ag = &h->out_agg[(size_t)(off[v] + i) * nc + c];
        if (out_agg[v] == VOUT_AGG_END) *ag = od[off[v] + i];
        else if (out_agg[v] == VOUT_AGG_SUM) *ag += od[off[v] + i];
        else *ag += od[off[v] + i] / out_step_ratio;
      }
    }
    h->out_agg[(size_t)off[VOUT_AERO_RESIST] * nc + c] = 1 / h->out_agg[(size_t)off[VOUT_AERO_COND] * nc + c];
    h->out_agg[(size_t)off[VOUT_AERO_RESIST1] * nc + c] = 1 / h->out_agg[(size_t)off[VOUT_AERO_COND1] * nc + c];
    h->out_agg[(size_t)off[VOUT_AERO_RESIST2] * nc + c] = 1 / h->out_agg[(size_t)off[VOUT_AERO_COND2] * nc + c];
  }
#undef OD
#undef PB
  return 0;
}

/* which: 0 = OutputData.data of the last call, 1 = .aggdata; out [nelem][ncell] */
int vicorc_get_output(void *hv, int id, int which, double *out) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const vicgpu_options *o = &h->model.opt;
  int ne;
  if (!h->out_data || id < 0 || id >= VOUT_NVAR) return -1;
  ne = kind_nelem(o, out_kind[id]);
  memcpy(out, (which ? h->out_agg : h->out_data) + (size_t)row0(o, id) * h->ncell, sizeof(double) * (size_t)ne * h->ncell);
  return ne;
}

int vicorc_reset_agg(void *hv) {                                                   /* vicNl.c:599-606 */
  vicorc_handle *h = (vicorc_handle *)hv;
  if (h->out_agg) memset(h->out_agg, 0, sizeof(double) * (size_t)nrow_total(&h->model.opt) * h->ncell);
  return 0;
}

int vicorc_get_balance(void *hv, double *pb) {
  vicorc_handle *h = (vicorc_handle *)hv;
  if (!h->pb) return -1;
  memcpy(pb, h->pb, sizeof(double) * (size_t)PB_NROW * h->ncell);
  return 0;
}
