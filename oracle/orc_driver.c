/*
 * orc_driver.c — TEST INFRASTRUCTURE (CPU oracle): full_energy / surface_fluxes restatement,
 * table <-> struct conversion and the vicorc_* C entry points used by tests/ and bench.py's
 * cpu_baseline leg.  Never linked into the product.
 */
#include "orc.h"
#include <stdlib.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void vc_nan(orc_vc *v) { int k; for (k = 0; k < ORC_NCASE; k++) v->v[k] = NAN; }

/* prepare_full_energy.c:8-94 */
static void orc_prepare_full_energy(const orc_model *m, orc_hru *h, const orc_soil *sc, double *moist0, double *ice0) {
  if (sc->AreaFract[h->band] > 0.0) {
    orc_layer layer[3];
    int l;
    for (l = 0; l < 3; l++) layer[l] = h->layer[l];
    *moist0 = layer[0].moist / (sc->depth[0] * 1000.);
    if (m->opt.FROZEN_SOIL && sc->FS_ACTIVE) {
      if ((h->energy.T[0] + h->energy.T[1]) / 2. < 0.) {
        *ice0 = *moist0 - orc_maximum_unfrozen_water((h->energy.T[0] + h->energy.T[1]) / 2.,
                                                     sc->max_moist[0] / (sc->depth[0] * 1000.), sc->bubble[0], sc->expt[0]);
        if (*ice0 < 0.) *ice0 = 0.;
      } else *ice0 = 0.;
    } else *ice0 = 0.;
    orc_layer_thermal_properties(layer, sc);
    h->energy.kappa[0] = layer[0].kappa;
    h->energy.Cs[0] = layer[0].Cs;
    h->energy.kappa[1] = layer[1].kappa;
    h->energy.Cs[1] = layer[1].Cs;
  } else *ice0 = 0.;
}

/* surface_fluxes.c:17-956 with CLOSE_ENERGY FALSE (MAX_ITER 0: both closure loops run once), Ndist 1 */
static int orc_surface_fluxes(const orc_model *m, orc_hru *h, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy,
                              int overstory, double BareAlbedo, double ice0, double moist0, double surf_atten,
                              orc_vc *aero_resist /*[7]*/, orc_vc *displacement, orc_vc *ref_height, orc_vc *roughness,
                              orc_vc *wind_speed, double *out_prec, double *out_rain, double *out_snow) {
  const int NF = m->NF, NR = m->NR, Nnode = m->opt.Nnode;
  orc_energy *energy = &h->energy;
  orc_snow *snow = &h->snow;
  int INCLUDE_SNOW = 0, UNSTABLE_SNOW = 0, N_steps = 0, UnderStory, hidx, endhidx, step_dt, l, p;
  double Le = 0, LongUnderIn, LongUnderOut, NetLongSnow, NetShortSnow, NetShortGrnd, OldTSurf, ShortUnderIn, Tair, Tcanopy,
         Tgrnd, Tsurf, VPDcanopy, VPcanopy, coverage, delta_coverage, last_snow_coverage, ppt, rainfall, snowfall, snow_flux,
         snow_grnd_flux, snow_inflow = 0, step_Wdew, step_melt, step_melt_energy, step_out_prec, step_out_rain, step_out_snow,
         step_ppt, step_prec;
  double ra_s[ORC_NPET], ra_o[ORC_NPET], iter_ra_used[2], stability_factor[2], iter_pot_evap[ORC_NPET], store_pot_evap[ORC_NPET];
  double st_AlbedoOver = 0, st_AlbedoUnder = 0, st_AtmosLatent = 0, st_AtmosLatentSub = 0, st_AtmosSensible = 0,
         st_LongOverIn = 0, st_LongUnderIn = 0, st_LongUnderOut = 0, st_NetLongAtmos = 0, st_NetLongOver = 0,
         st_NetLongUnder = 0, st_NetShortAtmos = 0, st_NetShortGrnd = 0, st_NetShortOver = 0, st_NetShortUnder = 0,
         st_ShortOverIn = 0, st_ShortUnderIn = 0, st_advected_sensible = 0, st_advection = 0, st_canopy_advection = 0,
         st_canopy_latent = 0, st_canopy_latent_sub = 0, st_canopy_sensible = 0, st_canopy_refreeze = 0, st_deltaCC = 0,
         st_deltaH = 0, st_fusion = 0, st_grnd_flux = 0, st_latent = 0, st_latent_sub = 0, st_melt_energy = 0,
         st_refreeze_energy = 0, st_sensible = 0, st_snow_flux = 0, st_canopy_vapor_flux = 0, st_melt = 0, st_vapor_flux = 0,
         st_blowing_flux = 0, st_surface_flux = 0, st_canopyevap = 0, st_throughfall = 0, st_layerevap[3] = {0, 0, 0},
         st_ppt = 0, st_cond_surface = 0, st_cond_overstory = 0;
  orc_energy snow_energy, soil_energy, iter_snow_energy, iter_soil_energy;
  orc_vegvar snow_vv, soil_vv, iter_snow_vv, iter_soil_vv;
  orc_snow step_snow, iter_snow;
  orc_layer step_layer[3], iter_layer[3];
  orc_vc iter_aero_resist;

  energy->advection = 0;
  energy->deltaCC = 0;
  if (snow->swq > 0) snow_flux = energy->snow_flux;
  else snow_flux = -(energy->grnd_flux + energy->deltaH + energy->fusion);
  energy->refreeze_energy = 0;
  coverage = snow->coverage;
  snow_energy = *energy;
  soil_energy = *energy;
  snow_vv = h->veg; soil_vv = h->veg;
  step_snow = *snow;
  for (l = 0; l < 3; l++) { step_layer[l] = h->layer[l]; step_layer[l].evap = 0; }
  soil_vv.canopyevap = 0; snow_vv.canopyevap = 0; soil_vv.throughfall = 0; snow_vv.throughfall = 0;

  if (snow->swq > 0 || snow->snow_canopy > 0 || atmos->snowflag[NR]) {             /* :331-341 */
    hidx = 0; endhidx = hidx + NF; step_dt = m->opt.snow_step;
  } else {
    hidx = NR; endhidx = hidx + 1; step_dt = m->opt.dt;
  }
  last_snow_coverage = snow->coverage;
  step_Wdew = h->veg.Wdew;
  for (p = 0; p < ORC_NPET; p++) store_pot_evap[p] = 0;

  do {
    Tair = atmos->air_temp[hidx] + sc->Tfactor[h->band];
    step_prec = atmos->prec[hidx] / 1.0 * sc->Pfactor[h->band];
    Tgrnd = energy->T[0];
    Tcanopy = Tair;
    VPcanopy = atmos->vp[hidx];
    VPDcanopy = atmos->vpd[hidx];
    if (!overstory && m->opt.BLOWING && step_snow.swq > 0.) {                      /* surface_fluxes.c:439-453 */
      double Ls = (677. - 0.07 * step_snow.surf_temp) * 4.1868 * 1000.0;
      step_snow.blowing_flux = orc_calc_blowing_snow((double)step_dt, Tair, step_snow.last_snow, step_snow.surf_water,
                                                     wind_speed->v[ORC_SNOW_COVERED], Ls, atmos->density[hidx], atmos->vp[hidx],
                                                     roughness->v[ORC_SNOW_COVERED], ref_height->v[ORC_SNOW_COVERED], step_snow.depth,
                                                     h->lag_one, h->sigma_slope, step_snow.surf_temp, h->is_artificial_bare, h->fetch,
                                                     displacement->v[ORC_CANOPY], roughness->v[ORC_CANOPY], &step_snow.transport);
      if ((int)step_snow.blowing_flux == ORC_ERROR) return -1;
      step_snow.blowing_flux *= step_dt * ORC_SECPHOUR / ORC_RHO_W;
    } else step_snow.blowing_flux = 0.0;
    UnderStory = ORC_NCASE;
    snow_grnd_flux = -snow_flux;

    iter_snow_energy = snow_energy;
    iter_soil_energy = soil_energy;
    iter_snow_vv = snow_vv; iter_soil_vv = soil_vv;
    iter_snow = step_snow;
    for (l = 0; l < 3; l++) iter_layer[l] = step_layer[l];
    iter_snow_vv.Wdew = step_Wdew; iter_soil_vv.Wdew = step_Wdew;
    iter_snow_vv.canopyevap = 0; iter_soil_vv.canopyevap = 0;
    for (l = 0; l < 3; l++) iter_layer[l].evap = 0;
    iter_aero_resist = aero_resist[ORC_NPET];
    iter_ra_used[0] = h->aero_resist_surface;
    iter_ra_used[1] = h->aero_resist_overstory;
    iter_snow.canopy_vapor_flux = 0;
    iter_snow.vapor_flux = 0;
    iter_snow.surface_flux = 0;
    LongUnderOut = iter_soil_energy.LongUnderOut;

    step_melt = orc_solve_snow(m, overstory, BareAlbedo, LongUnderOut, Tcanopy, Tgrnd, Tair, step_prec, snow_grnd_flux,
                               &energy->AlbedoUnder, &Le, &LongUnderIn, &NetLongSnow, &NetShortGrnd, &NetShortSnow,
                               &ShortUnderIn, &OldTSurf, &iter_aero_resist, iter_ra_used, &coverage, &delta_coverage,
                               displacement, &step_melt_energy, &step_out_prec, &step_out_rain, &step_out_snow, &step_ppt,
                               &rainfall, ref_height, roughness, &snow_inflow, &snowfall, &surf_atten, wind_speed, h->root,
                               UNSTABLE_SNOW, step_dt, hidx, h->veg_index, h->is_artificial_bare, &UnderStory, dmy, atmos,
                               &iter_snow_energy, iter_layer, &iter_snow, sc, &iter_snow_vv);
    if (step_melt == ORC_ERROR) return -1;

    if ((isnan(iter_snow.surf_temp) || UNSTABLE_SNOW) && iter_snow.swq > 0) {      /* :553-560 */
      INCLUDE_SNOW = UnderStory + 1;
      iter_soil_energy.advection = iter_snow_energy.advection;
      iter_snow.surf_temp = step_snow.surf_temp;
      step_melt_energy = 0;
    } else INCLUDE_SNOW = 0;

    Tsurf = orc_calc_surf_energy_bal(m, Le, LongUnderIn, NetLongSnow, NetShortGrnd, NetShortSnow, OldTSurf, ShortUnderIn,
                                     iter_snow.albedo, iter_snow_energy.latent, iter_snow_energy.latent_sub,
                                     iter_snow_energy.sensible, Tcanopy, VPDcanopy, VPcanopy, delta_coverage, sc->dp, ice0,
                                     step_melt_energy, moist0, iter_snow.coverage, (step_snow.depth + iter_snow.depth) / 2.,
                                     BareAlbedo, surf_atten, &iter_aero_resist, iter_ra_used, displacement, &step_melt,
                                     &step_ppt, &rainfall, ref_height, roughness, wind_speed, h->root, INCLUDE_SNOW,
                                     UnderStory, Nnode, step_dt, hidx, overstory, h->veg_index, h->is_artificial_bare,
                                     atmos, dmy, &iter_soil_energy, iter_layer, &iter_snow, sc, &iter_soil_vv);
    if ((int)Tsurf == (int)ORC_ERROR) return -1;
    if (INCLUDE_SNOW) step_ppt += step_melt;

    iter_soil_energy.AtmosLatent = iter_soil_energy.latent;                        /* :618-626 (MAX_ITER == 0 branch) */
    iter_soil_energy.AtmosLatentSub = iter_soil_energy.latent_sub;
    iter_soil_energy.AtmosSensible = iter_soil_energy.sensible;
    iter_soil_energy.NetLongAtmos = iter_soil_energy.NetLongUnder;
    iter_soil_energy.NetShortAtmos = iter_soil_energy.NetShortUnder;
    iter_soil_energy.Tcanopy = Tcanopy;
    iter_snow_energy.Tcanopy = Tcanopy;

    /* potential evaporation :658-693 */
    if (iter_ra_used[0] == ORC_HUGE_RESIST) stability_factor[0] = ORC_HUGE_RESIST;
    else stability_factor[0] = iter_ra_used[0] / aero_resist[ORC_NPET].v[UnderStory];
    if (iter_ra_used[1] == iter_ra_used[0]) stability_factor[1] = stability_factor[0];
    else {
      if (iter_ra_used[1] == ORC_HUGE_RESIST) stability_factor[1] = ORC_HUGE_RESIST;
      else stability_factor[1] = iter_ra_used[1] / aero_resist[ORC_NPET].v[ORC_CANOPY];
    }
    for (p = 0; p < ORC_NPET; p++) {
      if (stability_factor[0] == ORC_HUGE_RESIST) ra_s[p] = ORC_HUGE_RESIST;
      else ra_s[p] = aero_resist[p].v[UnderStory] * stability_factor[0];
      if (stability_factor[1] == ORC_HUGE_RESIST) ra_o[p] = ORC_HUGE_RESIST;
      else ra_o[p] = aero_resist[p].v[ORC_CANOPY] * stability_factor[1];
    }
    orc_compute_pot_evap(m, h->veg_index, dmy->month, m->opt.dt, atmos->shortwave[hidx], iter_soil_energy.NetLongAtmos, Tair,
                         VPDcanopy, sc->elevation, ra_s, ra_o, iter_pot_evap);

    /* store sub-step :699-816 */
    snow_energy = iter_snow_energy;
    soil_energy = iter_soil_energy;
    snow_vv = iter_snow_vv; soil_vv = iter_soil_vv;
    step_snow = iter_snow;
    for (l = 0; l < 3; l++) step_layer[l] = iter_layer[l];
    if (!h->is_artificial_bare) {
      if (step_snow.snow) {
        st_throughfall += snow_vv.throughfall;
        st_canopyevap += snow_vv.canopyevap;
        soil_vv.Wdew = snow_vv.Wdew;
      } else {
        st_throughfall += soil_vv.throughfall;
        st_canopyevap += soil_vv.canopyevap;
        snow_vv.Wdew = soil_vv.Wdew;
      }
      step_Wdew = soil_vv.Wdew;
    }
    for (l = 0; l < 3; l++) st_layerevap[l] += step_layer[l].evap;
    st_ppt += step_ppt;
    if (iter_ra_used[0] > 0) st_cond_surface += 1 / iter_ra_used[0]; else st_cond_surface += ORC_HUGE_RESIST;
    if (iter_ra_used[1] > 0) st_cond_overstory += 1 / iter_ra_used[1]; else st_cond_overstory += ORC_HUGE_RESIST;
    if (!h->is_artificial_bare) st_canopy_vapor_flux += step_snow.canopy_vapor_flux;
    st_melt += step_melt;
    st_vapor_flux += step_snow.vapor_flux;
    st_surface_flux += step_snow.surface_flux;
    st_blowing_flux += step_snow.blowing_flux;
    out_prec[0] += step_out_prec * 1.0;
    out_rain[0] += step_out_rain * 1.0;
    out_snow[0] += step_out_snow * 1.0;
    if (INCLUDE_SNOW) {
      snow_energy.advected_sensible = soil_energy.advected_sensible;
      snow_energy.advection = soil_energy.advection;
      snow_energy.deltaCC = soil_energy.deltaCC;
      snow_energy.latent = soil_energy.latent;
      snow_energy.latent_sub = soil_energy.latent_sub;
      snow_energy.refreeze_energy = soil_energy.refreeze_energy;
      snow_energy.sensible = soil_energy.sensible;
      snow_energy.snow_flux = soil_energy.snow_flux;
    }
    st_AlbedoOver += snow_energy.AlbedoOver;
    st_AlbedoUnder += soil_energy.AlbedoUnder;
    st_AtmosLatent += soil_energy.AtmosLatent;
    st_AtmosLatentSub += soil_energy.AtmosLatentSub;
    st_AtmosSensible += soil_energy.AtmosSensible;
    st_LongOverIn += snow_energy.LongOverIn;
    st_LongUnderIn += LongUnderIn;
    st_LongUnderOut += soil_energy.LongUnderOut;
    st_NetLongAtmos += soil_energy.NetLongAtmos;
    st_NetLongOver += snow_energy.NetLongOver;
    st_NetLongUnder += soil_energy.NetLongUnder;
    st_NetShortAtmos += soil_energy.NetShortAtmos;
    st_NetShortGrnd += NetShortGrnd;
    st_NetShortOver += snow_energy.NetShortOver;
    st_NetShortUnder += soil_energy.NetShortUnder;
    st_ShortOverIn += snow_energy.ShortOverIn;
    st_ShortUnderIn += soil_energy.ShortUnderIn;
    st_canopy_advection += snow_energy.canopy_advection;
    st_canopy_latent += snow_energy.canopy_latent;
    st_canopy_latent_sub += snow_energy.canopy_latent_sub;
    st_canopy_sensible += snow_energy.canopy_sensible;
    st_canopy_refreeze += snow_energy.canopy_refreeze;
    st_deltaH += soil_energy.deltaH;
    st_fusion += soil_energy.fusion;
    st_grnd_flux += soil_energy.grnd_flux;
    st_latent += soil_energy.latent;
    st_latent_sub += soil_energy.latent_sub;
    st_melt_energy += step_melt_energy;
    st_sensible += soil_energy.sensible;
    if (step_snow.swq == 0 && INCLUDE_SNOW) {
      if (last_snow_coverage == 0) last_snow_coverage = 1;                          /* pointer test is always true, Appendix C #5 */
      st_advected_sensible += snow_energy.advected_sensible * last_snow_coverage;
      st_advection += snow_energy.advection * last_snow_coverage;
      st_deltaCC += snow_energy.deltaCC * last_snow_coverage;
      st_snow_flux += soil_energy.snow_flux * last_snow_coverage;
      st_refreeze_energy += snow_energy.refreeze_energy * last_snow_coverage;
    } else if (step_snow.snow || INCLUDE_SNOW) {
      st_advected_sensible += snow_energy.advected_sensible * (step_snow.coverage + delta_coverage);
      st_advection += snow_energy.advection * (step_snow.coverage + delta_coverage);
      st_deltaCC += snow_energy.deltaCC * (step_snow.coverage + delta_coverage);
      st_snow_flux += soil_energy.snow_flux * (step_snow.coverage + delta_coverage);
      st_refreeze_energy += snow_energy.refreeze_energy * (step_snow.coverage + delta_coverage);
    }
    for (p = 0; p < ORC_NPET; p++) store_pot_evap[p] += iter_pot_evap[p];
    N_steps++;
    hidx += 1;
  } while (hidx < endhidx);

  *snow = step_snow;                                                               /* :828-836 */
  snow->vapor_flux = st_vapor_flux;
  snow->blowing_flux = st_blowing_flux;
  snow->surface_flux = st_surface_flux;
  snow->canopy_vapor_flux = st_canopy_vapor_flux;
  snow->melt = st_melt;
  ppt = st_ppt;

  *energy = soil_energy;                                                           /* :842-881 */
  energy->AlbedoOver = st_AlbedoOver / (double)N_steps;
  energy->AlbedoUnder = st_AlbedoUnder / (double)N_steps;
  energy->AtmosLatent = st_AtmosLatent / (double)N_steps;
  energy->AtmosLatentSub = st_AtmosLatentSub / (double)N_steps;
  energy->AtmosSensible = st_AtmosSensible / (double)N_steps;
  energy->LongOverIn = st_LongOverIn / (double)N_steps;
  energy->LongUnderIn = st_LongUnderIn / (double)N_steps;
  energy->LongUnderOut = st_LongUnderOut / (double)N_steps;
  energy->NetLongAtmos = st_NetLongAtmos / (double)N_steps;
  energy->NetLongOver = st_NetLongOver / (double)N_steps;
  energy->NetLongUnder = st_NetLongUnder / (double)N_steps;
  energy->NetShortAtmos = st_NetShortAtmos / (double)N_steps;
  energy->NetShortGrnd = st_NetShortGrnd / (double)N_steps;
  energy->NetShortOver = st_NetShortOver / (double)N_steps;
  energy->NetShortUnder = st_NetShortUnder / (double)N_steps;
  energy->ShortOverIn = st_ShortOverIn / (double)N_steps;
  energy->ShortUnderIn = st_ShortUnderIn / (double)N_steps;
  energy->advected_sensible = st_advected_sensible / (double)N_steps;
  energy->canopy_advection = st_canopy_advection / (double)N_steps;
  energy->canopy_latent = st_canopy_latent / (double)N_steps;
  energy->canopy_latent_sub = st_canopy_latent_sub / (double)N_steps;
  energy->canopy_refreeze = st_canopy_refreeze / (double)N_steps;
  energy->canopy_sensible = st_canopy_sensible / (double)N_steps;
  energy->deltaH = st_deltaH / (double)N_steps;
  energy->fusion = st_fusion / (double)N_steps;
  energy->grnd_flux = st_grnd_flux / (double)N_steps;
  energy->latent = st_latent / (double)N_steps;
  energy->latent_sub = st_latent_sub / (double)N_steps;
  energy->melt_energy = st_melt_energy / (double)N_steps;
  energy->sensible = st_sensible / (double)N_steps;
  if (snow->snow || INCLUDE_SNOW) {
    energy->advection = st_advection / (double)N_steps;
    energy->deltaCC = st_deltaCC / (double)N_steps;
    energy->refreeze_energy = st_refreeze_energy / (double)N_steps;
    energy->snow_flux = st_snow_flux / (double)N_steps;
  }
  energy->Tfoliage = snow_energy.Tfoliage;
  energy->Tfoliage_fbflag = snow_energy.Tfoliage_fbflag;
  energy->Tfoliage_fbcount = snow_energy.Tfoliage_fbcount;

  if (!h->is_artificial_bare) {                                                    /* :889-901 */
    h->veg.throughfall = st_throughfall;
    h->veg.canopyevap = st_canopyevap;
    if (snow->snow) h->veg.Wdew = snow_vv.Wdew; else h->veg.Wdew = soil_vv.Wdew;
  }
  for (l = 0; l < 3; l++) { h->layer[l] = step_layer[l]; h->layer[l].evap = st_layerevap[l]; }
  if (st_cond_surface > 0 && st_cond_surface < ORC_HUGE_RESIST) h->aero_resist_surface = 1 / (st_cond_surface / (double)N_steps);
  else if (st_cond_surface >= ORC_HUGE_RESIST) h->aero_resist_surface = 0;
  else h->aero_resist_surface = ORC_HUGE_RESIST;
  if (st_cond_overstory > 0 && st_cond_overstory < ORC_HUGE_RESIST) h->aero_resist_overstory = 1 / (st_cond_overstory / (double)N_steps);
  else if (st_cond_overstory >= ORC_HUGE_RESIST) h->aero_resist_overstory = 0;
  else h->aero_resist_overstory = ORC_HUGE_RESIST;
  for (p = 0; p < ORC_NPET; p++) h->pot_evap[p] = store_pot_evap[p] / (double)N_steps;

  ppt += h->excess_moist;                                                          /* :941-948 */
  h->excess_moist = 0.;
  h->inflow = ppt;
  return orc_runoff(m, h, sc, ppt);
}

/* full_energy.c:8-498 for one cell (no lakes, no EXCESS_ICE) */
int orc_full_energy(const orc_model *m, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy, orc_hru **hrus, int nhru) {
  /* correct_precip.c:9-53, applied as in full_energy.c:188-194 */
  atmos->gauge_correction[0] = 1; atmos->gauge_correction[1] = 1;
  if (m->opt.CORRPREC && atmos->prec[m->NR] > 0) {
    const double GAUGE_HEIGHT = 1.0;
    double gauge_wind = atmos->wind[m->NR] * (log((GAUGE_HEIGHT + sc->rough) / sc->rough) / log(m->opt.wind_h / sc->rough));
    atmos->gauge_correction[0] = 100. / exp(4.606 - 0.041 * pow(gauge_wind, 0.69));
    gauge_wind = atmos->wind[m->NR] * (log((GAUGE_HEIGHT + sc->snow_rough) / sc->snow_rough) / log(m->opt.wind_h / sc->snow_rough));
    atmos->gauge_correction[1] = 100. / exp(4.606 - 0.036 * pow(gauge_wind, 1.75));
  }
  const int NR = m->NR, month = dmy->month;
  orc_vc displacement, roughness, ref_height, wind_speed, aero_resist[ORC_NPET + 1];
  int k, p, l;
  vc_nan(&displacement); vc_nan(&roughness); vc_nan(&ref_height); vc_nan(&wind_speed);
  for (p = 0; p <= ORC_NPET; p++) vc_nan(&aero_resist[p]);
  atmos->out_prec = 0; atmos->out_rain = 0; atmos->out_snow = 0;
  for (k = 0; k < nhru; k++) {
    orc_hru *h = hrus[k];
    const double *vl;
    double Cv, wind_h, surf_atten, bare_albedo, height = 0, moist0 = 0, ice0 = 0, out_prec = 0, out_rain = 0, out_snow = 0;
    int overstory = 0, err;
    h->out_prec = h->out_rain = h->out_snow = 0;
    if (!((h->Cv > 0.0) || (h->is_glacier && m->opt.GLACIER_DYNAMICS && h->Cv >= 0.0))) continue;
    Cv = h->Cv;
    if (sc->AreaFract[h->band] > 0) { h->snow.vapor_flux = 0.; h->snow.canopy_vapor_flux = 0.; }
    vl = orc_veg(m, h->veg_index);
    wind_h = vl[VL_WIND_H];
    surf_atten = exp(-vl[VL_RAD_ATTEN] * vl[VL_LAI + month - 1]);
    orc_prepare_full_energy(m, h, sc, &moist0, &ice0);
    if (h->is_glacier) bare_albedo = sc->GLAC_ALBEDO;
    else bare_albedo = vl[VL_ALBEDO + month - 1];
    for (p = 0; p < ORC_NPET + 1; p++) {                                           /* :302-354 */
      int pet_idx = (p < ORC_NPET_NON_NAT) ? m->opt.nveg_types + p : h->veg_index;
      const double *pv = orc_veg(m, pet_idx);
      double tmp_z0, tmp_d, tmp_zref, wind_corr;
      if (pet_idx == m->opt.GLACIER_ID) roughness.v[ORC_SNOW_FREE] = sc->GLAC_ROUGH;   /* sic: index vs class id */
      else roughness.v[ORC_SNOW_FREE] = pv[VL_ROUGHNESS + month - 1];
      displacement.v[ORC_SNOW_FREE] = pv[VL_DISPLACEMENT + month - 1];
      overstory = (pv[VL_OVERSTORY] != 0);
      if (p >= ORC_NPET_NON_NAT)
        if (roughness.v[ORC_SNOW_FREE] == 0) roughness.v[ORC_SNOW_FREE] = sc->rough;
      height = orc_calc_veg_height(displacement.v[ORC_SNOW_FREE], vl[VL_LAI + month - 1]);
      if (displacement.v[ORC_SNOW_FREE] < wind_h) ref_height.v[ORC_SNOW_FREE] = wind_h;
      else ref_height.v[ORC_SNOW_FREE] = displacement.v[ORC_SNOW_FREE] + wind_h + roughness.v[ORC_SNOW_FREE];
      tmp_z0 = sc->rough; tmp_d = 0.; tmp_zref = m->opt.wind_h;
      wind_corr = log((ref_height.v[ORC_SNOW_FREE] - tmp_d) / tmp_z0) / log((tmp_zref - tmp_d) / tmp_z0);
      wind_speed.v[ORC_SNOW_FREE] = atmos->wind[NR] * wind_corr;
      wind_speed.v[ORC_CANOPY] = NAN; wind_speed.v[ORC_SNOW_COVERED] = NAN; wind_speed.v[ORC_GLACIER_SURF] = NAN;
      err = orc_calc_aerodynamic(overstory, height, pv[VL_TRUNK_RATIO], sc->snow_rough, sc->rough, pv[VL_WIND_ATTEN],
                                 &aero_resist[p], &wind_speed, &displacement, &ref_height, &roughness);
      if (err) return VICGPU_CELLERR_AERO;
    }
    if (sc->AreaFract[h->band] > 0) {
      h->aero_resist_surface = aero_resist[ORC_NPET].v[ORC_SNOW_FREE];
      h->aero_resist_overstory = aero_resist[ORC_NPET].v[ORC_CANOPY];
    }
    if ((sc->AreaFract[h->band] > 0) || (h->is_glacier && m->opt.GLACIER_DYNAMICS && sc->AreaFract[h->band] >= 0.0)) {
      for (p = 0; p < ORC_NPET; p++) h->pot_evap[p] = 0;
      if (h->is_glacier)
        err = orc_surface_fluxes_glac(m, h, sc, atmos, dmy, bare_albedo, ice0, moist0, aero_resist, &displacement,
                                      &ref_height, &roughness, &wind_speed, &out_prec, &out_rain, &out_snow);
      else
        err = orc_surface_fluxes(m, h, sc, atmos, dmy, overstory, bare_albedo, ice0, moist0, surf_atten, aero_resist,
                                 &displacement, &ref_height, &roughness, &wind_speed, &out_prec, &out_rain, &out_snow);
      if (err) return VICGPU_CELLERR_SOLVER;
      h->out_prec = out_prec; h->out_rain = out_rain; h->out_snow = out_snow;
      atmos->out_prec += out_prec * Cv;
      atmos->out_rain += out_rain * Cv;
      atmos->out_snow += out_snow * Cv;
      h->rootmoist = 0;
      h->wetness = 0;
      for (l = 0; l < 3; l++) {
        if (h->root[l] > 0) h->rootmoist += h->layer[l].moist;
        h->wetness += (h->layer[l].moist - sc->Wpwp[l]) / (sc->porosity[l] * sc->depth[l] * 1000 - sc->Wpwp[l]);
      }
      h->wetness /= 3;
    }
  }
  return 0;
}

/* ============================================================ table <-> struct and the C entry points */


#define CPV(row) cp[(size_t)(row) * ncell + c]

void *vicorc_create(const vicgpu_options *opt) {
  vicorc_handle *h;
  if (!opt || opt->abi_version != VICGPU_ABI_VERSION || opt->Nlayer != 3 || opt->Nnode > VIC_MAX_NODES || opt->Nnode < 3
      || opt->Nband > VIC_MAX_BANDS || opt->dt / opt->snow_step + 1 > ORC_MAX_SUB)
    return NULL;
  h = (vicorc_handle *)calloc(1, sizeof(*h));
  h->model.opt = *opt;
  h->model.NF = VICGPU_NF(opt);
  h->model.NR = VICGPU_NR(opt);
  h->model.node_macheps = 3e-8; h->model.node_ttol = 1e-7;                        /* root_brent.c:32-36 */
  return h;
}

void vicorc_destroy(void *hv) {
  vicorc_handle *h = (vicorc_handle *)hv;
  if (!h) return;
  free(h->veglib); free(h->soil); free(h->hru); free(h->cell_off); free(h->cell_list);
  free(h->out_data); free(h->out_agg); free(h->pb); free(h);
}

int vicorc_set_veglib(void *hv, int nrow, const double *t) {
  vicorc_handle *h = (vicorc_handle *)hv;
  if (nrow != h->model.opt.nveg_types + 4) return -1;
  free(h->veglib);
  h->veglib = (double *)malloc(sizeof(double) * nrow * VL_NFIELD);
  memcpy(h->veglib, t, sizeof(double) * nrow * VL_NFIELD);
  h->model.veglib = h->veglib;
  h->model.nveg_rows = nrow;
  return 0;
}

static void hru_defaults(orc_hru *u) {
  memset(u, 0, sizeof(*u));
  u->glac.cold_content = NAN; u->glac.Qnet = NAN; u->glac.mass_balance = NAN; u->glac.cum_mass_balance = NAN;
  u->glac.accumulation = NAN; u->glac.melt = NAN; u->glac.vapor_flux = NAN; u->glac.outflow = NAN;
  u->glac.outflow_coef = NAN; u->glac.inflow = NAN;   /* glac_data_struct ctor, vicNl_def.h:1344-1348 */
  u->aero_resist_surface = NAN; u->aero_resist_overstory = NAN;
}

int vicorc_set_domain(void *hv, int ncell, int nhru, const double *cp, const int *hpi, const double *hpd,
                      const int *cell_off, const int *cell_list) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const int Nn = h->model.opt.Nnode, Nb = h->model.opt.Nband;
  int c, g, l, n, b, i;
  h->ncell = ncell; h->nhru = nhru;
  free(h->soil); free(h->hru); free(h->cell_off); free(h->cell_list);
  h->soil = (orc_soil *)calloc(ncell, sizeof(orc_soil));
  h->hru = (orc_hru *)calloc(nhru, sizeof(orc_hru));
  h->cell_off = (int *)malloc(sizeof(int) * (ncell + 1));
  h->cell_list = (int *)malloc(sizeof(int) * nhru);
  memcpy(h->cell_off, cell_off, sizeof(int) * (ncell + 1));
  memcpy(h->cell_list, cell_list, sizeof(int) * nhru);
  for (c = 0; c < ncell; c++) {
    orc_soil *s = &h->soil[c];
    s->Ds = CPV(CP_DS); s->Dsmax = CPV(CP_DSMAX); s->Ws = CPV(CP_WS); s->c = CPV(CP_C); s->b_infilt = CPV(CP_B_INFILT);
    s->dp = CPV(CP_DP); s->avg_temp = CPV(CP_AVG_TEMP); s->rough = CPV(CP_ROUGH); s->snow_rough = CPV(CP_SNOW_ROUGH);
    s->elevation = (double)(float)CPV(CP_ELEVATION); s->lat = (double)(float)CPV(CP_LAT); s->FS_ACTIVE = (int)CPV(CP_FS_ACTIVE);
    s->NEW_SNOW_ALB = CPV(CP_NEW_SNOW_ALB); s->SNOW_ALB_ACCUM_A = CPV(CP_SNOW_ALB_ACCUM_A);
    s->SNOW_ALB_ACCUM_B = CPV(CP_SNOW_ALB_ACCUM_B); s->SNOW_ALB_THAW_A = CPV(CP_SNOW_ALB_THAW_A);
    s->SNOW_ALB_THAW_B = CPV(CP_SNOW_ALB_THAW_B); s->MIN_RAIN_TEMP = CPV(CP_MIN_RAIN_TEMP);
    s->MAX_SNOW_TEMP = CPV(CP_MAX_SNOW_TEMP); s->PADJ_R = CPV(CP_PADJ_R); s->PADJ_S = CPV(CP_PADJ_S);
    s->GLAC_SURF_THICK = CPV(CP_GLAC_SURF_THICK); s->GLAC_SURF_WE = CPV(CP_GLAC_SURF_WE); s->GLAC_KMIN = CPV(CP_GLAC_KMIN);
    s->GLAC_DK = CPV(CP_GLAC_DK); s->GLAC_A = CPV(CP_GLAC_A); s->GLAC_ALBEDO = CPV(CP_GLAC_ALBEDO);
    s->GLAC_ROUGH = CPV(CP_GLAC_ROUGH);
    for (l = 0; l < 3; l++) {
      s->Ksat[l] = CPV(VICGPU_CP_LAYER(CPL_KSAT, l)); s->Wcr[l] = CPV(VICGPU_CP_LAYER(CPL_WCR, l));
      s->Wpwp[l] = CPV(VICGPU_CP_LAYER(CPL_WPWP, l)); s->expt[l] = CPV(VICGPU_CP_LAYER(CPL_EXPT, l));
      s->bubble[l] = CPV(VICGPU_CP_LAYER(CPL_BUBBLE, l)); s->depth[l] = CPV(VICGPU_CP_LAYER(CPL_DEPTH, l));
      s->max_moist[l] = CPV(VICGPU_CP_LAYER(CPL_MAX_MOIST, l)); s->resid_moist[l] = CPV(VICGPU_CP_LAYER(CPL_RESID_MOIST, l));
      s->porosity[l] = CPV(VICGPU_CP_LAYER(CPL_POROSITY, l)); s->quartz[l] = CPV(VICGPU_CP_LAYER(CPL_QUARTZ, l));
      s->organic[l] = CPV(VICGPU_CP_LAYER(CPL_ORGANIC, l)); s->bulk_density[l] = CPV(VICGPU_CP_LAYER(CPL_BULK_DENSITY, l));
      s->soil_density[l] = CPV(VICGPU_CP_LAYER(CPL_SOIL_DENSITY, l));
      s->bulk_dens_min[l] = CPV(VICGPU_CP_LAYER(CPL_BULK_DENS_MIN, l));
      s->soil_dens_min[l] = CPV(VICGPU_CP_LAYER(CPL_SOIL_DENS_MIN, l));
    }
    for (n = 0; n < Nn; n++) {
      s->Zsum_node[n] = CPV(VICGPU_CP_NODE(CPN_ZSUM, n, Nn)); s->dz_node[n] = CPV(VICGPU_CP_NODE(CPN_DZ, n, Nn));
      s->alpha[n] = CPV(VICGPU_CP_NODE(CPN_ALPHA, n, Nn)); s->beta[n] = CPV(VICGPU_CP_NODE(CPN_BETA, n, Nn));
      s->gamma[n] = CPV(VICGPU_CP_NODE(CPN_GAMMA, n, Nn)); s->max_moist_node[n] = CPV(VICGPU_CP_NODE(CPN_MAX_MOIST, n, Nn));
      s->expt_node[n] = CPV(VICGPU_CP_NODE(CPN_EXPT, n, Nn)); s->bubble_node[n] = CPV(VICGPU_CP_NODE(CPN_BUBBLE, n, Nn));
    }
    for (b = 0; b < Nb; b++) {
      s->AreaFract[b] = CPV(VICGPU_CP_BAND(CPB_AREAFRACT, b, Nn, Nb)); s->Tfactor[b] = CPV(VICGPU_CP_BAND(CPB_TFACTOR, b, Nn, Nb));
      s->AboveTreeLine[b] = CPV(VICGPU_CP_BAND(CPB_ABOVETREELINE, b, Nn, Nb)) != 0;
      s->Pfactor[b] = CPV(VICGPU_CP_BAND(CPB_PFACTOR, b, Nn, Nb)); s->BandElev[b] = CPV(VICGPU_CP_BAND(CPB_BANDELEV, b, Nn, Nb));
    }
    for (l = 0; l < VIC_NLAYER + 2; l++)
      for (i = 0; i < VIC_MAX_ZWTVMOIST; i++) {
        s->zwt_zwt[l][i] = CPV(VICGPU_CP_ZWT_ZWT(l, i, Nn, Nb));
        s->zwt_moist[l][i] = CPV(VICGPU_CP_ZWT_MOIST(l, i, Nn, Nb));
      }
  }
  for (g = 0; g < nhru; g++) {
    orc_hru *u = &h->hru[g];
    hru_defaults(u);
    u->cell = hpi[(size_t)HPI_CELL * nhru + g]; u->band = hpi[(size_t)HPI_BAND * nhru + g];
    u->veg_index = hpi[(size_t)HPI_VEG_INDEX * nhru + g]; u->veg_class = hpi[(size_t)HPI_VEG_CLASS * nhru + g];
    u->is_glacier = hpi[(size_t)HPI_IS_GLACIER * nhru + g]; u->is_artificial_bare = hpi[(size_t)HPI_IS_ARTIFICIAL_BARE * nhru + g];
    u->Cv = hpd[(size_t)HPD_CV * nhru + g];
    for (l = 0; l < 3; l++) u->root[l] = (double)(float)hpd[(size_t)(HPD_ROOT0 + l) * nhru + g];
    u->sigma_slope = (float)hpd[(size_t)HPD_SIGMA_SLOPE * nhru + g]; u->lag_one = (float)hpd[(size_t)HPD_LAG_ONE * nhru + g];
    u->fetch = (float)hpd[(size_t)HPD_FETCH * nhru + g];
  }
  return 0;
}

#define SDP(row) sd[(size_t)(row) * nh + g]
#define SIP(row) si[(size_t)(row) * nh + g]

int vicorc_get_fluxes(void *hv, double *fx);
int vicorc_get_state(void *hv, double *sd, int *si) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const int Nn = h->model.opt.Nnode; const size_t nh = h->nhru;
  int g, l, n;
  for (g = 0; g < h->nhru; g++) {
    const orc_hru *u = &h->hru[g]; const orc_energy *e = &u->energy; const orc_snow *s = &u->snow;
    for (l = 0; l < 3; l++) { SDP(SD_MOIST0 + l) = u->layer[l].moist; SDP(SD_ICE0 + l) = u->layer[l].ice; SDP(SD_LAYER_T0 + l) = u->layer[l].T; }
    SDP(SD_SNOW_FLUX) = e->snow_flux; SDP(SD_GRND_FLUX) = e->grnd_flux; SDP(SD_DELTAH) = e->deltaH; SDP(SD_FUSION) = e->fusion;
    SDP(SD_LONGUNDEROUT) = e->LongUnderOut; SDP(SD_TFOLIAGE) = e->Tfoliage;
    SDP(SD_SNOW_ALBEDO) = s->albedo; SDP(SD_SNOW_COLDCONTENT) = s->coldcontent; SDP(SD_SNOW_COVERAGE) = s->coverage;
    SDP(SD_SNOW_DENSITY) = s->density; SDP(SD_SNOW_DEPTH) = s->depth; SDP(SD_SNOW_PACK_TEMP) = s->pack_temp;
    SDP(SD_SNOW_PACK_WATER) = s->pack_water; SDP(SD_SNOW_CANOPY) = s->snow_canopy; SDP(SD_SNOW_SURF_TEMP) = s->surf_temp;
    SDP(SD_SNOW_SURF_WATER) = s->surf_water; SDP(SD_SNOW_SWQ) = s->swq; SDP(SD_SNOW_TMP_INT_STORAGE) = s->tmp_int_storage;
    SDP(SD_SNOW_STORE_SWQ) = s->store_swq; SDP(SD_SNOW_STORE_COVERAGE) = s->store_coverage; SDP(SD_SNOW_SWQ_SLOPE) = s->swq_slope;
    SDP(SD_SNOW_MAX_SWQ) = s->max_swq; SDP(SD_WDEW) = u->veg.Wdew;
    SDP(SD_GLAC_SURF_TEMP) = u->glac.surf_temp; SDP(SD_GLAC_WATER_STORAGE) = u->glac.water_storage;
    SDP(SD_GLAC_CUM_MASS_BALANCE) = u->glac.cum_mass_balance;
    SDP(SD_TCANOPY) = e->Tcanopy; SDP(SD_TSURF) = e->Tsurf; SDP(SD_ALBEDO_OVER) = e->AlbedoOver; SDP(SD_ALBEDO_UNDER) = e->AlbedoUnder;
    SDP(SD_CANOPY_ADVECTION) = e->canopy_advection; SDP(SD_CANOPY_LATENT) = e->canopy_latent;
    SDP(SD_CANOPY_LATENT_SUB) = e->canopy_latent_sub; SDP(SD_CANOPY_SENSIBLE) = e->canopy_sensible;
    SDP(SD_CANOPY_REFREEZE) = e->canopy_refreeze; SDP(SD_ADVECTED_SENSIBLE) = e->advected_sensible;
    SDP(SD_ADVECTION) = e->advection; SDP(SD_DELTACC) = e->deltaCC; SDP(SD_REFREEZE_ENERGY) = e->refreeze_energy;
    SDP(SD_MELT_ENERGY) = e->melt_energy; SDP(SD_ERROR) = e->error; SDP(SD_LATENT) = e->latent;
    SDP(SD_LATENT_SUB) = e->latent_sub; SDP(SD_SENSIBLE) = e->sensible; SDP(SD_LONGOVERIN) = e->LongOverIn;
    SDP(SD_NETLONGOVER) = e->NetLongOver; SDP(SD_NETSHORTOVER) = e->NetShortOver; SDP(SD_SHORTOVERIN) = e->ShortOverIn;
    SDP(SD_NETLONGUNDER) = e->NetLongUnder;
    for (n = 0; n < Nn; n++) {
      SDP(VICGPU_SD_NODE(SDN_T, n, Nn)) = e->T[n]; SDP(VICGPU_SD_NODE(SDN_MOIST, n, Nn)) = e->moist[n];
      SDP(VICGPU_SD_NODE(SDN_ICE, n, Nn)) = e->ice[n]; SDP(VICGPU_SD_NODE(SDN_KAPPA, n, Nn)) = e->kappa_node[n];
      SDP(VICGPU_SD_NODE(SDN_CS, n, Nn)) = e->Cs_node[n];
      SIP(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)) = e->T_fbflag[n]; SIP(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn)) = e->T_fbcount[n];
    }
    SIP(SI_SNOW_LAST_SNOW) = s->last_snow; SIP(SI_SNOW_MELTING) = s->MELTING; SIP(SI_SNOW_SNOW) = s->snow;
    SIP(SI_SNOW_STORE_SNOW) = s->store_snow; SIP(SI_SNOW_SURF_TEMP_FBCOUNT) = s->surf_temp_fbcount;
    SIP(SI_SNOW_SURF_TEMP_FBFLAG) = s->surf_temp_fbflag; SIP(SI_TSURF_FBCOUNT) = e->Tsurf_fbcount;
    SIP(SI_TSURF_FBFLAG) = e->Tsurf_fbflag; SIP(SI_TFOLIAGE_FBCOUNT) = e->Tfoliage_fbcount;
    SIP(SI_TFOLIAGE_FBFLAG) = e->Tfoliage_fbflag; SIP(SI_TCANOPY_FBCOUNT) = e->Tcanopy_fbcount;
    SIP(SI_TCANOPY_FBFLAG) = e->Tcanopy_fbflag; SIP(SI_GLAC_SURF_TEMP_FBCOUNT) = u->glac.surf_temp_fbcount;
    SIP(SI_GLAC_SURF_TEMP_FBFLAG) = u->glac.surf_temp_fbflag; SIP(SI_FROZEN) = e->frozen; SIP(SI_NFROST) = e->Nfrost;
    SIP(SI_NTHAW) = e->Nthaw;
  }
  return 0;
}

int vicorc_set_state(void *hv, const double *sd, const int *si) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const int Nn = h->model.opt.Nnode; const size_t nh = h->nhru;
  int g, l, n;
  for (g = 0; g < h->nhru; g++) {
    orc_hru *u = &h->hru[g]; orc_energy *e = &u->energy; orc_snow *s = &u->snow;
    for (l = 0; l < 3; l++) { u->layer[l].moist = SDP(SD_MOIST0 + l); u->layer[l].ice = SDP(SD_ICE0 + l); u->layer[l].T = SDP(SD_LAYER_T0 + l); }
    e->snow_flux = SDP(SD_SNOW_FLUX); e->grnd_flux = SDP(SD_GRND_FLUX); e->deltaH = SDP(SD_DELTAH); e->fusion = SDP(SD_FUSION);
    e->LongUnderOut = SDP(SD_LONGUNDEROUT); e->Tfoliage = SDP(SD_TFOLIAGE);
    s->albedo = SDP(SD_SNOW_ALBEDO); s->coldcontent = SDP(SD_SNOW_COLDCONTENT); s->coverage = SDP(SD_SNOW_COVERAGE);
    s->density = SDP(SD_SNOW_DENSITY); s->depth = SDP(SD_SNOW_DEPTH); s->pack_temp = SDP(SD_SNOW_PACK_TEMP);
    s->pack_water = SDP(SD_SNOW_PACK_WATER); s->snow_canopy = SDP(SD_SNOW_CANOPY); s->surf_temp = SDP(SD_SNOW_SURF_TEMP);
    s->surf_water = SDP(SD_SNOW_SURF_WATER); s->swq = SDP(SD_SNOW_SWQ); s->tmp_int_storage = SDP(SD_SNOW_TMP_INT_STORAGE);
    s->store_swq = SDP(SD_SNOW_STORE_SWQ); s->store_coverage = SDP(SD_SNOW_STORE_COVERAGE); s->swq_slope = SDP(SD_SNOW_SWQ_SLOPE);
    s->max_swq = SDP(SD_SNOW_MAX_SWQ); u->veg.Wdew = SDP(SD_WDEW);
    u->glac.surf_temp = SDP(SD_GLAC_SURF_TEMP); u->glac.water_storage = SDP(SD_GLAC_WATER_STORAGE);
    u->glac.cum_mass_balance = SDP(SD_GLAC_CUM_MASS_BALANCE);
    e->Tcanopy = SDP(SD_TCANOPY); e->Tsurf = SDP(SD_TSURF); e->AlbedoOver = SDP(SD_ALBEDO_OVER); e->AlbedoUnder = SDP(SD_ALBEDO_UNDER);
    e->canopy_advection = SDP(SD_CANOPY_ADVECTION); e->canopy_latent = SDP(SD_CANOPY_LATENT);
    e->canopy_latent_sub = SDP(SD_CANOPY_LATENT_SUB); e->canopy_sensible = SDP(SD_CANOPY_SENSIBLE);
    e->canopy_refreeze = SDP(SD_CANOPY_REFREEZE); e->advected_sensible = SDP(SD_ADVECTED_SENSIBLE);
    e->advection = SDP(SD_ADVECTION); e->deltaCC = SDP(SD_DELTACC); e->refreeze_energy = SDP(SD_REFREEZE_ENERGY);
    e->melt_energy = SDP(SD_MELT_ENERGY); e->error = SDP(SD_ERROR); e->latent = SDP(SD_LATENT);
    e->latent_sub = SDP(SD_LATENT_SUB); e->sensible = SDP(SD_SENSIBLE); e->LongOverIn = SDP(SD_LONGOVERIN);
    e->NetLongOver = SDP(SD_NETLONGOVER); e->NetShortOver = SDP(SD_NETSHORTOVER); e->ShortOverIn = SDP(SD_SHORTOVERIN);
    e->NetLongUnder = SDP(SD_NETLONGUNDER);
    for (n = 0; n < Nn; n++) {
      e->T[n] = SDP(VICGPU_SD_NODE(SDN_T, n, Nn)); e->moist[n] = SDP(VICGPU_SD_NODE(SDN_MOIST, n, Nn));
      e->ice[n] = SDP(VICGPU_SD_NODE(SDN_ICE, n, Nn)); e->kappa_node[n] = SDP(VICGPU_SD_NODE(SDN_KAPPA, n, Nn));
      e->Cs_node[n] = SDP(VICGPU_SD_NODE(SDN_CS, n, Nn));
      e->T_fbflag[n] = SIP(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)); e->T_fbcount[n] = SIP(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn));
    }
    s->last_snow = SIP(SI_SNOW_LAST_SNOW); s->MELTING = SIP(SI_SNOW_MELTING); s->snow = SIP(SI_SNOW_SNOW);
    s->store_snow = SIP(SI_SNOW_STORE_SNOW); s->surf_temp_fbcount = SIP(SI_SNOW_SURF_TEMP_FBCOUNT);
    s->surf_temp_fbflag = SIP(SI_SNOW_SURF_TEMP_FBFLAG); e->Tsurf_fbcount = SIP(SI_TSURF_FBCOUNT);
    e->Tsurf_fbflag = SIP(SI_TSURF_FBFLAG); e->Tfoliage_fbcount = SIP(SI_TFOLIAGE_FBCOUNT);
    e->Tfoliage_fbflag = SIP(SI_TFOLIAGE_FBFLAG); e->Tcanopy_fbcount = SIP(SI_TCANOPY_FBCOUNT);
    e->Tcanopy_fbflag = SIP(SI_TCANOPY_FBFLAG); u->glac.surf_temp_fbcount = SIP(SI_GLAC_SURF_TEMP_FBCOUNT);
    u->glac.surf_temp_fbflag = SIP(SI_GLAC_SURF_TEMP_FBFLAG); e->frozen = SIP(SI_FROZEN); e->Nfrost = SIP(SI_NFROST);
    e->Nthaw = SIP(SI_NTHAW);
  }
  return 0;
}

/* The state as the reference's state file holds it: one record per HRU in hruList order, fields in the order
 * processCellForStateFile streams them (write_model_state.c:166-285; include/vicgpu.h SR_*).  write != 0: HRUs -> records;
 * otherwise records -> HRUs (the read side of the same function); returns 1 + index of the first record whose band or
 * vegetation class does not match (write_model_state.c:179-188), with nothing read. */
int vicorc_state_records(void *hv, double *rec, int write) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const int Nn = h->model.opt.Nnode, L = VICGPU_SR_LEN(Nn);
  int k, l, n;
  if (!write)
    for (k = 0; k < h->nhru; k++) {
      const orc_hru *u = &h->hru[h->cell_list[k]];
      if ((int)rec[(size_t)k * L + SR_BAND_INDEX] != u->band || (int)rec[(size_t)k * L + SR_VEG_CLASS] != u->veg_class) return k + 1;
    }
#define X(slot, field) do { if (write) r[slot] = (field); else (field) = r[slot]; } while (0)
  for (k = 0; k < h->nhru; k++) {
    orc_hru *u = &h->hru[h->cell_list[k]];
    orc_energy *e = &u->energy; orc_snow *s = &u->snow; orc_glac *g = &u->glac;
    double *r = rec + (size_t)k * L;
    if (write) { r[SR_BAND_INDEX] = u->band; r[SR_VEG_CLASS] = u->veg_class; }
    for (l = 0; l < 3; l++) { X(SR_MOIST0 + l, u->layer[l].moist); X(SR_ICE0 + l, u->layer[l].ice); }
    X(SR_WDEW, u->veg.Wdew);
    X(SR_SNOW_CANOPY, s->snow_canopy); X(SR_SNOW_DENSITY, s->density); X(SR_SNOW_DEPTH, s->depth); X(SR_SNOW_PACK_WATER, s->pack_water);
    X(SR_SNOW_SURF_WATER, s->surf_water); X(SR_SNOW_SWQ, s->swq);
    X(SR_GLAC_WATER_STORAGE, g->water_storage); X(SR_GLAC_CUM_MASS_BALANCE, g->cum_mass_balance);
    for (n = 0; n < Nn; n++) { X(SR_ENERGY_T + n, e->T[n]); X(VICGPU_SR_T(SRT_T_FBCOUNT, Nn) + n, e->T_fbcount[n]); }
    X(VICGPU_SR_T(SRT_TFOLIAGE, Nn), e->Tfoliage); X(VICGPU_SR_T(SRT_GLAC_SURF_TEMP, Nn), g->surf_temp);
    X(VICGPU_SR_T(SRT_SNOW_COLD_CONTENT, Nn), s->coldcontent); X(VICGPU_SR_T(SRT_SNOW_PACK_TEMP, Nn), s->pack_temp);
    X(VICGPU_SR_T(SRT_SNOW_SURF_TEMP, Nn), s->surf_temp); X(VICGPU_SR_T(SRT_SNOW_ALBEDO, Nn), s->albedo);
    X(VICGPU_SR_T(SRT_SNOW_LAST_SNOW, Nn), s->last_snow); X(VICGPU_SR_T(SRT_SNOW_MELTING, Nn), s->MELTING);
    X(VICGPU_SR_T(SRT_TCANOPY_FBCOUNT, Nn), e->Tcanopy_fbcount);
    X(VICGPU_SR_U(SRU_TFOLIAGE_FBCOUNT, Nn), e->Tfoliage_fbcount); X(VICGPU_SR_U(SRU_TSURF_FBCOUNT, Nn), e->Tsurf_fbcount);
    X(VICGPU_SR_U(SRU_GLAC_SURF_TEMP_FBCOUNT, Nn), g->surf_temp_fbcount); X(VICGPU_SR_U(SRU_SNOW_SURF_TEMP_FBCOUNT, Nn), s->surf_temp_fbcount);
    X(VICGPU_SR_U(SRU_GLAC_SURF_TEMP_FBFLAG, Nn), g->surf_temp_fbflag);
    X(VICGPU_SR_U(SRU_GLAC_VAPOR_FLUX, Nn), g->vapor_flux);
    if (write) r[VICGPU_SR_U(SRU_SNOW_CANOPY_ALBEDO, Nn)] = 0.0;   /* initialize_snow.c:62, never assigned again */
    X(VICGPU_SR_U(SRU_SNOW_SURFACE_FLUX, Nn), s->surface_flux);
    X(VICGPU_SR_U(SRU_SNOW_SURF_TEMP_FBFLAG, Nn), s->surf_temp_fbflag);
    X(VICGPU_SR_U(SRU_SNOW_TMP_INT_STORAGE, Nn), s->tmp_int_storage);
    X(VICGPU_SR_U(SRU_SNOW_VAPOR_FLUX, Nn), s->vapor_flux);
  }
#undef X
  return 0;
}

/* initialize_atmos.c: atmos[rec] from one record of hourly forcing, for the case every variable is supplied sub-daily
 * (raw [nsteps][VIC_NRAW][dt][ncell], kPa for the two pressures): kPa -> Pa (:290-295), the snow_step-hour aggregation
 * with mean / sum in sub-index NR (:886-893 and its siblings), the MIN_WIND_SPEED floor per hour (:518-536), density from
 * pressure (:980-1000; sub-index NR from pressure[NR] and air_temp[NR], not the mean of the sub-steps' densities), vpd =
 * svp(T) - vp clipped at 0 with vp reset (:1175-1193), snowflag (:1275-1303); MIN_WIND_SPEED is a float option
 * (vicNl_def.h:713).  Pinned bit for bit against the reference's own initialize_atmos run on in-memory records
 * (oracle/ref_build/vicref_shim.cpp vicref_derive_forcing, tests/test_forcing_stream.py). */
int vicorc_derive_forcing(void *hv, int nsteps, const double *raw, double min_wind_in, int plapse, double *forcing, unsigned char *snowflag) {
  const double min_wind = (double)(float)min_wind_in;
  vicorc_handle *h = (vicorc_handle *)hv;
  const vicgpu_options *o = &h->model.opt;
  const int NF = h->model.NF, NR = h->model.NR, ns = NR + 1, dt = o->dt, ss = o->snow_step;
  const size_t nc = h->ncell;
  int s, c, j, hh, b;
  for (s = 0; s < nsteps; s++)
    for (c = 0; c < h->ncell; c++) {
      const orc_soil *sc = &h->soil[c];
      const double *r = raw + (size_t)s * VIC_NRAW * dt * nc + c;
      double *f = forcing + (size_t)s * VIC_NFORCE * ns * nc + c;
      unsigned char *sf = snowflag + (size_t)s * ns * nc + c;
      double min_Tfactor = sc->Tfactor[0], thr;
      double sum[VIC_NFORCE] = {0};
      int any = 0, v;
      for (b = 1; b < o->Nband; b++) if (sc->Tfactor[b] < min_Tfactor) min_Tfactor = sc->Tfactor[b];
      thr = (o->TEMP_TH_TYPE == VIC_TEMP_TH_KIENZLE) ? (sc->MAX_SNOW_TEMP + sc->MIN_RAIN_TEMP / 2) : sc->MAX_SNOW_TEMP;
#define RAW(v, h) r[((size_t)(v) * dt + (h)) * nc]
#define FO(v, j) f[((size_t)(v) * ns + (j)) * nc]
      for (j = 0; j < NF; j++) {
        double T = 0, prec = 0, pr = 0, vp = 0, sw = 0, lw = 0, wind = 0, dens, vpd;
        int snow;
        for (hh = j * ss; hh < (j + 1) * ss; hh++) {
          double w = RAW(VIC_RAW_WIND, hh);
          T += RAW(VIC_RAW_AIR_TEMP, hh); prec += RAW(VIC_RAW_PREC, hh);
          pr += RAW(VIC_RAW_PRESSURE_KPA, hh) * 1000.0; vp += RAW(VIC_RAW_VP_KPA, hh) * 1000.0;
          sw += RAW(VIC_RAW_SHORTWAVE, hh); lw += RAW(VIC_RAW_LONGWAVE, hh);
          wind += (w < min_wind) ? min_wind : w;
        }
        T /= ss; pr /= ss; vp /= ss; sw /= ss; lw /= ss; wind /= ss;
        dens = plapse ? pr / (287.0 * (ORC_KELVIN + T)) : 0.003486 * pr / (275.0 + T);
        vpd = orc_svp(T) - vp;
        if (vpd < 0) { vpd = 0; vp = orc_svp(T); }
        FO(VIC_F_AIR_TEMP, j) = T; FO(VIC_F_PREC, j) = prec; FO(VIC_F_PRESSURE, j) = pr; FO(VIC_F_VP, j) = vp; FO(VIC_F_VPD, j) = vpd;
        FO(VIC_F_DENSITY, j) = dens; FO(VIC_F_SHORTWAVE, j) = sw; FO(VIC_F_LONGWAVE, j) = lw; FO(VIC_F_WIND, j) = wind;
        snow = ((T + min_Tfactor) < thr) && (prec > 0);
        sf[(size_t)j * nc] = (unsigned char)snow;
        any |= snow;
        sum[VIC_F_AIR_TEMP] += T; sum[VIC_F_PREC] += prec; sum[VIC_F_PRESSURE] += pr; sum[VIC_F_VP] += vp; sum[VIC_F_VPD] += vpd;
        sum[VIC_F_DENSITY] += dens; sum[VIC_F_SHORTWAVE] += sw; sum[VIC_F_LONGWAVE] += lw; sum[VIC_F_WIND] += wind;
      }
      if (NF > 1) {
        for (v = 0; v < VIC_NFORCE; v++) FO(v, NR) = (v == VIC_F_PREC) ? sum[v] : sum[v] / (double)(float)NF;
        FO(VIC_F_DENSITY, NR) = plapse ? FO(VIC_F_PRESSURE, NR) / (287.0 * (ORC_KELVIN + FO(VIC_F_AIR_TEMP, NR)))
                                       : 0.003486 * FO(VIC_F_PRESSURE, NR) / (275.0 + FO(VIC_F_AIR_TEMP, NR));
        sf[(size_t)NR * nc] = (unsigned char)any;
      }
#undef RAW
#undef FO
    }
  return 0;
}

int vicorc_implicit_stats(void *hv, long *ok, long *failed) { vicorc_handle *h = (vicorc_handle *)hv; *ok = h->model.implicit_ok; *failed = h->model.implicit_failed; return 0; }

static void export_flux(vicorc_handle *h, double *fx) {
  const size_t nh = h->nhru;
  int g, l, p;
  for (g = 0; g < h->nhru; g++) {
    const orc_hru *u = &h->hru[g]; const orc_energy *e = &u->energy; const orc_snow *s = &u->snow;
#define FXP(row) fx[(size_t)(row) * nh + g]
    FXP(FX_RUNOFF) = u->runoff; FXP(FX_BASEFLOW) = u->baseflow; FXP(FX_ASAT) = u->asat; FXP(FX_INFLOW) = u->inflow;
    for (l = 0; l < 3; l++) FXP(FX_EVAP0 + l) = u->layer[l].evap;
    FXP(FX_CANOPYEVAP) = u->veg.canopyevap; FXP(FX_THROUGHFALL) = u->veg.throughfall;
    FXP(FX_SNOW_VAPOR_FLUX) = s->vapor_flux; FXP(FX_SNOW_CANOPY_VAPOR_FLUX) = s->canopy_vapor_flux;
    FXP(FX_SNOW_BLOWING_FLUX) = s->blowing_flux; FXP(FX_SNOW_SURFACE_FLUX) = s->surface_flux; FXP(FX_SNOW_MELT) = s->melt;
    FXP(FX_SNOW_MASS_ERROR) = s->mass_error; FXP(FX_SNOW_QNET) = s->Qnet;
    for (l = 0; l < 3; l++) { FXP(FX_FDEPTH0 + l) = e->fdepth[l]; FXP(FX_TDEPTH0 + l) = e->tdepth[l]; FXP(FX_ZWTL0 + l) = u->layer[l].zwt; }
    for (p = 0; p < 6; p++) FXP(FX_POT_EVAP0 + p) = u->pot_evap[p];
    FXP(FX_AERO_RESIST_SURFACE) = u->aero_resist_surface; FXP(FX_AERO_RESIST_OVERSTORY) = u->aero_resist_overstory;
    FXP(FX_ROOTMOIST) = u->rootmoist; FXP(FX_WETNESS) = u->wetness; FXP(FX_ZWT) = u->zwt; FXP(FX_ZWT2) = u->zwt2; FXP(FX_ZWT3) = u->zwt3;
    FXP(FX_ATMOS_LATENT) = e->AtmosLatent; FXP(FX_ATMOS_LATENT_SUB) = e->AtmosLatentSub; FXP(FX_ATMOS_SENSIBLE) = e->AtmosSensible;
    FXP(FX_LONG_UNDER_IN) = e->LongUnderIn; FXP(FX_NET_LONG_ATMOS) = e->NetLongAtmos; FXP(FX_NET_LONG_UNDER) = e->NetLongUnder;
    FXP(FX_NET_SHORT_ATMOS) = e->NetShortAtmos; FXP(FX_NET_SHORT_GRND) = e->NetShortGrnd; FXP(FX_NET_SHORT_UNDER) = e->NetShortUnder;
    FXP(FX_SHORT_UNDER_IN) = e->ShortUnderIn;
    FXP(FX_OUT_PREC) = u->out_prec; FXP(FX_OUT_RAIN) = u->out_rain; FXP(FX_OUT_SNOW) = u->out_snow;
    FXP(FX_GLAC_MASS_BALANCE) = u->glac.mass_balance; FXP(FX_GLAC_ICE_MASS_BALANCE) = u->glac.ice_mass_balance;
    FXP(FX_GLAC_ACCUMULATION) = u->glac.accumulation; FXP(FX_GLAC_MELT) = u->glac.melt; FXP(FX_GLAC_VAPOR_FLUX) = u->glac.vapor_flux;
    FXP(FX_GLAC_INFLOW) = u->glac.inflow; FXP(FX_GLAC_OUTFLOW) = u->glac.outflow; FXP(FX_GLAC_OUTFLOW_COEF) = u->glac.outflow_coef;
    FXP(FX_GLAC_QNET) = u->glac.Qnet; FXP(FX_GLAC_COLD_CONTENT) = u->glac.cold_content;
    FXP(FX_GLACIER_FLUX) = e->glacier_flux; FXP(FX_DELTACC_GLAC) = e->deltaCC_glac; FXP(FX_GLACIER_MELT_ENERGY) = e->glacier_melt_energy;
#undef FXP
  }
}

/* the per-HRU values put_data reads that a step may leave untouched (initialize_model_state results, glacier HRUs' soil
 * diagnostics): taken from a flux table [FX_NROW][nhru] */
int vicorc_set_fluxes(void *hv, const double *fx) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const size_t nh = h->nhru;
  int g, l, p;
  for (g = 0; g < h->nhru; g++) {
    orc_hru *u = &h->hru[g]; orc_energy *e = &u->energy; orc_snow *s = &u->snow;
#define FXP(row) fx[(size_t)(row) * nh + g]
    u->runoff = FXP(FX_RUNOFF); u->baseflow = FXP(FX_BASEFLOW); u->asat = FXP(FX_ASAT); u->inflow = FXP(FX_INFLOW);
    for (l = 0; l < 3; l++) u->layer[l].evap = FXP(FX_EVAP0 + l);
    u->veg.canopyevap = FXP(FX_CANOPYEVAP); u->veg.throughfall = FXP(FX_THROUGHFALL);
    s->vapor_flux = FXP(FX_SNOW_VAPOR_FLUX); s->canopy_vapor_flux = FXP(FX_SNOW_CANOPY_VAPOR_FLUX);
    s->blowing_flux = FXP(FX_SNOW_BLOWING_FLUX); s->surface_flux = FXP(FX_SNOW_SURFACE_FLUX); s->melt = FXP(FX_SNOW_MELT);
    for (l = 0; l < 3; l++) { e->fdepth[l] = FXP(FX_FDEPTH0 + l); e->tdepth[l] = FXP(FX_TDEPTH0 + l); u->layer[l].zwt = FXP(FX_ZWTL0 + l); }
    for (p = 0; p < 6; p++) u->pot_evap[p] = FXP(FX_POT_EVAP0 + p);
    u->aero_resist_surface = FXP(FX_AERO_RESIST_SURFACE); u->aero_resist_overstory = FXP(FX_AERO_RESIST_OVERSTORY);
    u->rootmoist = FXP(FX_ROOTMOIST); u->wetness = FXP(FX_WETNESS); u->zwt = FXP(FX_ZWT); u->zwt2 = FXP(FX_ZWT2); u->zwt3 = FXP(FX_ZWT3);
    e->AtmosLatent = FXP(FX_ATMOS_LATENT); e->AtmosLatentSub = FXP(FX_ATMOS_LATENT_SUB); e->AtmosSensible = FXP(FX_ATMOS_SENSIBLE);
    e->LongUnderIn = FXP(FX_LONG_UNDER_IN); e->NetLongAtmos = FXP(FX_NET_LONG_ATMOS); e->NetShortAtmos = FXP(FX_NET_SHORT_ATMOS);
    u->glac.mass_balance = FXP(FX_GLAC_MASS_BALANCE); u->glac.ice_mass_balance = FXP(FX_GLAC_ICE_MASS_BALANCE);
    u->glac.accumulation = FXP(FX_GLAC_ACCUMULATION); u->glac.melt = FXP(FX_GLAC_MELT); u->glac.vapor_flux = FXP(FX_GLAC_VAPOR_FLUX);
    u->glac.inflow = FXP(FX_GLAC_INFLOW); u->glac.outflow = FXP(FX_GLAC_OUTFLOW); u->glac.outflow_coef = FXP(FX_GLAC_OUTFLOW_COEF);
    e->glacier_flux = FXP(FX_GLACIER_FLUX); e->deltaCC_glac = FXP(FX_DELTACC_GLAC); e->glacier_melt_energy = FXP(FX_GLACIER_MELT_ENERGY);
#undef FXP
  }
  return 0;
}

/* one model step for all cells; same contract as vicref_step (oracle/ref_build/vicref_shim.cpp) */
int vicorc_step(void *hv, const double *forcing, const unsigned char *snowflag, const int *dmyv, double *flux,
                double *cell_out, int *cell_err, int nthreads) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const int ns = h->model.NR + 1, nc = h->ncell;
  orc_dmy d;
  int c, nerr = 0;
  d.month = dmyv[VIC_DMY_MONTH]; d.day_in_year = dmyv[VIC_DMY_DAY_IN_YEAR]; d.hour = dmyv[VIC_DMY_HOUR];
  d.day = dmyv[VIC_DMY_DAY]; d.year = dmyv[VIC_DMY_YEAR];
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+:nerr)
  for (c = 0; c < nc; c++) {
    orc_atmos a;
    orc_hru *list[VIC_MAX_BANDS * 8];
    int s, k, n = h->cell_off[c + 1] - h->cell_off[c], err;
    for (s = 0; s < ns; s++) {
#define FV(v) forcing[((size_t)(v) * ns + s) * nc + c]
      a.air_temp[s] = FV(VIC_F_AIR_TEMP); a.prec[s] = FV(VIC_F_PREC); a.pressure[s] = FV(VIC_F_PRESSURE); a.vp[s] = FV(VIC_F_VP);
      a.vpd[s] = FV(VIC_F_VPD); a.density[s] = FV(VIC_F_DENSITY); a.shortwave[s] = FV(VIC_F_SHORTWAVE);
      a.longwave[s] = FV(VIC_F_LONGWAVE); a.wind[s] = FV(VIC_F_WIND);
#undef FV
      a.snowflag[s] = snowflag ? snowflag[(size_t)s * nc + c] : 0;
    }
    if (n > VIC_MAX_BANDS * 8) { if (cell_err) cell_err[c] = VICGPU_CELLERR_SOLVER; nerr++; continue; }
    for (k = 0; k < n; k++) list[k] = &h->hru[h->cell_list[h->cell_off[c] + k]];
    err = orc_full_energy(&h->model, &h->soil[c], &a, &d, list, n);
    if (cell_err) cell_err[c] = err;
    if (err) nerr++;
    for (k = 0; k < n; k++)   /* accumulateGlacierMassBalance.c:13-67: the += (window logic is driver state) */
      if (list[k]->is_glacier && !isnan(list[k]->glac.cum_mass_balance) && !isnan(list[k]->glac.mass_balance))
        list[k]->glac.cum_mass_balance += list[k]->glac.mass_balance;
    if (cell_out) {
      cell_out[(size_t)CO_OUT_PREC * nc + c] = a.out_prec;
      cell_out[(size_t)CO_OUT_RAIN * nc + c] = a.out_rain;
      cell_out[(size_t)CO_OUT_SNOW * nc + c] = a.out_snow;
    }
  }
  if (flux) export_flux(h, flux);
  return nerr;
}

int vicorc_get_fluxes(void *hv, double *fx) { export_flux((vicorc_handle *)hv, fx); return 0; }

double vicorc_run(void *hv, int nsteps, const double *forcing, const unsigned char *snowflag, const int *dmyv, int nthreads) {
  vicorc_handle *h = (vicorc_handle *)hv;
  const size_t fstride = (size_t)VIC_NFORCE * (h->model.NR + 1) * h->ncell;
  const size_t sstride = (size_t)(h->model.NR + 1) * h->ncell;
  struct timespec t0, t1;
  int s;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (s = 0; s < nsteps; s++)
    vicorc_step(hv, forcing + s * fstride, snowflag ? snowflag + s * sstride : NULL, dmyv + (size_t)s * VIC_NDMY, NULL, NULL,
                NULL, nthreads);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* accumulateGlacierMassBalance.c:53-66 at the end of an accumulation interval: fit per cell, then reset */
void orc_glacier_mass_balance_fit(const orc_soil *sc, orc_hru **hrus, int nhru, double *eq);

int vicorc_glacier_fit(void *hv, double *eq, int reset) {
  vicorc_handle *h = (vicorc_handle *)hv;
  int c, k;
  if (!h || !eq || !h->hru) return -1;
  for (c = 0; c < h->ncell; c++) {
    orc_hru *list[VIC_MAX_BANDS * 8 + 8];
    double e[4];
    int n = 0;
    for (k = h->cell_off[c]; k < h->cell_off[c + 1] && n < (int)(sizeof(list) / sizeof(list[0])); k++) list[n++] = &h->hru[h->cell_list[k]];
    orc_glacier_mass_balance_fit(&h->soil[c], list, n, e);
    for (k = 0; k < 4; k++) eq[(size_t)k * h->ncell + c] = e[k];
    if (reset)
      for (k = 0; k < n; k++)
        if (list[k]->is_glacier) list[k]->glac.cum_mass_balance = 0;
  }
  return 0;
}

/* the pure functions of the path one by one (include/vicgpu.h VICGPU_PURE_*) */
double orc_snow_albedo_x(const orc_model *m, double new_snow, double swq, double depth, double albedo, double cold_content,
                         double dt, int last_snow, int MELTING, const orc_soil *sc);
double orc_new_snow_density_x(const orc_model *m, double air_temp);
double orc_estimate_T1_x(double Ts, double T1_old, double T2, double D1, double D2, double kappa1, double kappa2, double Cs1, double Cs2,
                         double dp, double delta_t);

int vicorc_pure(void *hv, int fn, int n, const double *in, double *out) {
  vicorc_handle *h = (vicorc_handle *)hv;
  int i;
  if (!h || n < 0 || !in || !out) return -1;
  for (i = 0; i < n; i++) {
    const double *a = in + (size_t)i * VICGPU_PURE_NIN;
    double r;
    switch (fn) {
      case VICGPU_PURE_SVP: r = orc_svp(a[0]); break;
      case VICGPU_PURE_SVP_SLOPE: r = orc_svp_slope(a[0]); break;
      case VICGPU_PURE_CALC_RAINONLY: r = orc_calc_rainonly(&h->model, a[0], a[1], a[2], a[3]); break;
      case VICGPU_PURE_SNOW_ALBEDO:
        if (!h->soil) return -1;
        r = orc_snow_albedo_x(&h->model, a[0], a[1], a[2], a[3], a[4], a[5], (int)a[6], a[7] != 0.0, &h->soil[0]); break;
      case VICGPU_PURE_NEW_SNOW_DENSITY: r = orc_new_snow_density_x(&h->model, a[0]); break;
      case VICGPU_PURE_STABILITY: r = orc_stability_correction(a[0], a[1], a[2], a[3], a[4], a[5]); break;
      case VICGPU_PURE_PENMAN: r = orc_penman(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
      case VICGPU_PURE_CALC_RC: r = orc_calc_rc(a[0], a[1], (float)a[2], a[3], a[4], a[5], a[6], a[7] != 0.0); break;
      case VICGPU_PURE_ESTIMATE_T1: r = orc_estimate_T1_x(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[7], a[8], a[9]); break;
      case VICGPU_PURE_SOIL_CONDUCTIVITY: r = orc_soil_conductivity(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
      case VICGPU_PURE_VOL_HEAT_CAPACITY: r = orc_volumetric_heat_capacity(a[0], a[1], a[2], a[3]); break;
      case VICGPU_PURE_MAX_UNFROZEN_WATER: r = orc_maximum_unfrozen_water(a[0], a[1], a[2], a[3]); break;
      case VICGPU_PURE_LINEAR_INTERP: r = orc_linear_interp(a[0], a[1], a[2], a[3], a[4]); break;
      case VICGPU_PURE_VEG_HEIGHT: r = orc_calc_veg_height(a[0], a[1]); break;
      default: return -1;
    }
    out[i] = r;
  }
  return 0;
}

/* test-only: stopping tolerance of the frozen-node root finds (orc.h) */
int vicorc_set_node_tolerance(void *hv, double macheps, double ttol) {
  vicorc_handle *h = (vicorc_handle *)hv;
  if (!h || !(macheps >= 0) || !(ttol > 0)) return -1;
  h->model.node_macheps = macheps; h->model.node_ttol = ttol;
  return 0;
}

int vicorc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
