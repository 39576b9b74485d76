/*
 * orc_glacier.c — TEST INFRASTRUCTURE (CPU oracle): glacier HRU path
 * (surface_fluxes_glac.c, solve_snow_glac.c, snow_melt_glac.c, solve_glacier.c, glacier_melt.c,
 *  GlacierEnergyBalance.c, latent_heat_from_glacier.c).
 */
#include "orc.h"

void orc_latent_heat_from_snow(double AirDens, double EactAir, double Lv, double Press, double Ra, double TMean, double Vpd,
                               double *LatentHeat, double *LatentHeatSub, double *VaporMassFlux, double *BlowingMassFlux,
                               double *SurfaceMassFlux);

/* ---- the snow-pack residual, same physics as orc_snow.c's (SnowPackEnergyBalance.c:85-197) ---- */
typedef struct {
  double Dt, Ra, Z, z0_snow, AirDens, EactAir, LongSnowIn, Lv, Press, Rain, NetShortUnder, Vpd, Wind, OldTSurf,
         SnowDepth, SnowDensity, SurfaceLiquidWater, SweSurfaceLayer, Tair, TGrnd;
  double *ra_used_surface;
  double *AdvectedEnergy, *AdvectedSensibleHeat, *DeltaColdContent, *GroundFlux, *LatentHeat, *LatentHeatSub,
         *NetLongUnder, *RefreezeEnergy, *SensibleHeat, *vapor_flux, *blowing_flux, *surface_flux;
} gl_spe_ctx;

static double gl_snowpack_energy_balance(double TSurf, void *vctx) {
  gl_spe_ctx *c = (gl_spe_ctx *)vctx;
  double TMean = TSurf, Density = ORC_RHO_W, Tmp, NetRad, RestTerm, VaporMassFlux, BlowingMassFlux, SurfaceMassFlux;
  if (c->Wind > 0.0) *c->ra_used_surface = c->Ra / orc_stability_correction(c->Z, 0.f, TMean, c->Tair, c->Wind, c->z0_snow);
  else *c->ra_used_surface = ORC_HUGE_RESIST;
  Tmp = TMean + ORC_KELVIN;
  *c->NetLongUnder = c->LongSnowIn - ORC_STEFAN_B * Tmp * Tmp * Tmp * Tmp;
  NetRad = c->NetShortUnder + *c->NetLongUnder;
  *c->SensibleHeat = c->AirDens * ORC_CP * (c->Tair - TMean) / *c->ra_used_surface;
  *c->AdvectedSensibleHeat = 0;
  VaporMassFlux = *c->vapor_flux * Density / c->Dt;
  BlowingMassFlux = *c->blowing_flux * Density / c->Dt;
  SurfaceMassFlux = *c->surface_flux * Density / c->Dt;
  orc_latent_heat_from_snow(c->AirDens, c->EactAir, c->Lv, c->Press, *c->ra_used_surface, TMean, c->Vpd, c->LatentHeat,
                            c->LatentHeatSub, &VaporMassFlux, &BlowingMassFlux, &SurfaceMassFlux);
  *c->vapor_flux = VaporMassFlux * c->Dt / Density;
  *c->blowing_flux = BlowingMassFlux * c->Dt / Density;
  *c->surface_flux = SurfaceMassFlux * c->Dt / Density;
  if (TMean == 0) *c->AdvectedEnergy = (ORC_CH_WATER * (c->Tair) * c->Rain) / (c->Dt);
  else *c->AdvectedEnergy = 0.;
  *c->DeltaColdContent = ORC_CH_ICE * c->SweSurfaceLayer * (TSurf - c->OldTSurf) / (c->Dt);
  if (c->SnowDepth > 0.) *c->GroundFlux = ORC_K_SNOW * c->SnowDensity * c->SnowDensity * (c->TGrnd - TMean) / c->SnowDepth / (c->Dt);
  else *c->GroundFlux = 0;
  RestTerm = NetRad + *c->SensibleHeat + *c->LatentHeat + *c->LatentHeatSub + *c->AdvectedEnergy + *c->AdvectedSensibleHeat
             - *c->DeltaColdContent + *c->GroundFlux;
  *c->RefreezeEnergy = (c->SurfaceLiquidWater * ORC_LF * Density) / (c->Dt);
  if (TSurf == 0.0 && RestTerm > -(*c->RefreezeEnergy)) {
    *c->RefreezeEnergy = -RestTerm;
    RestTerm = 0.0;
  } else RestTerm += *c->RefreezeEnergy;
  return RestTerm;
}

/* snow_melt_glac.c:14-420: snow on glacier ice.  melt stays in m; firn -> ice conversion feeds glacier->accumulation. */
static int orc_snow_melt_glac(const orc_model *m, double Le, double NetShortSnow, double Tgrnd, double z0_snow,
                              double aero_resist, double *ra_used_surface, double air_temp, double delta_t, double density,
                              double LongSnowIn, double pressure, double rainfall, double snowfall, double vp, double vpd,
                              double wind, double z2, double *NetLongSnow, double *OldTSurf, double *melt, double *save_Qnet,
                              double *save_advected_sensible, double *save_advection, double *save_deltaCC,
                              double *save_grnd_flux, double *save_latent, double *save_latent_sub,
                              double *save_refreeze_energy, double *save_sensible, orc_snow *snow, orc_glac *glacier) {
  double DeltaPackCC, DeltaPackSwq, Ice, InitialSwq, MaxLiquidWater, PackCC, PackSwq, Qnet, RefreezeEnergy,
         PackRefreezeEnergy, RefrozenWater, SnowFallCC, SnowMelt = 0, SurfaceCC, SurfaceSwq, SnowFall, RainFall;
  double advection, deltaCC, latent_heat, latent_heat_sub, sensible_heat, advected_sensible_heat, grnd_flux = 0, FirnToIce = 0.;
  gl_spe_ctx c;
  SnowFall = snowfall / 1000.;
  RainFall = rainfall / 1000.;
  InitialSwq = snow->swq;
  *OldTSurf = snow->surf_temp;
  Ice = snow->swq - snow->pack_water - snow->surf_water;
  if (Ice > ORC_MAX_SURFACE_SWE) SurfaceSwq = ORC_MAX_SURFACE_SWE; else SurfaceSwq = Ice;
  PackSwq = Ice - SurfaceSwq;
  SurfaceCC = ORC_CH_ICE * SurfaceSwq * snow->surf_temp;
  PackCC = ORC_CH_ICE * PackSwq * snow->pack_temp;
  if (air_temp > 0.0) SnowFallCC = 0.0; else SnowFallCC = ORC_CH_ICE * SnowFall * air_temp;
  if (SnowFall > (ORC_MAX_SURFACE_SWE - SurfaceSwq) && (ORC_MAX_SURFACE_SWE - SurfaceSwq) > ORC_SMALL) {
    DeltaPackSwq = SurfaceSwq + SnowFall - ORC_MAX_SURFACE_SWE;
    if (DeltaPackSwq > SurfaceSwq) DeltaPackCC = SurfaceCC + (SnowFall - ORC_MAX_SURFACE_SWE) / SnowFall * SnowFallCC;
    else DeltaPackCC = DeltaPackSwq / SurfaceSwq * SurfaceCC;
    SurfaceSwq = ORC_MAX_SURFACE_SWE;
    SurfaceCC += SnowFallCC - DeltaPackCC;
    PackSwq += DeltaPackSwq;
    PackCC += DeltaPackCC;
  } else {
    SurfaceSwq += SnowFall;
    SurfaceCC += SnowFallCC;
  }
  if (SurfaceSwq > 0.0) snow->surf_temp = SurfaceCC / (ORC_CH_ICE * SurfaceSwq); else snow->surf_temp = 0.0;
  if (PackSwq > 0.0) {                                                             /* firn to ice, :110-132 */
    if (snow->density > ORC_SNOW_SURF_DENSITY) {
      double zco = (ORC_CUTOFF_DENSITY - ORC_SNOW_SURF_DENSITY) * (snow->depth / 2) / (snow->density - ORC_SNOW_SURF_DENSITY);
      if (zco < snow->depth) {
        double density_zsnow = ORC_SNOW_SURF_DENSITY + 2 * (snow->density - ORC_SNOW_SURF_DENSITY);
        FirnToIce = (density_zsnow + ORC_CUTOFF_DENSITY) / (2 * ORC_RHO_W) * (snow->depth - zco);
        if (FirnToIce >= PackSwq) {
          FirnToIce = PackSwq;
          PackSwq = 0.0;
          snow->pack_temp = 0.0;
          PackCC = 0.0;
        } else PackSwq -= FirnToIce;
      }
    }
    snow->pack_temp = PackCC / (ORC_CH_ICE * PackSwq);          /* 0/0 = NaN when all firn converted, as in the reference */
  } else snow->pack_temp = 0.0;
  glacier->accumulation = FirnToIce;
  Ice += SnowFall;
  snow->surf_water += RainFall;

  c.Dt = delta_t; c.Ra = aero_resist; c.ra_used_surface = ra_used_surface; c.Z = z2; c.z0_snow = z0_snow;
  c.AirDens = density; c.EactAir = vp; c.LongSnowIn = LongSnowIn; c.Lv = Le; c.Press = pressure; c.Rain = RainFall;
  c.NetShortUnder = NetShortSnow; c.Vpd = vpd; c.Wind = wind; c.OldTSurf = *OldTSurf; c.SnowDepth = snow->depth;
  c.SnowDensity = snow->density; c.SurfaceLiquidWater = snow->surf_water; c.SweSurfaceLayer = SurfaceSwq;
  c.Tair = air_temp; c.TGrnd = Tgrnd;
  c.AdvectedEnergy = &advection; c.AdvectedSensibleHeat = &advected_sensible_heat; c.DeltaColdContent = &deltaCC;
  c.GroundFlux = &grnd_flux; c.LatentHeat = &latent_heat; c.LatentHeatSub = &latent_heat_sub; c.NetLongUnder = NetLongSnow;
  c.RefreezeEnergy = &RefreezeEnergy; c.SensibleHeat = &sensible_heat; c.vapor_flux = &snow->vapor_flux;
  c.blowing_flux = &snow->blowing_flux; c.surface_flux = &snow->surface_flux;

  Qnet = gl_snowpack_energy_balance(0.0, &c);
  if (Qnet == 0.0) {
    snow->surf_temp = 0.0;
    if (RefreezeEnergy >= 0.0) {
      RefrozenWater = RefreezeEnergy / (ORC_LF * ORC_RHO_W) * delta_t;
      if (RefrozenWater > snow->surf_water) {
        RefrozenWater = snow->surf_water;
        RefreezeEnergy = RefrozenWater * ORC_LF * ORC_RHO_W / (delta_t);
      }
      SurfaceSwq += RefrozenWater;
      Ice += RefrozenWater;
      snow->surf_water -= RefrozenWater;
      if (snow->surf_water < 0.0) snow->surf_water = 0.0;
      SnowMelt = 0.0;
    } else SnowMelt = fabs(RefreezeEnergy) / (ORC_LF * ORC_RHO_W) * delta_t;
    if (snow->surf_water < -(snow->vapor_flux)) {
      snow->blowing_flux *= -(snow->surf_water / snow->vapor_flux);
      snow->vapor_flux = -(snow->surf_water);
      snow->surface_flux = -(snow->surf_water) - snow->blowing_flux;
      snow->surf_water = 0.0;
    } else snow->surf_water += snow->vapor_flux;
    if (SnowMelt < Ice) {
      if (SnowMelt <= PackSwq) {
        snow->surf_water += SnowMelt;
        PackSwq -= SnowMelt;
        Ice -= SnowMelt;
      } else {
        snow->surf_water += SnowMelt + snow->pack_water;
        snow->pack_water = 0.0;
        PackSwq = 0.0;
        Ice -= SnowMelt;
        SurfaceSwq = Ice;
      }
    } else {
      SnowMelt = Ice;
      snow->surf_water += Ice;
      SurfaceSwq = 0.0;
      snow->surf_temp = 0.0;
      PackSwq = 0.0;
      snow->pack_temp = 0.0;
      Ice = 0.0;
      RefreezeEnergy = RefreezeEnergy / fabs(RefreezeEnergy) * SnowMelt * ORC_LF * ORC_RHO_W / (delta_t);
    }
  } else {
    snow->surf_temp = orc_root_brent((double)(snow->surf_temp - ORC_SNOW_DT), (double)(snow->surf_temp + ORC_SNOW_DT),
                                     gl_snowpack_energy_balance, &c);
    if (orc_is_error(snow->surf_temp)) {
      if (m->opt.TFALLBACK) {
        snow->surf_temp = *OldTSurf;
        snow->surf_temp_fbflag = 1;
        snow->surf_temp_fbcount++;
      } else return -1;
    }
    if (!isnan(snow->surf_temp) && !orc_is_error(snow->surf_temp)) {
      Qnet = gl_snowpack_energy_balance(snow->surf_temp, &c);
      SnowMelt = 0.0;
      SurfaceSwq += snow->surf_water;
      Ice += snow->surf_water;
      snow->surf_water = 0.0;
      if (SurfaceSwq < -(snow->vapor_flux)) {
        snow->blowing_flux *= -(SurfaceSwq / snow->vapor_flux);
        snow->vapor_flux = -SurfaceSwq;
        snow->surface_flux = -SurfaceSwq - snow->blowing_flux;
        SurfaceSwq = 0.0;
        Ice = PackSwq;
      } else {
        SurfaceSwq += snow->vapor_flux;
        Ice += snow->vapor_flux;
      }
    }
  }
  MaxLiquidWater = ORC_LIQUID_WATER_CAPACITY * SurfaceSwq;
  if (snow->surf_water > MaxLiquidWater) { melt[0] = snow->surf_water - MaxLiquidWater; snow->surf_water = MaxLiquidWater; }
  else melt[0] = 0.0;
  snow->pack_water += melt[0];
  PackRefreezeEnergy = snow->pack_water * ORC_LF * ORC_RHO_W;
  if (PackCC < -PackRefreezeEnergy) {
    PackSwq += snow->pack_water;
    Ice += snow->pack_water;
    snow->pack_water = 0.0;
    if (PackSwq > 0.0) {
      PackCC = PackSwq * ORC_CH_ICE * snow->pack_temp + PackRefreezeEnergy;
      snow->pack_temp = PackCC / (ORC_CH_ICE * PackSwq);
      if (snow->pack_temp > 0.) snow->pack_temp = 0.;
    } else snow->pack_temp = 0.0;
  } else {
    snow->pack_temp = 0.0;
    DeltaPackSwq = -PackCC / (ORC_LF * ORC_RHO_W);
    snow->pack_water -= DeltaPackSwq;
    PackSwq += DeltaPackSwq;
    Ice += DeltaPackSwq;
  }
  MaxLiquidWater = ORC_LIQUID_WATER_CAPACITY * PackSwq;
  if (snow->pack_water > MaxLiquidWater) { melt[0] = snow->pack_water - MaxLiquidWater; snow->pack_water = MaxLiquidWater; }
  else melt[0] = 0.0;
  Ice = PackSwq + SurfaceSwq;
  if (Ice > ORC_MAX_SURFACE_SWE) {
    SurfaceCC = ORC_CH_ICE * snow->surf_temp * SurfaceSwq;
    PackCC = ORC_CH_ICE * snow->pack_temp * PackSwq;
    if (SurfaceSwq > ORC_MAX_SURFACE_SWE) {
      PackCC += SurfaceCC * (SurfaceSwq - ORC_MAX_SURFACE_SWE) / SurfaceSwq;
      SurfaceCC -= SurfaceCC * (SurfaceSwq - ORC_MAX_SURFACE_SWE) / SurfaceSwq;
      PackSwq += SurfaceSwq - ORC_MAX_SURFACE_SWE;
      SurfaceSwq -= SurfaceSwq - ORC_MAX_SURFACE_SWE;
    } else if (SurfaceSwq < ORC_MAX_SURFACE_SWE) {
      PackCC -= PackCC * (ORC_MAX_SURFACE_SWE - SurfaceSwq) / PackSwq;
      SurfaceCC += PackCC * (ORC_MAX_SURFACE_SWE - SurfaceSwq) / PackSwq;
      PackSwq -= ORC_MAX_SURFACE_SWE - SurfaceSwq;
      SurfaceSwq += ORC_MAX_SURFACE_SWE - SurfaceSwq;
    }
    snow->pack_temp = PackCC / (ORC_CH_ICE * PackSwq);
    snow->surf_temp = SurfaceCC / (ORC_CH_ICE * SurfaceSwq);
  } else {
    PackSwq = 0.0;
    PackCC = 0.0;
    snow->pack_temp = 0.0;
  }
  snow->swq = Ice + snow->pack_water + snow->surf_water;
  if (snow->swq == 0.0) { snow->surf_temp = 0.0; snow->pack_temp = 0.0; }
  snow->mass_error = (InitialSwq - snow->swq) + (RainFall + SnowFall) - melt[0] + snow->vapor_flux;
  snow->coldcontent = SurfaceCC;
  snow->vapor_flux *= -1.;
  *save_advection = advection;
  *save_deltaCC = deltaCC;
  *save_grnd_flux = grnd_flux;
  *save_latent = latent_heat;
  *save_latent_sub = latent_heat_sub;
  *save_sensible = sensible_heat;
  *save_advected_sensible = advected_sensible_heat;
  *save_refreeze_energy = RefreezeEnergy;
  *save_Qnet = Qnet;
  (void)SnowMelt;
  return 0;
}

/* snow_utility.c functions are static in orc_snow.c; small local copies of the two entry points used here would duplicate
 * code, so orc_snow.c exports thin wrappers */
double orc_snow_albedo_x(const orc_model *m, double new_snow, double swq, double depth, double albedo, double cold_content,
                         double dt, int last_snow, int MELTING, const orc_soil *sc);
double orc_snow_density_x(const orc_model *m, const orc_snow *snow, double new_snow, double sswq, double Tair, double dt);
double orc_new_snow_density_x(const orc_model *m, double air_temp);

/* solve_snow_glac.c:4-290 */
static double orc_solve_snow_glac(const orc_model *m, double BareAlbedo, double Tgrnd, double air_temp, double *AlbedoUnder,
                                  double *Le, double *LongUnderIn, double *NetLongSnow, double *NetShortSnow,
                                  double *ShortUnderIn, double *Torg_snow, const orc_vc *aero_resist, double *ra_used,
                                  double *coverage, double *delta_coverage, double *melt_energy, double *ppt, double *rainfall,
                                  const orc_vc *ref_height, const orc_vc *roughness, double *snow_inflow, double *snowfall,
                                  const orc_vc *wind_speed, int dt, int hidx, int *UnderStory, const orc_dmy *dmy,
                                  const orc_atmos *atmos, orc_energy *energy, orc_snow *snow, const orc_soil *sc, orc_glac *glacier) {
  double melt = 0., old_coverage, old_swq;
  const int day_in_year = dmy->day_in_year;
  *ppt = 0.;
  *melt_energy = 0.;
  *Le = (2.501e6 - 0.002361e6 * air_temp);
  *ShortUnderIn = atmos->shortwave[hidx];
  *LongUnderIn = atmos->longwave[hidx];
  snow->snow = 1;
  old_coverage = snow->coverage;
  energy->NetLongOver = 0;
  energy->LongOverIn = 0;
  *snow_inflow = *rainfall + *snowfall;
  old_swq = snow->swq;
  *UnderStory = ORC_SNOW_COVERED;
  if (snow->swq > 0. && *snowfall == 0.) {
    snow->last_snow++;
    snow->albedo = orc_snow_albedo_x(m, *snowfall, snow->swq, snow->depth, snow->albedo, snow->coldcontent, (double)dt,
                                     snow->last_snow, snow->MELTING, sc);
    *AlbedoUnder = (*coverage * snow->albedo + (1. - *coverage) * BareAlbedo);
  } else {
    snow->last_snow = 0;
    snow->albedo = sc->NEW_SNOW_ALB;
    *AlbedoUnder = snow->albedo;
  }
  *NetShortSnow = (1.0 - *AlbedoUnder) * (*ShortUnderIn);
  if (orc_snow_melt_glac(m, *Le, *NetShortSnow, Tgrnd, roughness->v[ORC_SNOW_COVERED], aero_resist->v[*UnderStory], &ra_used[0],
                         air_temp, (double)dt * ORC_SECPHOUR, atmos->density[hidx], *LongUnderIn, atmos->pressure[hidx],
                         *rainfall, *snowfall, atmos->vp[hidx], atmos->vpd[hidx], wind_speed->v[*UnderStory],
                         ref_height->v[*UnderStory], NetLongSnow, Torg_snow, &melt, &energy->error, &energy->advected_sensible,
                         &energy->advection, &energy->deltaCC, &energy->grnd_flux, &energy->latent, &energy->latent_sub,
                         &energy->refreeze_energy, &energy->sensible, snow, glacier))
    return ORC_ERROR;
  *ppt += melt;
  energy->AlbedoUnder = *AlbedoUnder;
  if (snow->swq > 0.) {
    if (!isnan(snow->surf_temp) && snow->surf_temp <= 0)
      snow->density = orc_snow_density_x(m, snow, *snowfall, old_swq, air_temp, (double)dt);
    else if (snow->last_snow == 0) snow->density = orc_new_snow_density_x(m, air_temp);
    snow->depth = 1000. * snow->swq / snow->density;
    if (snow->coldcontent >= 0 && ((sc->lat >= 0 && (day_in_year > 60 && day_in_year < 273))
                                   || (sc->lat < 0 && (day_in_year < 60 || day_in_year > 273))))
      snow->MELTING = 1;
    else if (snow->MELTING && *snowfall > ORC_TRACESNOW) snow->MELTING = 0;
    if (snow->swq > 0) snow->coverage = 1.; else snow->coverage = 0.;
  } else snow->coverage = 0.;
  *delta_coverage = old_coverage - snow->coverage;
  if (*delta_coverage != 0) {
    if (old_coverage > snow->coverage) {
      *coverage = (old_coverage);
      *AlbedoUnder = (*coverage - snow->coverage) / (1. - snow->coverage) * snow->albedo;
      *AlbedoUnder += (1. - *coverage) / (1. - snow->coverage) * BareAlbedo;
      *melt_energy = (*delta_coverage) * (energy->advection - energy->deltaCC + energy->latent + energy->latent_sub
                                           + energy->sensible + energy->refreeze_energy + energy->advected_sensible);
    } else {
      *coverage = snow->coverage;
      *delta_coverage = 0;
    }
  } else if (old_coverage == 0 && snow->coverage == 0) {
    *delta_coverage = 1.;
    *coverage = 0.;
    *melt_energy = (energy->advection - energy->deltaCC + energy->latent + energy->latent_sub + energy->sensible
                    + energy->refreeze_energy + energy->advected_sensible);
  }
  *NetLongSnow *= (snow->coverage + *delta_coverage);
  *NetShortSnow *= (snow->coverage + *delta_coverage);
  energy->latent *= (snow->coverage + *delta_coverage);
  energy->latent_sub *= (snow->coverage + *delta_coverage);
  energy->sensible *= (snow->coverage + *delta_coverage);
  if (snow->swq == 0) {
    snow->density = 0.; snow->depth = 0.; snow->surf_water = 0; snow->pack_water = 0; snow->surf_temp = 0; snow->pack_temp = 0;
    snow->coverage = 0; snow->swq_slope = 0; snow->store_snow = 1; snow->MELTING = 0;
  }
  *snowfall = 0;
  *rainfall = 0;
  energy->melt_energy *= -1.;
  return melt;
}

/* ---- bare-ice surface energy balance: GlacierEnergyBalance.c:15-92 + latent_heat_from_glacier.c:8-51 ---- */
typedef struct {
  double Dt, Ra, Z, z0_snow, AirDens, EactAir, LongSnowIn, Lv, Press, Rain, NetShortUnder, Vpd, Wind, OldTSurf, IceDepth, Tair, TGrnd;
  double *ra_used_surface, *AdvectedEnergy, *DeltaColdContent, *GroundFlux, *LatentHeat, *LatentHeatSub, *NetLongUnder,
         *SensibleHeat, *vapor_flux;
} gl_geb_ctx;

static double gl_glacier_energy_balance(double TSurf, void *vctx) {
  gl_geb_ctx *c = (gl_geb_ctx *)vctx;
  double Density = ORC_RHO_W, NetRad, RestTerm, TMean, OldTMean, Tmp, VaporMassFlux, Fbal, EsSnow;
  const double temp_IceDepth = c->IceDepth / 1000.;
  TMean = (TSurf + c->TGrnd) / 2;
  OldTMean = (c->OldTSurf + c->TGrnd) / 2;
  if (c->Wind > 0.0) *c->ra_used_surface = c->Ra / orc_stability_correction(c->Z, 0.f, TSurf, c->Tair, c->Wind, c->z0_snow);
  else *c->ra_used_surface = ORC_HUGE_RESIST;
  Tmp = TSurf + ORC_KELVIN;
  *c->NetLongUnder = c->LongSnowIn - ORC_STEFAN_B * Tmp * Tmp * Tmp * Tmp;
  NetRad = c->NetShortUnder + *c->NetLongUnder;
  *c->SensibleHeat = c->AirDens * ORC_CP * (c->Tair - TSurf) / *c->ra_used_surface;
  VaporMassFlux = *c->vapor_flux * Density / c->Dt;
  /* latent_heat_from_glacier */
  EsSnow = orc_svp(TSurf);
  VaporMassFlux = c->AirDens * (ORC_EPS / c->Press) * (c->EactAir - EsSnow) / *c->ra_used_surface;
  if (c->Vpd == 0.0 && VaporMassFlux < 0.0) VaporMassFlux = 0.0;
  if (TSurf >= 0.0) { *c->LatentHeat = c->Lv * VaporMassFlux; *c->LatentHeatSub = 0; }
  else {
    double Ls = (677. - 0.07 * TSurf) * ORC_JOULESPCAL * ORC_GRAMSPKG;
    *c->LatentHeatSub = Ls * VaporMassFlux;
    *c->LatentHeat = 0;
  }
  *c->vapor_flux = VaporMassFlux * c->Dt / Density;
  if (TSurf == 0) *c->AdvectedEnergy = (ORC_CH_WATER * (c->Tair) * c->Rain) / (c->Dt);
  else *c->AdvectedEnergy = 0.;
  *c->DeltaColdContent = ORC_CH_ICE * temp_IceDepth * (TMean - OldTMean) / (c->Dt);
  *c->GroundFlux = (ORC_GLAC_K_ICE + TSurf * (-0.0142)) * (c->TGrnd - TSurf) / temp_IceDepth;
  Fbal = NetRad + *c->SensibleHeat + *c->LatentHeat + *c->LatentHeatSub + *c->AdvectedEnergy;
  RestTerm = Fbal - *c->DeltaColdContent + *c->GroundFlux;
  if (TSurf == 0.0 && RestTerm >= 0.) RestTerm = 0.;
  return RestTerm;
}

/* glacier_melt.c:64-222 */
static int orc_glacier_melt(const orc_model *m, double Le, double NetShort, double Tgrnd, double z0_snow, double aero_resist,
                            double *ra_used_surface, double air_temp, double delta_t, double density, double LongIn,
                            double pressure, double rainfall, double vp, double vpd, double wind, double z2, double *NetLong,
                            double *OldTSurf, double *melt, double *save_Qnet, double *save_advection,
                            double *save_deltaCC_glac, double *save_glacier_melt_energy, double *save_grnd_flux,
                            double *save_latent, double *save_latent_sub, double *save_sensible, orc_glac *glacier,
                            const orc_soil *sc) {
  double Qnet, GlacMelt = 0, GlacCC = 0, RainFall, advection, deltaCC_glac, latent_heat, latent_heat_sub, sensible_heat,
         melt_energy = 0., grnd_flux;
  gl_geb_ctx c;
  RainFall = rainfall / 1000.;
  *OldTSurf = glacier->surf_temp;
  c.Dt = delta_t; c.Ra = aero_resist; c.ra_used_surface = ra_used_surface; c.Z = z2; c.z0_snow = z0_snow; c.AirDens = density;
  c.EactAir = vp; c.LongSnowIn = LongIn; c.Lv = Le; c.Press = pressure; c.Rain = RainFall; c.NetShortUnder = NetShort;
  c.Vpd = vpd; c.Wind = wind; c.OldTSurf = *OldTSurf; c.IceDepth = sc->GLAC_SURF_THICK; c.Tair = air_temp; c.TGrnd = Tgrnd;
  c.AdvectedEnergy = &advection; c.DeltaColdContent = &deltaCC_glac; c.GroundFlux = &grnd_flux; c.LatentHeat = &latent_heat;
  c.LatentHeatSub = &latent_heat_sub; c.NetLongUnder = NetLong; c.SensibleHeat = &sensible_heat;
  c.vapor_flux = &glacier->vapor_flux;
  Qnet = gl_glacier_energy_balance(0.0, &c);
  if (Qnet == 0.0) {
    glacier->surf_temp = 0.;
    melt_energy = NetShort + (*NetLong) + sensible_heat + latent_heat + latent_heat_sub + advection - deltaCC_glac;
    GlacMelt = melt_energy / (ORC_LF * ORC_RHO_W) * delta_t;
    GlacCC = 0.;
  } else {
    glacier->surf_temp = orc_root_brent((double)(glacier->surf_temp - ORC_SNOW_DT), (double)(glacier->surf_temp + ORC_SNOW_DT),
                                        gl_glacier_energy_balance, &c);
    if (orc_is_error(glacier->surf_temp)) {
      if (m->opt.TFALLBACK) {
        glacier->surf_temp = *OldTSurf;
        glacier->surf_temp_fbflag = 1;
        glacier->surf_temp_fbcount++;
      } else return -1;
    }
    if (!orc_is_error(glacier->surf_temp)) {
      Qnet = gl_glacier_energy_balance(glacier->surf_temp, &c);
      GlacMelt = 0.0;
      GlacCC = ORC_CH_ICE * glacier->surf_temp * sc->GLAC_SURF_THICK / 1000.;
    }
  }
  melt[0] = GlacMelt;
  glacier->cold_content = GlacCC;
  glacier->vapor_flux *= -1.;
  *save_advection = advection;
  *save_deltaCC_glac = deltaCC_glac;
  *save_glacier_melt_energy = melt_energy;
  *save_grnd_flux = grnd_flux;
  *save_latent = latent_heat;
  *save_latent_sub = latent_heat_sub;
  *save_sensible = sensible_heat;
  *save_Qnet = Qnet;
  return 0;
}

/* solve_glacier.c:5-104 */
static double orc_solve_glacier(const orc_model *m, double BareAlbedo, double Tgrnd, double air_temp, double *AlbedoUnder,
                                double *Le, double *LongUnderIn, double *NetLongGlac, double *NetShortGlac,
                                double *ShortUnderIn, double *Torg_snow, const orc_vc *aero_resist, double *ra_used,
                                double *melt_energy, double *ppt, double *rainfall, const orc_vc *ref_height,
                                const orc_vc *roughness, const orc_vc *wind_speed, int dt, int hidx, int *UnderStory,
                                const orc_atmos *atmos, orc_energy *energy, orc_glac *glacier, const orc_soil *sc) {
  double melt = 0.;
  *ppt = 0.;
  *melt_energy = 0.;
  *Le = (2.501e6 - 0.002361e6 * air_temp);
  *ShortUnderIn = atmos->shortwave[hidx];
  *LongUnderIn = atmos->longwave[hidx];
  *AlbedoUnder = BareAlbedo;
  *NetShortGlac = (1.0 - *AlbedoUnder) * (*ShortUnderIn);
  *UnderStory = ORC_GLACIER_SURF;
  if (orc_glacier_melt(m, *Le, *NetShortGlac, Tgrnd, roughness->v[ORC_SNOW_COVERED], aero_resist->v[*UnderStory], &ra_used[0],
                       air_temp, (double)dt * ORC_SECPHOUR, atmos->density[hidx], *LongUnderIn, atmos->pressure[hidx], *rainfall,
                       atmos->vp[hidx], atmos->vpd[hidx], wind_speed->v[*UnderStory], ref_height->v[*UnderStory], NetLongGlac,
                       Torg_snow, &melt, &energy->error, &energy->advection, &energy->deltaCC_glac,
                       &energy->glacier_melt_energy, &energy->grnd_flux, &energy->latent, &energy->latent_sub,
                       &energy->sensible, glacier, sc))
    return ORC_ERROR;
  *ppt = (melt + *rainfall / 1000.);
  energy->AlbedoUnder = *AlbedoUnder;
  *rainfall = 0;
  return melt;
}

/* surface_fluxes_glac.c:6-614 */
int orc_surface_fluxes_glac(const orc_model *m, orc_hru *h, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy,
                            double BareAlbedo, double ice0, double moist0, orc_vc *aero_resist, orc_vc *displacement,
                            orc_vc *ref_height, orc_vc *roughness, orc_vc *wind_speed, double *out_prec, double *out_rain,
                            double *out_snow) {
  const int NF = m->NF;
  int N_steps = 0, UnderStory = ORC_SNOW_COVERED, hidx = 0, endhidx = NF, step_dt = m->opt.snow_step, p, l;
  double LongUnderIn, NetLongSnow, NetShortSnow, OldTSurf, ShortUnderIn, Tair, Tgrnd, VPDcanopy, Le = 0, coverage, delta_coverage = 0,
         ppt, rainfall, snowfall, snow_inflow = 0, step_melt, step_melt_glac, step_melt_energy, step_out_prec, step_out_rain,
         step_out_snow, step_ppt, step_prec, rainOnly;
  double st_AlbedoUnder = 0, st_AtmosLatent = 0, st_AtmosLatentSub = 0, st_AtmosSensible = 0, st_LongUnderIn = 0,
         st_LongUnderOut = 0, st_NetLongAtmos = 0, st_NetLongUnder = 0, st_NetShortAtmos = 0, st_NetShortUnder = 0,
         st_ShortUnderIn = 0, st_advected_sensible = 0, st_advection = 0, st_deltaCC = 0, st_grnd_flux = 0, st_latent = 0,
         st_latent_sub = 0, st_melt_energy = 0, st_refreeze_energy = 0, st_sensible = 0, st_snow_flux = 0, st_deltaCC_glac = 0,
         st_glacier_flux = 0, st_glacier_melt_energy = 0, st_melt_glac = 0, st_vapor_flux_glac = 0, st_accum_glac = 0,
         st_melt = 0, st_vapor_flux = 0, st_blowing_flux = 0, st_surface_flux = 0, st_ppt = 0, st_cond_surface = 0,
         st_cond_overstory = 0;
  double ra_s[ORC_NPET], ra_o[ORC_NPET], ra_used[2], stability_factor[2], step_pot_evap[ORC_NPET], store_pot_evap[ORC_NPET];
  orc_energy step_energy;
  orc_snow step_snow;
  orc_glac step_glacier;
  orc_vc temp_aero_resist;
  (void)ice0; (void)moist0; (void)displacement;

  coverage = h->snow.coverage;
  step_energy = h->energy;
  step_snow = h->snow;
  step_glacier = h->glac;
  for (l = 0; l < 3; l++) h->layer[l].evap = 0;     /* step_layer[].evap = 0, written back at :548-552 */
  for (p = 0; p < ORC_NPET; p++) store_pot_evap[p] = 0;

  do {
    Tair = atmos->air_temp[hidx] + sc->Tfactor[h->band];
    step_prec = atmos->prec[hidx] / 1.0 * sc->Pfactor[h->band];
    rainOnly = orc_calc_rainonly(m, Tair, step_prec, sc->MAX_SNOW_TEMP, sc->MIN_RAIN_TEMP);
    snowfall = atmos->gauge_correction[1] * (step_prec - rainOnly) * sc->PADJ_S;      /* surface_fluxes_glac.c:244-245 */
    rainfall = atmos->gauge_correction[0] * rainOnly * sc->PADJ_R;
    step_out_prec = snowfall + rainfall;
    step_out_rain = rainfall;
    step_out_snow = snowfall;
    Tgrnd = ORC_GLAC_TEMP;
    VPDcanopy = 0.;
    if (m->opt.BLOWING && step_snow.swq > 0.) {                                     /* surface_fluxes_glac.c:260-274 */
      double Ls = (677. - 0.07 * step_snow.surf_temp) * 4.1868 * 1000.0;
      step_snow.blowing_flux = orc_calc_blowing_snow((double)step_dt, Tair, step_snow.last_snow, step_snow.surf_water,
                                                     wind_speed->v[ORC_SNOW_COVERED], Ls, atmos->density[hidx], atmos->vp[hidx],
                                                     roughness->v[ORC_SNOW_COVERED], ref_height->v[ORC_SNOW_COVERED], step_snow.depth,
                                                     h->lag_one, h->sigma_slope, step_snow.surf_temp, h->is_artificial_bare, h->fetch,
                                                     displacement->v[ORC_CANOPY], roughness->v[ORC_CANOPY], &step_snow.transport);
      if ((int)step_snow.blowing_flux == ORC_ERROR) return -1;
      step_snow.blowing_flux *= step_dt * ORC_SECPHOUR / ORC_RHO_W;
    } else step_snow.blowing_flux = 0.0;
    temp_aero_resist = aero_resist[ORC_NPET];
    ra_used[0] = h->aero_resist_surface;
    ra_used[1] = h->aero_resist_overstory;
    step_snow.canopy_vapor_flux = 0;
    step_snow.vapor_flux = 0;
    step_snow.surface_flux = 0;

    if (step_snow.swq > 0. || snowfall > 0.) {
      step_melt = orc_solve_snow_glac(m, BareAlbedo, Tgrnd, Tair, &h->energy.AlbedoUnder, &Le, &LongUnderIn, &NetLongSnow,
                                      &NetShortSnow, &ShortUnderIn, &OldTSurf, &temp_aero_resist, ra_used, &coverage,
                                      &delta_coverage, &step_melt_energy, &step_ppt, &rainfall, ref_height, roughness,
                                      &snow_inflow, &snowfall, wind_speed, step_dt, hidx, &UnderStory, dmy, atmos, &step_energy,
                                      &step_snow, sc, &step_glacier);
      if (step_melt == ORC_ERROR) return -1;
      step_melt_glac = 0.;
      step_glacier.vapor_flux = 0.;
      step_energy.glacier_flux = 0.;
      step_energy.deltaCC_glac = 0.;
      step_energy.glacier_melt_energy = 0.;
      step_energy.snow_flux = -step_energy.grnd_flux;
      step_energy.LongUnderOut = LongUnderIn - NetLongSnow;
    } else {
      step_melt_glac = orc_solve_glacier(m, BareAlbedo, Tgrnd, Tair, &h->energy.AlbedoUnder, &Le, &LongUnderIn, &NetLongSnow,
                                         &NetShortSnow, &ShortUnderIn, &OldTSurf, &temp_aero_resist, ra_used, &step_melt_energy,
                                         &step_ppt, &rainfall, ref_height, roughness, wind_speed, step_dt, hidx, &UnderStory,
                                         atmos, &step_energy, &step_glacier, sc);
      if (step_melt_glac == ORC_ERROR) return -1;
      step_melt = 0.;
      /* the reference resets these on hru.snow directly (:324-328); they are overwritten by step_snow at :467 */
      step_energy.deltaCC = 0.;
      step_energy.refreeze_energy = 0.;
      step_energy.snow_flux = 0.;
      step_energy.advected_sensible = 0.;
      step_energy.glacier_flux = -step_energy.grnd_flux;
      step_energy.LongUnderOut = LongUnderIn - NetLongSnow;
      step_glacier.accumulation = 0.;
    }
    step_energy.AtmosLatent = step_energy.latent;
    step_energy.AtmosLatentSub = step_energy.latent_sub;
    step_energy.AtmosSensible = step_energy.sensible;
    step_energy.NetLongAtmos = step_energy.NetLongUnder;
    step_energy.NetShortAtmos = step_energy.NetShortUnder;

    if (ra_used[0] == ORC_HUGE_RESIST) stability_factor[0] = ORC_HUGE_RESIST;
    else stability_factor[0] = ra_used[0] / aero_resist[ORC_NPET].v[UnderStory];
    if (ra_used[1] == ra_used[0]) stability_factor[1] = stability_factor[0];
    else {
      if (ra_used[1] == ORC_HUGE_RESIST) stability_factor[1] = ORC_HUGE_RESIST;
      else stability_factor[1] = ra_used[1] / aero_resist[ORC_NPET].v[ORC_CANOPY];
    }
    for (p = 0; p < ORC_NPET; p++) {
      if (stability_factor[0] == ORC_HUGE_RESIST) ra_s[p] = ORC_HUGE_RESIST;
      else ra_s[p] = aero_resist[p].v[UnderStory] * stability_factor[0];
      if (stability_factor[1] == ORC_HUGE_RESIST) ra_o[p] = ORC_HUGE_RESIST;
      else ra_o[p] = aero_resist[p].v[ORC_CANOPY] * stability_factor[1];
    }
    orc_compute_pot_evap(m, h->veg_index, dmy->month, m->opt.dt, atmos->shortwave[hidx], step_energy.NetLongAtmos, Tair, VPDcanopy,
                         sc->elevation, ra_s, ra_o, step_pot_evap);

    st_ppt += step_ppt;
    if (ra_used[0] > 0) st_cond_surface += 1 / ra_used[0]; else st_cond_surface += ORC_HUGE_RESIST;
    if (ra_used[1] > 0) st_cond_overstory += 1 / ra_used[1]; else st_cond_overstory += ORC_HUGE_RESIST;
    st_melt += step_melt;
    st_vapor_flux += step_snow.vapor_flux;
    st_surface_flux += step_snow.surface_flux;
    st_blowing_flux += step_snow.blowing_flux;
    out_prec[0] += step_out_prec * 1.0;
    out_rain[0] += step_out_rain * 1.0;
    out_snow[0] += step_out_snow * 1.0;
    st_AlbedoUnder += step_energy.AlbedoUnder;
    st_AtmosLatent += step_energy.AtmosLatent;
    st_AtmosLatentSub += step_energy.AtmosLatentSub;
    st_AtmosSensible += step_energy.AtmosSensible;
    st_LongUnderIn += LongUnderIn;
    st_LongUnderOut += step_energy.LongUnderOut;
    st_NetLongAtmos += NetLongSnow;
    st_NetLongUnder += NetLongSnow;
    st_NetShortAtmos += NetShortSnow;
    st_NetShortUnder += NetShortSnow;
    st_ShortUnderIn += ShortUnderIn;
    st_latent += step_energy.latent;
    st_latent_sub += step_energy.latent_sub;
    st_melt_energy += step_melt_energy;
    st_sensible += step_energy.sensible;
    st_grnd_flux += step_energy.grnd_flux;
    st_melt_glac += step_melt_glac;
    st_vapor_flux_glac += step_glacier.vapor_flux;
    st_accum_glac += step_glacier.accumulation;
    st_glacier_flux += step_energy.glacier_flux;
    st_deltaCC_glac += step_energy.deltaCC_glac;
    st_glacier_melt_energy += step_energy.glacier_melt_energy;
    st_advected_sensible += step_energy.advected_sensible * (step_snow.coverage + delta_coverage);
    st_advection += step_energy.advection * (step_snow.coverage + delta_coverage);
    st_deltaCC += step_energy.deltaCC * (step_snow.coverage + delta_coverage);
    st_snow_flux += step_energy.snow_flux * (step_snow.coverage + delta_coverage);
    st_refreeze_energy += step_energy.refreeze_energy * (step_snow.coverage + delta_coverage);
    for (p = 0; p < ORC_NPET; p++) store_pot_evap[p] += step_pot_evap[p];
    N_steps++;
    hidx += 1;
  } while (hidx < endhidx);

  h->glac = step_glacier;
  h->glac.melt = st_melt_glac;
  h->glac.vapor_flux = st_vapor_flux_glac;
  h->glac.accumulation = st_accum_glac;
  h->snow = step_snow;
  h->snow.vapor_flux = st_vapor_flux;
  h->snow.blowing_flux = st_blowing_flux;
  h->snow.surface_flux = st_surface_flux;
  h->snow.canopy_vapor_flux = 0;
  h->snow.melt = st_melt;
  ppt = st_ppt;
  h->glac.mass_balance = out_prec[0] / 1000. - ppt - h->snow.vapor_flux - h->glac.vapor_flux;
  h->glac.ice_mass_balance = h->glac.accumulation - h->glac.melt - h->glac.vapor_flux;

  h->energy = step_energy;
  {
    orc_energy *e = &h->energy;
    const double N = (double)N_steps;
    e->AlbedoOver = 0 / N; e->AlbedoUnder = st_AlbedoUnder / N; e->AtmosLatent = st_AtmosLatent / N;
    e->AtmosLatentSub = st_AtmosLatentSub / N; e->AtmosSensible = st_AtmosSensible / N; e->LongOverIn = 0 / N;
    e->LongUnderIn = st_LongUnderIn / N; e->LongUnderOut = st_LongUnderOut / N; e->NetLongAtmos = st_NetLongAtmos / N;
    e->NetLongOver = 0 / N; e->NetLongUnder = st_NetLongUnder / N; e->NetShortAtmos = st_NetShortAtmos / N;
    e->NetShortGrnd = 0 / N; e->NetShortOver = 0 / N; e->NetShortUnder = st_NetShortUnder / N; e->ShortOverIn = 0 / N;
    e->ShortUnderIn = st_ShortUnderIn / N; e->advected_sensible = st_advected_sensible / N; e->canopy_advection = 0 / N;
    e->canopy_latent = 0 / N; e->canopy_latent_sub = 0 / N; e->canopy_refreeze = 0 / N; e->canopy_sensible = 0 / N;
    e->deltaH = 0 / N; e->fusion = 0 / N; e->grnd_flux = st_grnd_flux / N; e->latent = st_latent / N;
    e->latent_sub = st_latent_sub / N; e->melt_energy = st_melt_energy / N; e->sensible = st_sensible / N;
    e->glacier_flux = st_glacier_flux / N; e->deltaCC_glac = st_deltaCC_glac / N;
    e->glacier_melt_energy = st_glacier_melt_energy / N; e->advection = st_advection / N; e->deltaCC = st_deltaCC / N;
    e->refreeze_energy = st_refreeze_energy / N; e->snow_flux = st_snow_flux / N;
    e->Tcanopy = 0.;
  }
  h->veg.throughfall = 0;
  h->veg.canopyevap = 0;
  if (st_cond_surface > 0 && st_cond_surface < ORC_HUGE_RESIST) h->aero_resist_surface = 1 / (st_cond_surface / (double)N_steps);
  else if (st_cond_surface >= ORC_HUGE_RESIST) h->aero_resist_surface = 0;
  else h->aero_resist_surface = ORC_HUGE_RESIST;
  if (st_cond_overstory > 0 && st_cond_overstory < ORC_HUGE_RESIST) h->aero_resist_overstory = 1 / (st_cond_overstory / (double)N_steps);
  else if (st_cond_overstory >= ORC_HUGE_RESIST) h->aero_resist_overstory = 0;
  else h->aero_resist_overstory = ORC_HUGE_RESIST;
  for (p = 0; p < ORC_NPET; p++) h->pot_evap[p] = store_pot_evap[p] / (double)N_steps;

  /* glacier linear reservoir and runoff, :580-601 */
  h->glac.inflow = ppt + 0.0;
  ppt = h->excess_moist;
  h->excess_moist = 0.;
  h->glac.outflow_coef = sc->GLAC_KMIN + sc->GLAC_DK * exp(-sc->GLAC_A * h->snow.swq);
  h->glac.water_storage += h->glac.inflow;
  h->glac.outflow = h->glac.outflow_coef * h->glac.water_storage;
  h->glac.water_storage -= h->glac.outflow;
  h->inflow = ppt;
  if (orc_runoff(m, h, sc, ppt) != 0) return -1;
  h->runoff += (h->glac.outflow * 1000.);
  return 0;
}

/* GlacierMassBalanceResult::GlacierMassBalanceResult (GlacierMassBalanceResult.c:34-73) + GraphingEquation
 * (GraphingEquation.c:8-125): quadratic least-squares fit of accumulated mass balance against band elevation through
 * closed-form normal equations.  hrus: the cell's HRUs in hruList order.  eq[4] = b0, b1, b2, fitError. */
void orc_glacier_mass_balance_fit(const orc_soil *sc, orc_hru **hrus, int nhru, double *eq) {
  double X[VIC_MAX_BANDS * 8 + 8], Y[VIC_MAX_BANDS * 8 + 8];
  int np = 0, i, j, k;
  double b0 = 0, b1 = 0, b2 = 0, fit = -1;
  for (i = 0; i < nhru; i++) {
    const orc_hru *h = hrus[i];
    if (h->is_glacier && !isnan(h->glac.cum_mass_balance)) {           /* IS_VALID */
      const double x = (double)(float)sc->BandElev[h->band];
      int found = 0;
      for (j = 0; j < np; j++)
        if (X[j] == x) { Y[j] += h->glac.cum_mass_balance; found = 1; }
      if (!found && np < (int)(sizeof(X) / sizeof(X[0]))) { X[np] = x; Y[np] = h->glac.cum_mass_balance; np++; }
    }
  }
  /* "Remove graph points which are meaningless": elevation 0 (lastElevation stays 0, GlacierMassBalanceResult.c:58-66) */
  for (i = 0, k = 0; i < np; i++)
    if (!(X[i] == 0 && X[i] <= 0)) { X[k] = X[i]; Y[k] = Y[i]; k++; }
  np = k;
  if (np == 1) { b0 = Y[0]; b1 = 0; b2 = 0; }
  else if (np == 2) {
    double slope = (Y[1] - Y[0]) / (X[1] - X[0]);
    b0 = Y[0] - slope * X[0]; b1 = slope; b2 = 0;
  } else if (np >= 3) {
    double sumx4 = 0, sumx3 = 0, sumx2 = 0, sumx1 = 0, det, inv[3][3], a[3] = {0, 0, 0};
    const int size = np;
    for (i = 0; i < np; i++) {
      sumx4 += X[i] * X[i] * X[i] * X[i];
      sumx3 += X[i] * X[i] * X[i];
      sumx2 += X[i] * X[i];
      sumx1 += X[i];
    }
    det = (sumx4 * sumx2 * size) + (sumx3 * sumx1 * sumx2) + (sumx2 * sumx3 * sumx1) - (sumx2 * sumx2 * sumx2) - (sumx1 * sumx1 * sumx4)
          - (size * sumx3 * sumx3);
    inv[0][0] = size * sumx2 - sumx1 * sumx1; inv[0][1] = -(size * sumx3 - sumx1 * sumx2); inv[0][2] = sumx1 * sumx3 - sumx2 * sumx2;
    inv[1][0] = -(size * sumx3 - sumx2 * sumx1); inv[1][1] = size * sumx4 - sumx2 * sumx2; inv[1][2] = -(sumx1 * sumx4 - sumx3 * sumx2);
    inv[2][0] = sumx1 * sumx3 - sumx2 * sumx2; inv[2][1] = -(sumx1 * sumx4 - sumx2 * sumx3); inv[2][2] = sumx2 * sumx4 - sumx3 * sumx3;
    for (i = 0; i < 3; i++) {
      for (j = 0; j < np; j++) {
        const double stuff = inv[i][0] * (X[j] * X[j]) + inv[i][1] * X[j] + inv[i][2] * 1;
        a[i] += stuff * Y[j];
      }
      a[i] /= det;
    }
    b0 = a[2]; b1 = a[1]; b2 = a[0];
  }
  if (np > 0) {
    fit = 0;
    for (i = 0; i < np; i++) {
      const double curve = b0 + b1 * X[i] + b2 * (X[i] * X[i]);
      fit += fabs(curve - Y[i]);
    }
  }
  eq[0] = b0; eq[1] = b1; eq[2] = b2; eq[3] = fit;
}
