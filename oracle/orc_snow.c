/*
 * orc_snow.c — TEST INFRASTRUCTURE (CPU oracle): snow accumulation / ablation,
 * canopy snow interception and their energy balances.
 */
#include "orc.h"

/* calc_rainonly.c:12-103 (mu == 1) */
double orc_calc_rainonly(const orc_model *m, double air_temp, double prec, double MAX_SNOW_TEMP, double MIN_RAIN_TEMP) {
  const double MIN_PREC = 1.e-5;
  double rainonly = 0;
  if (m->opt.TEMP_TH_TYPE == VIC_TEMP_TH_VIC_412) {
    if (air_temp < MAX_SNOW_TEMP && air_temp > MIN_RAIN_TEMP)
      rainonly = (air_temp - MIN_RAIN_TEMP) / (MAX_SNOW_TEMP - MIN_RAIN_TEMP) * prec;
    else if (air_temp >= MAX_SNOW_TEMP) rainonly = prec;
  } else if (m->opt.TEMP_TH_TYPE == VIC_TEMP_TH_KIENZLE) {
    double rfrac, TT = MIN_RAIN_TEMP, TR = MAX_SNOW_TEMP, D = 1.4 * TR;
    double E1 = 5. * pow((air_temp - TT) / D, 3.0);
    double E2 = 6.76 * pow((air_temp - TT) / D, 2.0);
    double E3 = 3.19 * (air_temp - TT) / D;
    if (air_temp <= TT) rfrac = E1 + E2 + E3 + 0.5;
    else rfrac = E1 - E2 + E3 + 0.5;
    if (rfrac < 0.) rfrac = 0.;
    if (rfrac > 1.) rfrac = 1.;
    rainonly = rfrac * prec;
  }
  if (rainonly < MIN_PREC) rainonly = 0.;
  if ((prec - rainonly) < MIN_PREC) rainonly = prec;
  return rainonly;
}

/* snow_utility.c:199-226 */
static double orc_new_snow_density(const orc_model *m, double air_temp) {
  double dn;
  if (m->opt.SNOW_DENSITY == VIC_DENS_SNTHRM) dn = 67.9 + 51.3 * exp(air_temp / 2.6);
  else {
    air_temp = air_temp * 9. / 5. + 32.;
    if (air_temp > 0) dn = (double)ORC_NEW_SNOW_DENSITY + 1000. * (air_temp / 100.) * (air_temp / 100.);
    else dn = (double)ORC_NEW_SNOW_DENSITY;
  }
  return dn;
}

/* snow_utility.c:9-196 */
static double orc_snow_density(const orc_model *m, const orc_snow *snow, double new_snow, double sswq, double Tair, double dt) {
  const double MAX_CHANGE = 0.9;
  double density_new, density, depth, swq, Tavg;
  if (new_snow > 0.) density_new = orc_new_snow_density(m, Tair);
  else density_new = 0.0;
  Tavg = snow->surf_temp + ORC_KELVIN;
  if (m->opt.SNOW_DENSITY == VIC_DENS_SNTHRM) {
    double dexpf, dm, c3, c4, ddz1, ddz2, f, Ps, CR;
    if (new_snow > 0.) { if (snow->depth > 0.0) density = snow->density; else density = density_new; }
    else density = snow->density;
    dexpf = exp(-ORC_SNDENS_C1 * (ORC_KELVIN - Tavg));
    if (new_snow > 0.0 && density_new > 0.0) dm = (ORC_SNDENS_DMLIMIT > 1.15 * density_new) ? ORC_SNDENS_DMLIMIT : 1.15 * density_new;
    else dm = ORC_SNDENS_DMLIMIT;
    if (density <= dm) { c3 = 1.0; c4 = 1.0; }
    else { c3 = exp(-0.046 * (density - dm)); c4 = 1.0; }
    if ((snow->surf_water + snow->pack_water) / snow->depth > 0.01) c4 = 2.0;
    ddz1 = -ORC_SNDENS_C2 * c3 * c4 * dexpf;
    f = ORC_SNDENS_F;
    swq = new_snow / 1000. + f * sswq;
    if (new_snow > 0.0) {
      Ps = 0.5 * ORC_G * ORC_RHO_W * swq;
      ddz2 = -Ps / ORC_SNDENS_ETA0 * exp(-(-ORC_SNDENS_C5 * (Tavg - ORC_KELVIN) + ORC_SNDENS_C6 * density));
    } else ddz2 = 0.0;
    CR = -ddz1 - ddz2;
    density = density * (1 + CR * dt * ORC_SECPHOUR);
  } else {
    double delta_depth, depth_new, overburden, viscosity;
    depth = snow->depth;
    swq = sswq;
    if (new_snow > 0) {
      if (depth > 0.) {
        delta_depth = (((new_snow / 25.4) * (depth / 0.0254)) / (swq / 0.0254) * pow((depth / 0.0254) / 10., 0.35)) * 0.0254;
        if (delta_depth > MAX_CHANGE * depth) delta_depth = MAX_CHANGE * depth;
        depth_new = new_snow / density_new;
        depth = depth - delta_depth + depth_new;
        swq += new_snow / 1000.;
        density = 1000. * swq / depth;
      } else {
        density = density_new;
        swq += new_snow / 1000.;
        depth = 1000. * swq / density;
      }
    } else density = 1000. * swq / snow->depth;
    if (depth > 0.) {
      overburden = 0.5 * ORC_G * ORC_RHO_W * swq;
      viscosity = ORC_SNDENS_ETA0 * exp(-ORC_SNDENS_C5 * (Tavg - ORC_KELVIN) + ORC_SNDENS_C6 * density);
      delta_depth = overburden / viscosity * depth * dt * ORC_SECPHOUR;
      if (delta_depth > MAX_CHANGE * depth) delta_depth = MAX_CHANGE * depth;
      depth -= delta_depth;
      density = 1000. * swq / depth;
    }
  }
  return density;
}

/* snow_utility.c:229-307 */
static double orc_snow_albedo(const orc_model *m, double new_snow, double swq, double depth, double albedo, double cold_content,
                              double dt, int last_snow, int MELTING, const orc_soil *sc) {
  if (new_snow > ORC_TRACESNOW && cold_content < 0.0) albedo = sc->NEW_SNOW_ALB;
  else if (swq > 0.0) {
    if (m->opt.SNOW_ALBEDO == VIC_SNOW_ALBEDO_SUN1999) {
      if (depth > 0.025) albedo = 0.5 + (albedo - 0.5) * exp(-0.01 * dt / 24);
      else if (cold_content < 0.0) albedo = albedo - 0.006 * dt / 24;
      else albedo = albedo - 0.071 * dt / 24;
      if (albedo < 0) albedo = 0;
    } else {
      if (cold_content < 0.0 && !MELTING)
        albedo = sc->NEW_SNOW_ALB * pow(sc->SNOW_ALB_ACCUM_A, pow((double)last_snow * dt / 24., sc->SNOW_ALB_ACCUM_B));
      else
        albedo = sc->NEW_SNOW_ALB * pow(sc->SNOW_ALB_THAW_A, pow((double)last_snow * dt / 24., sc->SNOW_ALB_THAW_B));
    }
  } else albedo = 0;
  return albedo;
}

/* latent_heat_from_snow.c:8-68 */
void orc_latent_heat_from_snow(double AirDens, double EactAir, double Lv, double Press, double Ra, double TMean, double Vpd,
                               double *LatentHeat, double *LatentHeatSub, double *VaporMassFlux, double *BlowingMassFlux,
                               double *SurfaceMassFlux) {
  double EsSnow = orc_svp(TMean);
  *SurfaceMassFlux = AirDens * (ORC_EPS / Press) * (EactAir - EsSnow) / Ra;
  if (Vpd == 0.0 && *SurfaceMassFlux < 0.0) *SurfaceMassFlux = 0.0;
  *VaporMassFlux = *SurfaceMassFlux + *BlowingMassFlux;
  if (TMean >= 0.0) {
    *LatentHeat = Lv * (*VaporMassFlux);
    *LatentHeatSub = 0;
  } else {
    double Ls = (677. - 0.07 * TMean) * ORC_JOULESPCAL * ORC_GRAMSPKG;
    *LatentHeatSub = Ls * (*VaporMassFlux);
    *LatentHeat = 0;
  }
}

/* ---- snow pack surface energy balance: SnowPackEnergyBalance.{h,c} ---- */
typedef struct {
  double Dt, Ra, Z, z0_snow, AirDens, EactAir, LongSnowIn, Lv, Press, Rain, NetShortUnder, Vpd, Wind, OldTSurf,
         SnowDepth, SnowDensity, SurfaceLiquidWater, SweSurfaceLayer, Tair, TGrnd;
  double *ra_used_surface;
  double *AdvectedEnergy, *AdvectedSensibleHeat, *DeltaColdContent, *GroundFlux, *LatentHeat, *LatentHeatSub,
         *NetLongUnder, *RefreezeEnergy, *SensibleHeat, *vapor_flux, *blowing_flux, *surface_flux;
} orc_spe_ctx;

/* SnowPackEnergyBalance.c:85-197 */
static double orc_snowpack_energy_balance(double TSurf, void *vctx) {
  orc_spe_ctx *c = (orc_spe_ctx *)vctx;
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

/* snow_melt.c:119-564.  Returns 0, or -1 when the surface-temperature solve fails with TFALLBACK off. */
int orc_snow_melt(const orc_model *m, double Le, double NetShortSnow, double Tcanopy, double Tgrnd, double z0_snow,
                  double aero_resist, double *ra_used_surface, double air_temp, double delta_t, double density,
                  double grnd_flux, double LongSnowIn, double pressure, double rainfall, double snowfall, double vp,
                  double vpd, double wind, double z2, double *NetLongSnow, double *OldTSurf, double *melt,
                  double *save_Qnet, double *save_advected_sensible, double *save_advection, double *save_deltaCC,
                  double *save_grnd_flux, double *save_latent, double *save_latent_sub, double *save_refreeze_energy,
                  double *save_sensible, int UNSTABLE_SNOW, orc_snow *snow) {
  double DeltaPackCC, DeltaPackSwq, Ice, InitialSwq, MaxLiquidWater, PackCC, PackSwq, Qnet, RefreezeEnergy,
         PackRefreezeEnergy, RefrozenWater, SnowFallCC, SnowMelt = 0, SurfaceCC, SurfaceSwq, SnowFall, RainFall;
  double advection, deltaCC, latent_heat, latent_heat_sub, sensible_heat, advected_sensible_heat, melt_energy = 0.;
  orc_spe_ctx c;

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
  if (PackSwq > 0.0) snow->pack_temp = PackCC / (ORC_CH_ICE * PackSwq); else snow->pack_temp = 0.0;
  Ice += SnowFall;
  snow->surf_water += RainFall;

  /* the three SnowPackEnergyBalance objects of snow_melt.c:229,325,378 capture identical values */
  c.Dt = delta_t; c.Ra = aero_resist; c.ra_used_surface = ra_used_surface; c.Z = z2; c.z0_snow = z0_snow;
  c.AirDens = density; c.EactAir = vp; c.LongSnowIn = LongSnowIn; c.Lv = Le; c.Press = pressure; c.Rain = RainFall;
  c.NetShortUnder = NetShortSnow; c.Vpd = vpd; c.Wind = wind; c.OldTSurf = *OldTSurf; c.SnowDepth = snow->depth;
  c.SnowDensity = snow->density; c.SurfaceLiquidWater = snow->surf_water; c.SweSurfaceLayer = SurfaceSwq;
  c.Tair = Tcanopy; c.TGrnd = Tgrnd;
  c.AdvectedEnergy = &advection; c.AdvectedSensibleHeat = &advected_sensible_heat; c.DeltaColdContent = &deltaCC;
  c.GroundFlux = &grnd_flux; c.LatentHeat = &latent_heat; c.LatentHeatSub = &latent_heat_sub; c.NetLongUnder = NetLongSnow;
  c.RefreezeEnergy = &RefreezeEnergy; c.SensibleHeat = &sensible_heat; c.vapor_flux = &snow->vapor_flux;
  c.blowing_flux = &snow->blowing_flux; c.surface_flux = &snow->surface_flux;

  Qnet = orc_snowpack_energy_balance(0.0, &c);                                      /* :245 */

  if (!UNSTABLE_SNOW) {
    if (Qnet == 0.0) {                                                              /* :252-319 */
      snow->surf_temp = 0.0;
      if (RefreezeEnergy >= 0.0) {
        RefrozenWater = RefreezeEnergy / (ORC_LF * ORC_RHO_W) * delta_t;
        if (RefrozenWater > snow->surf_water) {
          RefrozenWater = snow->surf_water;
          RefreezeEnergy = RefrozenWater * ORC_LF * ORC_RHO_W / (delta_t);
        }
        melt_energy += RefreezeEnergy;
        SurfaceSwq += RefrozenWater;
        Ice += RefrozenWater;
        snow->surf_water -= RefrozenWater;
        if (snow->surf_water < 0.0) snow->surf_water = 0.0;
        SnowMelt = 0.0;
      } else {
        SnowMelt = fabs(RefreezeEnergy) / (ORC_LF * ORC_RHO_W) * delta_t;
        melt_energy += RefreezeEnergy;
      }
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
        melt_energy -= RefreezeEnergy;
        RefreezeEnergy = RefreezeEnergy / fabs(RefreezeEnergy) * SnowMelt * ORC_LF * ORC_RHO_W / (delta_t);
        melt_energy += RefreezeEnergy;
      }
    } else {                                                                        /* :322-424 */
      if (SurfaceSwq > ORC_MIN_SWQ_EB_THRES) {
        snow->surf_temp = orc_root_brent((double)(snow->surf_temp - ORC_SNOW_DT), (double)(snow->surf_temp + ORC_SNOW_DT),
                                         orc_snowpack_energy_balance, &c);
        if (orc_is_error(snow->surf_temp)) {
          if (m->opt.TFALLBACK) {
            snow->surf_temp = *OldTSurf;
            snow->surf_temp_fbflag = 1;
            snow->surf_temp_fbcount++;
          } else return -1;
        }
      } else snow->surf_temp = NAN;                                                 /* thin pack: solved with the ground */
      if (!isnan(snow->surf_temp) && !orc_is_error(snow->surf_temp)) {
        Qnet = orc_snowpack_energy_balance(snow->surf_temp, &c);
        SnowMelt = 0.0;
        SurfaceSwq += snow->surf_water;
        Ice += snow->surf_water;
        snow->surf_water = 0.0;
        melt_energy += snow->surf_water * ORC_LF * ORC_RHO_W / (delta_t);          /* adds 0 (Appendix C #6) */
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
  } else snow->surf_temp = NAN;

  MaxLiquidWater = ORC_LIQUID_WATER_CAPACITY * SurfaceSwq;                          /* :447-453 */
  if (snow->surf_water > MaxLiquidWater) {
    melt[0] = snow->surf_water - MaxLiquidWater;
    snow->surf_water = MaxLiquidWater;
  } else melt[0] = 0.0;
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
  if (snow->pack_water > MaxLiquidWater) {
    melt[0] = snow->pack_water - MaxLiquidWater;
    snow->pack_water = MaxLiquidWater;
  } else melt[0] = 0.0;
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
  melt[0] *= 1000.;
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
  (void)melt_energy; (void)SnowMelt;
  return 0;
}

/* massrelease.c:40-93 (tail recursion written as a loop) */
static void orc_mass_release(double *InterceptedSnow, double *TempInterceptionStorage, double *ReleasedMass, double *Drip) {
  for (;;) {
    if (*InterceptedSnow > ORC_MIN_INTERCEPTION_STORAGE) {
      double Threshold = 0.10 * *InterceptedSnow, MaxRelease = 0.17 * *InterceptedSnow;
      if ((*TempInterceptionStorage) >= Threshold) {
        double TempReleasedMass;
        *Drip += Threshold;
        *InterceptedSnow -= Threshold;
        *TempInterceptionStorage -= Threshold;
        if (*InterceptedSnow < ORC_MIN_INTERCEPTION_STORAGE) TempReleasedMass = 0.0;
        else TempReleasedMass = fmin((*InterceptedSnow - ORC_MIN_INTERCEPTION_STORAGE), MaxRelease);
        *ReleasedMass += TempReleasedMass;
        *InterceptedSnow -= TempReleasedMass;
        continue;
      } else {
        double TempDrip = fmin(*TempInterceptionStorage, *InterceptedSnow);
        *Drip += TempDrip;
        *InterceptedSnow -= TempDrip;
      }
    } else {
      double TempDrip = fmin(*TempInterceptionStorage, *InterceptedSnow);
      *Drip += TempDrip;
      *InterceptedSnow -= TempDrip;
      *TempInterceptionStorage = 0.0;
    }
    return;
  }
}

/* ---- canopy energy balance: canopy_energy_bal.h + func_canopy_energy_bal.c:9-149 ---- */
typedef struct {
  const orc_model *m;
  const orc_soil *sc;
  int month, veg_idx;
  double delta_t, AirDens, EactAir, Press, Le, Tcanopy, Vpd;
  double *Evap;
  const orc_vc *Ra, *wind_speed, *displacement, *ref_height, *roughness;
  double *ra_used;            /* [0] surface, [1] overstory */
  double *Rainfall;           /* rainfall[WET], m */
  const double *root;
  double IntRain, IntSnow;    /* IntRainOrg, *IntSnow at construction (snow_intercept.c:330-339) */
  orc_layer *layer;
  orc_vegvar *vv;             /* vv->Wdew aliases *IntRain, in m while inside snow_intercept */
  double LongOverIn, LongUnderOut, NetShortOver;
  double *AdvectedEnergy, *LatentHeat, *LatentHeatSub, *LongOverOut, *NetLongOver, *NetRadiation, *RefreezeEnergy,
         *SensibleHeat, *VaporMassFlux;
} orc_ceb_ctx;

static double orc_canopy_energy_bal(double Tfoliage, void *vctx) {
  orc_ceb_ctx *c = (orc_ceb_ctx *)vctx;
  const int AR = c->m->opt.AERO_RESIST_CANSNOW;
  double Tmp, RestTerm;
  Tmp = Tfoliage + ORC_KELVIN;
  *c->LongOverOut = ORC_STEFAN_B * (Tmp * Tmp * Tmp * Tmp);
  *c->NetRadiation = c->NetShortOver + c->LongOverIn + c->LongUnderOut - 2 * (*c->LongOverOut);
  *c->NetLongOver = c->LongOverIn - (*c->LongOverOut);
  if (c->IntSnow > 0) {
    double EsSnow, Ls;
    c->ra_used[0] = c->Ra->v[ORC_SNOW_FREE];
    c->ra_used[1] = c->Ra->v[ORC_CANOPY];
    if (AR == VIC_AR_COMBO || AR == VIC_AR_406 || AR == VIC_AR_406_LS || AR == VIC_AR_406_FULL) c->ra_used[1] *= 10.;
    EsSnow = orc_svp(Tfoliage);
    if (AR == VIC_AR_COMBO || AR == VIC_AR_410) {
      if (c->wind_speed->v[ORC_CANOPY] > 0.0)
        c->ra_used[1] /= orc_stability_correction(c->ref_height->v[ORC_CANOPY], c->displacement->v[ORC_CANOPY], Tfoliage,
                                                  c->Tcanopy, c->wind_speed->v[ORC_CANOPY], c->roughness->v[ORC_CANOPY]);
      else c->ra_used[1] = ORC_HUGE_RESIST;
    }
    *c->VaporMassFlux = c->AirDens * (ORC_EPS / c->Press) * (c->EactAir - EsSnow) / c->ra_used[1] / ORC_RHO_W;
    if (c->Vpd == 0.0 && *c->VaporMassFlux < 0.0) *c->VaporMassFlux = 0.0;
    Ls = (677. - 0.07 * Tfoliage) * ORC_JOULESPCAL * ORC_GRAMSPKG;
    *c->LatentHeatSub = Ls * *c->VaporMassFlux * ORC_RHO_W;
    *c->LatentHeat = 0;
    *c->Evap = 0;
    c->vv->throughfall = 0;
    if (AR == VIC_AR_406) c->ra_used[1] /= 10;
  } else {
    double wdew_mm, prec_mm;
    if (AR == VIC_AR_406_FULL || AR == VIC_AR_410 || AR == VIC_AR_COMBO) {
      c->ra_used[0] = c->Ra->v[ORC_SNOW_FREE];
      c->ra_used[1] = c->Ra->v[ORC_CANOPY];
    } else {
      c->ra_used[0] = c->Ra->v[ORC_SNOW_FREE];
      c->ra_used[1] = c->Ra->v[ORC_SNOW_FREE];
    }
    /* Wdew[WET] is the same memory as veg_var_wet->Wdew (func_canopy_energy_bal.c:97-106) */
    c->vv->Wdew = c->IntRain * 1000.;
    wdew_mm = c->vv->Wdew;
    prec_mm = *c->Rainfall * 1000;
    *c->Evap = orc_canopy_evap(c->m, c->layer, c->vv, 0, c->veg_idx, c->month, &wdew_mm, c->delta_t, *c->NetRadiation,
                               c->Vpd, c->NetShortOver, c->Tcanopy, c->ra_used[1], c->sc->elevation, prec_mm, c->sc, c->root);
    c->vv->Wdew /= 1000.;
    *c->LatentHeat = c->Le * *c->Evap * ORC_RHO_W;
    *c->LatentHeatSub = 0;
  }
  *c->SensibleHeat = c->AirDens * ORC_CP * (c->Tcanopy - Tfoliage) / c->ra_used[1];
  *c->AdvectedEnergy = (4186.8 * c->Tcanopy * c->Rainfall[0]) / (c->delta_t);
  RestTerm = *c->SensibleHeat + *c->LatentHeat + *c->LatentHeatSub + *c->NetRadiation + *c->AdvectedEnergy;
  if (c->IntSnow > 0) {
    *c->RefreezeEnergy = (c->IntRain * ORC_LF * ORC_RHO_W) / (c->delta_t);
    if (Tfoliage == 0.0 && RestTerm > -(*c->RefreezeEnergy)) {
      *c->RefreezeEnergy = -RestTerm;
      RestTerm = 0.0;
    } else RestTerm += *c->RefreezeEnergy;
  } else *c->RefreezeEnergy = 0;
  return RestTerm;
}

/* snow_intercept.c:81-582 (F = 1, mu = 1).  vv->Wdew is IntRain, snow->snow_canopy is IntSnow. */
static int orc_snow_intercept(const orc_model *m, double Dt, double F, double LAI, double Le, double LongOverIn,
                              double LongUnderOut, double MaxInt, double ShortOverIn, double Tcanopy, double bare_albedo,
                              double *AdvectedEnergy, double *AlbedoOver, double *IntRain, double *IntSnow,
                              double *LatentHeat, double *LatentHeatSub, double *LongOverOut, double *MeltEnergy,
                              double *NetLongOver, double *NetShortOver, const orc_vc *Ra, double *ra_used,
                              double *RainFall, double *SensibleHeat, double *SnowFall, double *Tfoliage,
                              int *Tfoliage_fbflag, int *Tfoliage_fbcount, double *TempIntStorage, double *VaporMassFlux,
                              const orc_vc *wind_speed, const orc_vc *displacement, const orc_vc *ref_height,
                              const orc_vc *roughness, const double *root, int month, int hidx, int veg_idx,
                              const orc_atmos *atmos, orc_layer *layer, const orc_soil *sc, orc_vegvar *vv) {
  double BlownSnow, DeltaSnowInt, Drip, ExcessSnowMelt, InitialSnowInt, InitialWaterInt, IntRainOrg, MaxWaterInt,
         MaxSnowInt, NetRadiation, PotSnowMelt, RainThroughFall, RefreezeEnergy = 0, ReleasedMass, SnowThroughFall,
         Imax1, IntRainFract, IntSnowFract, Overload, Qnet, Tupper, Tlower, Evap, OldTfoliage;
  orc_ceb_ctx c;

  *Tfoliage_fbflag = 0;
  *RainFall /= 1000.;
  *SnowFall /= 1000.;
  *IntRain /= 1000.;
  MaxInt /= 1000.;
  IntRainOrg = *IntRain;
  InitialWaterInt = *IntSnow + *IntRain;
  *IntSnow /= F;
  *IntRain /= F;
  InitialSnowInt = *IntSnow;
  Drip = 0.0;
  ReleasedMass = 0.0;
  OldTfoliage = *Tfoliage;
  Imax1 = 4.0 * ORC_LAI_SNOW_MULTIPLIER * LAI;
  if ((*Tfoliage) < -1.0 && (*Tfoliage) > -3.0) MaxSnowInt = ((*Tfoliage) * 3.0 / 2.0) + (11.0 / 2.0);
  else if ((*Tfoliage) > -1.0) MaxSnowInt = 4.0;
  else MaxSnowInt = 1.0;
  MaxSnowInt *= ORC_LAI_SNOW_MULTIPLIER * LAI;
  DeltaSnowInt = (1 - *IntSnow / MaxSnowInt) * *SnowFall;
  if (DeltaSnowInt + *IntSnow > MaxSnowInt) DeltaSnowInt = MaxSnowInt - *IntSnow;
  if (DeltaSnowInt < 0.0) DeltaSnowInt = 0.0;
  if ((*Tfoliage) < -3.0 && DeltaSnowInt > 0.0 && wind_speed->v[ORC_CANOPY] > 1.0) {
    BlownSnow = (0.2 * wind_speed->v[ORC_CANOPY] - 0.2) * DeltaSnowInt;
    if (BlownSnow >= DeltaSnowInt) BlownSnow = DeltaSnowInt;
    DeltaSnowInt -= BlownSnow;
  }
  if (*IntSnow + DeltaSnowInt > Imax1) DeltaSnowInt = 0.0;
  SnowThroughFall = (*SnowFall - DeltaSnowInt) * F + (*SnowFall) * (1 - F);
  if (*SnowFall == 0 && *IntSnow < ORC_MIN_SWQ_EB_THRES) {
    SnowThroughFall += *IntSnow;
    DeltaSnowInt -= *IntSnow;
  }
  *IntSnow += DeltaSnowInt;
  if (*IntSnow < ORC_SMALL) *IntSnow = 0.0;
  MaxWaterInt = ORC_LIQUID_WATER_CAPACITY * (*IntSnow) + MaxInt;
  if ((*IntRain + *RainFall) <= MaxWaterInt) {
    *IntRain += *RainFall;
    RainThroughFall = *RainFall * (1 - F);
  } else {
    RainThroughFall = (*IntRain + *RainFall - MaxWaterInt) * F + (*RainFall * (1 - F));
    *IntRain = MaxWaterInt;
  }
  if (*RainFall == 0 && *IntRain < ORC_MIN_SWQ_EB_THRES) {
    RainThroughFall += *IntRain;
    *IntRain = 0.0;
  }
  if (*IntRain + *IntSnow > Imax1) {
    Overload = (*IntSnow + *IntRain) - Imax1;
    IntRainFract = *IntRain / (*IntRain + *IntSnow);
    IntSnowFract = *IntSnow / (*IntRain + *IntSnow);
    *IntRain = *IntRain - Overload * IntRainFract;
    *IntSnow = *IntSnow - Overload * IntSnowFract;
    RainThroughFall = RainThroughFall + (Overload * IntRainFract) * F;
    SnowThroughFall = SnowThroughFall + (Overload * IntSnowFract) * F;
  }
  if (*IntRain + *IntSnow < ORC_SMALL) *Tfoliage = Tcanopy;

  Tupper = Tlower = NAN;
  c.m = m; c.sc = sc; c.month = month; c.veg_idx = veg_idx; c.delta_t = Dt;
  c.AirDens = atmos->density[hidx]; c.EactAir = atmos->vp[hidx]; c.Press = atmos->pressure[hidx]; c.Le = Le;
  c.Tcanopy = Tcanopy; c.Vpd = atmos->vpd[hidx]; c.Evap = &Evap; c.Ra = Ra; c.ra_used = ra_used; c.Rainfall = RainFall;
  c.wind_speed = wind_speed; c.displacement = displacement; c.ref_height = ref_height; c.roughness = roughness;
  c.root = root; c.IntRain = IntRainOrg; c.layer = layer; c.vv = vv; c.LongOverIn = LongOverIn; c.LongUnderOut = LongUnderOut;
  c.AdvectedEnergy = AdvectedEnergy; c.LatentHeat = LatentHeat; c.LatentHeatSub = LatentHeatSub; c.LongOverOut = LongOverOut;
  c.NetLongOver = NetLongOver; c.NetRadiation = &NetRadiation; c.RefreezeEnergy = &RefreezeEnergy;
  c.SensibleHeat = SensibleHeat; c.VaporMassFlux = VaporMassFlux;

  if (*IntSnow > 0 || *SnowFall > 0) {
    *AlbedoOver = sc->NEW_SNOW_ALB;
    *NetShortOver = (1. - *AlbedoOver) * ShortOverIn;
    c.IntSnow = *IntSnow; c.NetShortOver = *NetShortOver;
    Qnet = orc_canopy_energy_bal(0., &c);
    if (Qnet != 0) {
      Tupper = 0;
      if ((*Tfoliage) <= 0.) Tlower = (*Tfoliage) - ORC_SNOW_DT;
      else Tlower = -ORC_SNOW_DT;
    } else *Tfoliage = 0.;
  } else {
    *AlbedoOver = bare_albedo;
    *NetShortOver = (1. - *AlbedoOver) * ShortOverIn;
    Qnet = NAN;
    Tupper = (*Tfoliage) + ORC_SNOW_DT;
    Tlower = (*Tfoliage) - ORC_SNOW_DT;
  }
  if (!isnan(Tupper) && !isnan(Tlower)) {
    c.IntSnow = *IntSnow; c.NetShortOver = *NetShortOver;
    *Tfoliage = orc_root_brent(Tlower, Tupper, orc_canopy_energy_bal, &c);
    if (orc_is_error(*Tfoliage)) {
      if (m->opt.TFALLBACK) {
        *Tfoliage = OldTfoliage;
        *Tfoliage_fbflag = 1;
        (*Tfoliage_fbcount)++;
      } else return -1;
    }
    c.IntSnow = *IntSnow;
    Qnet = orc_canopy_energy_bal(*Tfoliage, &c);
  }
  if (*IntSnow <= 0) RainThroughFall = vv->throughfall / 1000.;
  RefreezeEnergy *= Dt;
  MaxWaterInt = ORC_LIQUID_WATER_CAPACITY * (*IntSnow) + MaxInt;
  *VaporMassFlux *= Dt;
  if (*Tfoliage == 0) {
    if (-(*VaporMassFlux) > *IntRain) {
      *VaporMassFlux = -(*IntRain);
      *IntRain = 0.;
    } else *IntRain += *VaporMassFlux;
    if (RefreezeEnergy < 0) {
      PotSnowMelt = fmin((-RefreezeEnergy / ORC_LF / ORC_RHO_W), *IntSnow);
      *MeltEnergy -= (ORC_LF * PotSnowMelt * ORC_RHO_W) / (Dt);
    } else {
      PotSnowMelt = 0;
      *MeltEnergy -= (ORC_LF * PotSnowMelt * ORC_RHO_W) / (Dt);
    }
    if ((*IntRain + PotSnowMelt) <= MaxWaterInt) {
      *IntSnow -= PotSnowMelt;
      *IntRain += PotSnowMelt;
      PotSnowMelt = 0.0;
    } else {
      ExcessSnowMelt = PotSnowMelt + *IntRain - MaxWaterInt;
      *IntSnow -= MaxWaterInt - (*IntRain);
      *IntRain = MaxWaterInt;
      if (*IntSnow < 0.0) *IntSnow = 0.0;
      if (SnowThroughFall > 0.0 && InitialSnowInt <= ORC_MIN_INTERCEPTION_STORAGE) {
        Drip += ExcessSnowMelt;
        *IntSnow -= ExcessSnowMelt;
        if (*IntSnow < 0.0) *IntSnow = 0.0;
      } else *TempIntStorage += ExcessSnowMelt;
      orc_mass_release(IntSnow, TempIntStorage, &ReleasedMass, &Drip);
    }
    MaxWaterInt = ORC_LIQUID_WATER_CAPACITY * (*IntSnow) + MaxInt;
    if (*IntRain > MaxWaterInt) {
      Drip += *IntRain - MaxWaterInt;
      *IntRain = MaxWaterInt;
    }
  } else {
    *TempIntStorage = 0.0;
    if (-RefreezeEnergy > -(*IntRain) * ORC_LF) {
      *IntSnow += fabs(RefreezeEnergy) / ORC_LF;
      *IntRain -= fabs(RefreezeEnergy) / ORC_LF;
      *MeltEnergy += (fabs(RefreezeEnergy) * ORC_RHO_W) / (Dt);
      RefreezeEnergy = 0.0;
    } else {
      *IntSnow += *IntRain;
      *MeltEnergy += (ORC_LF * *IntRain * ORC_RHO_W) / (Dt);
      *IntRain = 0.0;
    }
    if (-(*VaporMassFlux) > *IntSnow) {
      *VaporMassFlux = -(*IntSnow);
      *IntSnow = 0.0;
    } else *IntSnow += *VaporMassFlux;
  }
  *IntSnow *= F;
  *IntRain *= F;
  *MeltEnergy *= F;
  *VaporMassFlux *= F;
  Drip *= F;
  ReleasedMass *= F;
  if (*IntSnow == 0 && *IntRain > MaxInt) {
    RainThroughFall += *IntRain - MaxInt;
    *IntRain = MaxInt;
  }
  *RainFall = RainThroughFall + Drip;
  *SnowFall = SnowThroughFall + ReleasedMass;
  *VaporMassFlux *= -1.;
  *RainFall *= 1000.;
  *SnowFall *= 1000.;
  *IntRain *= 1000.;
  *MeltEnergy = RefreezeEnergy / Dt;
  (void)InitialWaterInt; (void)Qnet;
  return 0;
}

/* solve_snow.c:7-544 (mu = 1, SPATIAL_SNOW off).  Returns melt (mm) or ORC_ERROR. */
double orc_solve_snow(const orc_model *m, int overstory, double BareAlbedo, double LongUnderOut, double Tcanopy, double Tgrnd,
                      double air_temp, double prec, double snow_grnd_flux, double *AlbedoUnder, double *Le,
                      double *LongUnderIn, double *NetLongSnow, double *NetShortGrnd, double *NetShortSnow,
                      double *ShortUnderIn, double *Torg_snow, orc_vc *aero_resist, double *ra_used,
                      double *coverage, double *delta_coverage, orc_vc *displacement, double *melt_energy,
                      double *out_prec, double *out_rain, double *out_snow, double *ppt, double *rainfall,
                      orc_vc *ref_height, orc_vc *roughness, double *snow_inflow, double *snowfall, double *surf_atten,
                      orc_vc *wind_speed, const double *root, int UNSTABLE_SNOW, int dt, int hidx, int veg_idx,
                      int is_artificial_bare, int *UnderStory, const orc_dmy *dmy, const orc_atmos *atmos,
                      orc_energy *energy, orc_layer *layer, orc_snow *snow, const orc_soil *sc, orc_vegvar *vv) {
  const double *vl = orc_veg(m, veg_idx);
  const int month = dmy->month, day_in_year = dmy->day_in_year;
  double ShortOverIn, melt = 0., old_coverage, old_swq, rainonly, tmp_grnd_flux, store_snowfall;
  *ppt = 0.;
  *melt_energy = 0.;
  rainonly = orc_calc_rainonly(m, air_temp, prec, sc->MAX_SNOW_TEMP, sc->MIN_RAIN_TEMP);
  *snowfall = atmos->gauge_correction[1] * (prec - rainonly) * sc->PADJ_S;      /* solve_snow.c:159-160 */
  *rainfall = atmos->gauge_correction[0] * rainonly * sc->PADJ_R;
  *out_prec = *snowfall + *rainfall;
  *out_rain = *rainfall;
  *out_snow = *snowfall;
  store_snowfall = *snowfall;
  *Le = (2.501e6 - 0.002361e6 * air_temp);
  if (*UnderStory == ORC_NCASE) {
    if (snow->swq > 0 || *snowfall > 0) *UnderStory = ORC_SNOW_COVERED;
    else *UnderStory = ORC_SNOW_FREE;
  }
  *ShortUnderIn = atmos->shortwave[hidx];
  *LongUnderIn = atmos->longwave[hidx];

  if (snow->swq > 0 || *snowfall > 0. || (snow->snow_canopy > 0. && overstory)) {
    snow->snow = 1;
    if (!overstory) *surf_atten = 1.;
    old_coverage = snow->coverage;
    if (!is_artificial_bare) {
      if (overstory) {
        int err;
        *ShortUnderIn *= *surf_atten;
        ShortOverIn = (1. - *surf_atten) * atmos->shortwave[hidx];
        err = orc_snow_intercept(m, (double)dt * ORC_SECPHOUR, 1., vl[VL_LAI + month - 1], *Le, atmos->longwave[hidx],
                                 LongUnderOut, vl[VL_WDMAX + month - 1], ShortOverIn, Tcanopy, BareAlbedo,
                                 &energy->canopy_advection, &energy->AlbedoOver, &vv->Wdew, &snow->snow_canopy,
                                 &energy->canopy_latent, &energy->canopy_latent_sub, LongUnderIn, &energy->canopy_refreeze,
                                 &energy->NetLongOver, &energy->NetShortOver, aero_resist, ra_used, rainfall,
                                 &energy->canopy_sensible, snowfall, &energy->Tfoliage, &energy->Tfoliage_fbflag,
                                 &energy->Tfoliage_fbcount, &snow->tmp_int_storage, &snow->canopy_vapor_flux, wind_speed,
                                 displacement, ref_height, roughness, root, month, hidx, veg_idx, atmos, layer, sc, vv);
        if (err) return ORC_ERROR;
        vv->throughfall = *rainfall + *snowfall;
        energy->LongOverIn = atmos->longwave[hidx];
      } else if (*snowfall > 0. && vv->Wdew > 0.) {
        *rainfall += vv->Wdew;
        vv->throughfall = *rainfall + *snowfall;
        vv->Wdew = 0.;
        energy->NetLongOver = 0;
        energy->LongOverIn = 0;
        energy->Tfoliage = air_temp;
        energy->Tfoliage_fbflag = 0;
      } else {
        vv->throughfall = *rainfall + *snowfall;
        energy->NetLongOver = 0;
        energy->LongOverIn = 0;
        energy->Tfoliage = air_temp;
        energy->Tfoliage_fbflag = 0;
      }
    } else {
      energy->NetLongOver = 0;
      energy->LongOverIn = 0;
    }
    if (snow->swq > 0.0 || *snowfall > 0) {
      int err;
      *NetShortGrnd = 0.;
      *snow_inflow += *rainfall + *snowfall;
      old_swq = snow->swq;
      *UnderStory = ORC_SNOW_COVERED;
      if (snow->swq > 0 && store_snowfall == 0) {
        snow->last_snow++;
        snow->albedo = orc_snow_albedo(m, *snowfall, snow->swq, snow->depth, snow->albedo, snow->coldcontent, (double)dt,
                                       snow->last_snow, snow->MELTING, sc);
        *AlbedoUnder = (*coverage * snow->albedo + (1. - *coverage) * BareAlbedo);
      } else {
        snow->last_snow = 0;
        snow->albedo = sc->NEW_SNOW_ALB;
        *AlbedoUnder = snow->albedo;
      }
      *NetShortSnow = (1.0 - *AlbedoUnder) * (*ShortUnderIn);
      err = orc_snow_melt(m, *Le, *NetShortSnow, Tcanopy, Tgrnd, roughness->v[ORC_SNOW_COVERED], aero_resist->v[*UnderStory],
                          &ra_used[0], air_temp, (double)dt * ORC_SECPHOUR, atmos->density[hidx], snow_grnd_flux,
                          *LongUnderIn, atmos->pressure[hidx], *rainfall, *snowfall, atmos->vp[hidx], atmos->vpd[hidx],
                          wind_speed->v[*UnderStory], ref_height->v[*UnderStory], NetLongSnow, Torg_snow, &melt,
                          &energy->error, &energy->advected_sensible, &energy->advection, &energy->deltaCC, &tmp_grnd_flux,
                          &energy->latent, &energy->latent_sub, &energy->refreeze_energy, &energy->sensible,
                          UNSTABLE_SNOW, snow);
      if (err) return ORC_ERROR;
      *ppt += melt;
      energy->AlbedoUnder = *AlbedoUnder;
      if (snow->swq > 0.) {
        if (!isnan(snow->surf_temp) && snow->surf_temp <= 0)
          snow->density = orc_snow_density(m, snow, *snowfall, old_swq, air_temp, (double)dt);
        else if (snow->last_snow == 0) snow->density = orc_new_snow_density(m, air_temp);
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
        } else if (old_coverage < snow->coverage) {
          *coverage = snow->coverage;
          *delta_coverage = 0;
        } else {
          *coverage = snow->coverage;
          *delta_coverage = 0.;
        }
      } else if (old_coverage == 0 && snow->coverage == 0) {
        *delta_coverage = 1.;
        *coverage = 0.;
        *melt_energy = (energy->advection - energy->deltaCC + energy->latent + energy->latent_sub + energy->sensible
                        + energy->refreeze_energy + energy->advected_sensible);
      }
      *NetLongSnow *= (snow->coverage);
      *NetShortSnow *= (snow->coverage);
      *NetShortGrnd *= (snow->coverage);
      energy->latent *= (snow->coverage + *delta_coverage);
      energy->latent_sub *= (snow->coverage + *delta_coverage);
      energy->sensible *= (snow->coverage + *delta_coverage);
      if (snow->swq == 0) {
        snow->density = 0.;
        snow->depth = 0.;
        snow->surf_water = 0;
        snow->pack_water = 0;
        snow->surf_temp = 0;
        snow->pack_temp = 0;
        snow->coverage = 0;
        snow->swq_slope = 0;
        snow->store_snow = 1;
        snow->MELTING = 0;
      }
      *snowfall = 0;
      *rainfall = 0;
    } else {
      *ppt += *rainfall;
      energy->AlbedoOver = 0.;
      *AlbedoUnder = BareAlbedo;
      *NetLongSnow = 0.;
      *NetShortSnow = 0.;
      *NetShortGrnd = 0.;
      *delta_coverage = 0.;
      energy->latent = 0.;
      energy->latent_sub = 0.;
      energy->sensible = 0.;
      snow->last_snow = ORC_INVALID_INT;
      snow->store_swq = 0;
      snow->store_coverage = 1;
      snow->MELTING = 0;
    }
  } else {
    *UnderStory = ORC_SNOW_FREE;
    snow->snow = 0;
    energy->Tfoliage = air_temp;
    energy->AlbedoOver = 0.;
    *AlbedoUnder = BareAlbedo;
    energy->NetLongOver = 0.;
    energy->LongOverIn = 0.;
    energy->NetShortOver = 0.;
    energy->ShortOverIn = 0.;
    energy->latent = 0.;
    energy->latent_sub = 0.;
    energy->sensible = 0.;
    *NetLongSnow = 0.;
    *NetShortSnow = 0.;
    *NetShortGrnd = 0.;
    *delta_coverage = 0.;
    energy->Tfoliage = Tcanopy;
    snow->store_swq = 0;
    snow->store_coverage = 1;
    snow->MELTING = 0;
    snow->last_snow = ORC_INVALID_INT;
    snow->albedo = sc->NEW_SNOW_ALB;
  }
  energy->melt_energy *= -1.;
  return melt;
}

/* exported wrappers for orc_glacier.c (solve_snow_glac uses the same snow_utility.c functions) */
double orc_snow_albedo_x(const orc_model *m, double new_snow, double swq, double depth, double albedo, double cold_content,
                         double dt, int last_snow, int MELTING, const orc_soil *sc) {
  return orc_snow_albedo(m, new_snow, swq, depth, albedo, cold_content, dt, last_snow, MELTING, sc);
}
double orc_snow_density_x(const orc_model *m, const orc_snow *snow, double new_snow, double sswq, double Tair, double dt) {
  return orc_snow_density(m, snow, new_snow, sswq, Tair, dt);
}
double orc_new_snow_density_x(const orc_model *m, double air_temp) { return orc_new_snow_density(m, air_temp); }
