// vic_snow.hpp — snow pack accumulation/ablation, canopy snow interception (device only, gfx950).
#pragma once
#include "vic_soil.hpp"

namespace vic {

// calc_rainonly.c:12-103 (mu == 1)
// correct_precip.c:9-53 as full_energy.c:188-194 applies it: WMO catch-ratio factors for rain and snow from the wind at
// gauge height, 1 when CORRPREC is off or the step is dry.  gc[0] rain, gc[1] snow.
VIC_DEV void gauge_correction(const Opt& o, const CellView& cv, const Forcing& fc, double* gc) {
  gc[0] = 1; gc[1] = 1;
  if (o.CORRPREC && fc.v(VIC_F_PREC, o.NR) > 0) {
    const double GAUGE_HEIGHT = 1.0;
    const double wind = fc.v(VIC_F_WIND, o.NR), rough = cv.s(CP_ROUGH), snow_rough = cv.s(CP_SNOW_ROUGH);
    double gauge_wind = wind * (log((GAUGE_HEIGHT + rough) / rough) / log(o.wind_h / rough));
    gc[0] = 100. / exp(4.606 - 0.041 * pow(gauge_wind, 0.69));
    gauge_wind = wind * (log((GAUGE_HEIGHT + snow_rough) / snow_rough) / log(o.wind_h / snow_rough));
    gc[1] = 100. / exp(4.606 - 0.036 * pow(gauge_wind, 1.75));
  }
}

VIC_DEV double calc_rainonly(const Opt& o, double air_temp, double prec, double MAX_SNOW_TEMP, double MIN_RAIN_TEMP) {
  const double MIN_PREC = 1.e-5;
  double rainonly = 0;
  if (o.TEMP_TH_TYPE == VIC_TEMP_TH_VIC_412) {
    if (air_temp < MAX_SNOW_TEMP && air_temp > MIN_RAIN_TEMP) rainonly = (air_temp - MIN_RAIN_TEMP) / (MAX_SNOW_TEMP - MIN_RAIN_TEMP) * prec;
    else if (air_temp >= MAX_SNOW_TEMP) rainonly = prec;
  } else {
    double TT = MIN_RAIN_TEMP, TR = MAX_SNOW_TEMP, D = 1.4 * TR;
    double E1 = 5. * pow((air_temp - TT) / D, 3.0);
    double E2 = 6.76 * pow((air_temp - TT) / D, 2.0);
    double E3 = 3.19 * (air_temp - TT) / D;
    double rfrac = (air_temp <= TT) ? (E1 + E2 + E3 + 0.5) : (E1 - E2 + E3 + 0.5);
    if (rfrac < 0.) rfrac = 0.;
    if (rfrac > 1.) rfrac = 1.;
    rainonly = rfrac * prec;
  }
  if (rainonly < MIN_PREC) rainonly = 0.;
  if ((prec - rainonly) < MIN_PREC) rainonly = prec;
  return rainonly;
}

// snow_utility.c:199-226
VIC_DEV double new_snow_density(const Opt& o, double air_temp) {
  if (o.SNOW_DENSITY == VIC_DENS_SNTHRM) return 67.9 + 51.3 * exp(air_temp / 2.6);
  air_temp = air_temp * 9. / 5. + 32.;
  if (air_temp > 0) return NEW_SNOW_DENSITY + 1000. * (air_temp / 100.) * (air_temp / 100.);
  return NEW_SNOW_DENSITY;
}

// snow_utility.c:9-196
VIC_DEV double snow_density(const Opt& o, const Snow& snow, double new_snow, double sswq, double Tair, double dt) {
  const double MAX_CHANGE = 0.9;
  double density_new = (new_snow > 0.) ? new_snow_density(o, Tair) : 0.0;
  double Tavg = snow.surf_temp + KELVIN;
  double density;
  if (o.SNOW_DENSITY == VIC_DENS_SNTHRM) {
    if (new_snow > 0.) density = (snow.depth > 0.0) ? snow.density : density_new;
    else density = snow.density;
    double dexpf = exp(-SNDENS_C1 * (KELVIN - Tavg));
    double dm = (new_snow > 0.0 && density_new > 0.0) ? ((SNDENS_DMLIMIT > 1.15 * density_new) ? SNDENS_DMLIMIT : 1.15 * density_new)
                                                       : SNDENS_DMLIMIT;
    double c3 = (density <= dm) ? 1.0 : exp(-0.046 * (density - dm)), c4 = 1.0;
    if ((snow.surf_water + snow.pack_water) / snow.depth > 0.01) c4 = 2.0;
    double ddz1 = -SNDENS_C2 * c3 * c4 * dexpf;
    double swq = new_snow / 1000. + SNDENS_F * sswq;
    double ddz2 = 0.0;
    if (new_snow > 0.0) {
      double Ps = 0.5 * G_GRAV * RHO_W * swq;
      ddz2 = -Ps / SNDENS_ETA0 * exp(-(-SNDENS_C5 * (Tavg - KELVIN) + SNDENS_C6 * density));
    }
    double CR = -ddz1 - ddz2;
    density = density * (1 + CR * dt * SECPHOUR);
  } else {
    double depth = snow.depth, swq = sswq;
    if (new_snow > 0) {
      if (depth > 0.) {
        double delta_depth = (((new_snow / 25.4) * (depth / 0.0254)) / (swq / 0.0254) * pow((depth / 0.0254) / 10., 0.35)) * 0.0254;
        if (delta_depth > MAX_CHANGE * depth) delta_depth = MAX_CHANGE * depth;
        double depth_new = new_snow / density_new;
        depth = depth - delta_depth + depth_new;
        swq += new_snow / 1000.;
        density = 1000. * swq / depth;
      } else {
        density = density_new;
        swq += new_snow / 1000.;
        depth = 1000. * swq / density;
      }
    } else density = 1000. * swq / snow.depth;
    if (depth > 0.) {
      double overburden = 0.5 * G_GRAV * RHO_W * swq;
      double viscosity = SNDENS_ETA0 * exp(-SNDENS_C5 * (Tavg - KELVIN) + SNDENS_C6 * density);
      double delta_depth = overburden / viscosity * depth * dt * SECPHOUR;
      if (delta_depth > MAX_CHANGE * depth) delta_depth = MAX_CHANGE * depth;
      depth -= delta_depth;
      density = 1000. * swq / depth;
    }
  }
  return density;
}

// snow_utility.c:229-307
VIC_DEV double snow_albedo(const Opt& o, const CellView& cv, double new_snow, double swq, double depth, double albedo,
                           double cold_content, double dt, int last_snow, int MELTING) {
  const double nsa = cv.s(CP_NEW_SNOW_ALB);
  if (new_snow > TRACESNOW && cold_content < 0.0) albedo = nsa;
  else if (swq > 0.0) {
    if (o.SNOW_ALBEDO == VIC_SNOW_ALBEDO_SUN1999) {
      if (depth > 0.025) albedo = 0.5 + (albedo - 0.5) * exp(-0.01 * dt / 24);
      else if (cold_content < 0.0) albedo = albedo - 0.006 * dt / 24;
      else albedo = albedo - 0.071 * dt / 24;
      if (albedo < 0) albedo = 0;
    } else {
      if (cold_content < 0.0 && !MELTING)
        albedo = nsa * pow(cv.s(CP_SNOW_ALB_ACCUM_A), pow((double)last_snow * dt / 24., cv.s(CP_SNOW_ALB_ACCUM_B)));
      else
        albedo = nsa * pow(cv.s(CP_SNOW_ALB_THAW_A), pow((double)last_snow * dt / 24., cv.s(CP_SNOW_ALB_THAW_B)));
    }
  } else albedo = 0;
  return albedo;
}

// latent_heat_from_snow.c:8-68
VIC_DEV void latent_heat_from_snow(double AirDens, double EactAir, double Lv, double Press, double Ra, double TMean, double Vpd,
                                   double& LatentHeat, double& LatentHeatSub, double& VaporMassFlux, double BlowingMassFlux,
                                   double& SurfaceMassFlux) {
  double EsSnow = svp(TMean);
  SurfaceMassFlux = AirDens * (EPS_MW / Press) * (EactAir - EsSnow) / Ra;
  if (Vpd == 0.0 && SurfaceMassFlux < 0.0) SurfaceMassFlux = 0.0;
  VaporMassFlux = SurfaceMassFlux + BlowingMassFlux;
  if (TMean >= 0.0) { LatentHeat = Lv * VaporMassFlux; LatentHeatSub = 0; }
  else {
    double Ls = (677. - 0.07 * TMean) * JOULESPCAL * GRAMSPKG;
    LatentHeatSub = Ls * VaporMassFlux;
    LatentHeat = 0;
  }
}

// Residual of the snow-surface energy balance (SnowPackEnergyBalance.c:85-197).  The fluxes it leaves behind are
// members: "the last evaluation wins", as in the reference where they are written through pointers.
struct SnowPackEB {
  // inputs
  double Dt, Ra, Z, z0_snow, AirDens, EactAir, LongSnowIn, Lv, Press, Rain, NetShortUnder, Vpd, Wind, OldTSurf, SnowDepth,
         SnowDensity, SurfaceLiquidWater, SweSurfaceLayer, Tair, TGrnd;
  // in/out and outputs
  double ra_used_surface, AdvectedEnergy, DeltaColdContent, GroundFlux, LatentHeat, LatentHeatSub, NetLongUnder,
         RefreezeEnergy, SensibleHeat, vapor_flux, blowing_flux, surface_flux;

  VIC_DEV double operator()(double TSurf) {
    PROF_WAVE(9); PROF_LANE(10);
    const double TMean = TSurf, Density = RHO_W;
    if (Wind > 0.0) ra_used_surface = Ra / stability_correction(Z, 0.f, TMean, Tair, Wind, z0_snow);
    else ra_used_surface = HUGE_RESIST;
    double Tmp = TMean + KELVIN;
    NetLongUnder = LongSnowIn - STEFAN_B * Tmp * Tmp * Tmp * Tmp;
    double NetRad = NetShortUnder + NetLongUnder;
    SensibleHeat = AirDens * CP_AIR * (Tair - TMean) / ra_used_surface;
    double VaporMassFlux = vapor_flux * Density / Dt;
    double BlowingMassFlux = blowing_flux * Density / Dt;
    double SurfaceMassFlux = surface_flux * Density / Dt;
    latent_heat_from_snow(AirDens, EactAir, Lv, Press, ra_used_surface, TMean, Vpd, LatentHeat, LatentHeatSub, VaporMassFlux,
                          BlowingMassFlux, SurfaceMassFlux);
    vapor_flux = VaporMassFlux * Dt / Density;
    blowing_flux = BlowingMassFlux * Dt / Density;
    surface_flux = SurfaceMassFlux * Dt / Density;
    AdvectedEnergy = (TMean == 0) ? (CH_WATER * (Tair) * Rain) / (Dt) : 0.;
    DeltaColdContent = CH_ICE * SweSurfaceLayer * (TSurf - OldTSurf) / (Dt);
    GroundFlux = (SnowDepth > 0.) ? K_SNOW * SnowDensity * SnowDensity * (TGrnd - TMean) / SnowDepth / (Dt) : 0;
    double RestTerm = NetRad + SensibleHeat + LatentHeat + LatentHeatSub + AdvectedEnergy + 0.0 - DeltaColdContent + GroundFlux;
    RefreezeEnergy = (SurfaceLiquidWater * LF * Density) / (Dt);
    if (TSurf == 0.0 && RestTerm > -(RefreezeEnergy)) {
      RefreezeEnergy = -RestTerm;
      RestTerm = 0.0;
    } else RestTerm += RefreezeEnergy;
    return RestTerm;
  }
};

struct SnowMeltOut { double melt, NetLongSnow, OldTSurf; bool ok; };

// snow_melt (snow_melt.c:119-564).  se receives the snow-side energy terms; ra_used_surface is in/out.
VIC_DEV SnowMeltOut snow_melt(const Opt& o, double Le, double NetShortSnow, double Tcanopy, double Tgrnd, double z0_snow,
                              double aero_resist, double& ra_used_surface, double air_temp, double delta_t, double density,
                              double LongSnowIn, double pressure, double rainfall, double snowfall, double vp, double vpd,
                              double wind, double z2, Snow& snow, SnowEnergy& se) {
  SnowMeltOut out;
  out.ok = true;
  const double SnowFall = snowfall / 1000., RainFall = rainfall / 1000.;
  const double InitialSwq = snow.swq;
  out.OldTSurf = snow.surf_temp;
  double Ice = snow.swq - snow.pack_water - snow.surf_water;
  double SurfaceSwq = (Ice > MAX_SURFACE_SWE) ? MAX_SURFACE_SWE : Ice;
  double PackSwq = Ice - SurfaceSwq;
  double SurfaceCC = CH_ICE * SurfaceSwq * snow.surf_temp;
  double PackCC = CH_ICE * PackSwq * snow.pack_temp;
  double SnowFallCC = (air_temp > 0.0) ? 0.0 : CH_ICE * SnowFall * air_temp;
  double melt_energy = 0.;
  if (SnowFall > (MAX_SURFACE_SWE - SurfaceSwq) && (MAX_SURFACE_SWE - SurfaceSwq) > SMALL) {
    double DeltaPackSwq = SurfaceSwq + SnowFall - MAX_SURFACE_SWE, DeltaPackCC;
    if (DeltaPackSwq > SurfaceSwq) DeltaPackCC = SurfaceCC + (SnowFall - MAX_SURFACE_SWE) / SnowFall * SnowFallCC;
    else DeltaPackCC = DeltaPackSwq / SurfaceSwq * SurfaceCC;
    SurfaceSwq = MAX_SURFACE_SWE;
    SurfaceCC += SnowFallCC - DeltaPackCC;
    PackSwq += DeltaPackSwq;
    PackCC += DeltaPackCC;
  } else {
    SurfaceSwq += SnowFall;
    SurfaceCC += SnowFallCC;
  }
  snow.surf_temp = (SurfaceSwq > 0.0) ? SurfaceCC / (CH_ICE * SurfaceSwq) : 0.0;
  snow.pack_temp = (PackSwq > 0.0) ? PackCC / (CH_ICE * PackSwq) : 0.0;
  Ice += SnowFall;
  snow.surf_water += RainFall;

  SnowPackEB eb;
  eb.Dt = delta_t; eb.Ra = aero_resist; eb.Z = z2; eb.z0_snow = z0_snow; eb.AirDens = density; eb.EactAir = vp;
  eb.LongSnowIn = LongSnowIn; eb.Lv = Le; eb.Press = pressure; eb.Rain = RainFall; eb.NetShortUnder = NetShortSnow; eb.Vpd = vpd;
  eb.Wind = wind; eb.OldTSurf = out.OldTSurf; eb.SnowDepth = snow.depth; eb.SnowDensity = snow.density;
  eb.SurfaceLiquidWater = snow.surf_water; eb.SweSurfaceLayer = SurfaceSwq; eb.Tair = Tcanopy; eb.TGrnd = Tgrnd;
  eb.ra_used_surface = ra_used_surface; eb.vapor_flux = snow.vapor_flux; eb.blowing_flux = snow.blowing_flux;
  eb.surface_flux = snow.surface_flux;

  double Qnet = eb(0.0);

  if (Qnet == 0.0) {                                                     // snow_melt.c:252-319
    snow.surf_temp = 0.0;
    double SnowMelt;
    if (eb.RefreezeEnergy >= 0.0) {
      double RefrozenWater = eb.RefreezeEnergy / (LF * RHO_W) * delta_t;
      if (RefrozenWater > snow.surf_water) {
        RefrozenWater = snow.surf_water;
        eb.RefreezeEnergy = RefrozenWater * LF * RHO_W / (delta_t);
      }
      melt_energy += eb.RefreezeEnergy;
      SurfaceSwq += RefrozenWater;
      Ice += RefrozenWater;
      snow.surf_water -= RefrozenWater;
      if (snow.surf_water < 0.0) snow.surf_water = 0.0;
      SnowMelt = 0.0;
    } else {
      SnowMelt = fabs(eb.RefreezeEnergy) / (LF * RHO_W) * delta_t;
      melt_energy += eb.RefreezeEnergy;
    }
    if (snow.surf_water < -(eb.vapor_flux)) {
      eb.blowing_flux *= -(snow.surf_water / eb.vapor_flux);
      eb.vapor_flux = -(snow.surf_water);
      eb.surface_flux = -(snow.surf_water) - eb.blowing_flux;
      snow.surf_water = 0.0;
    } else snow.surf_water += eb.vapor_flux;
    if (SnowMelt < Ice) {
      if (SnowMelt <= PackSwq) {
        snow.surf_water += SnowMelt;
        PackSwq -= SnowMelt;
        Ice -= SnowMelt;
      } else {
        snow.surf_water += SnowMelt + snow.pack_water;
        snow.pack_water = 0.0;
        PackSwq = 0.0;
        Ice -= SnowMelt;
        SurfaceSwq = Ice;
      }
    } else {
      SnowMelt = Ice;
      snow.surf_water += Ice;
      SurfaceSwq = 0.0;
      snow.surf_temp = 0.0;
      PackSwq = 0.0;
      snow.pack_temp = 0.0;
      Ice = 0.0;
      melt_energy -= eb.RefreezeEnergy;
      eb.RefreezeEnergy = eb.RefreezeEnergy / fabs(eb.RefreezeEnergy) * SnowMelt * LF * RHO_W / (delta_t);
      melt_energy += eb.RefreezeEnergy;
    }
  } else {                                                               // snow_melt.c:322-424
    if (SurfaceSwq > MIN_SWQ_EB_THRES) {
      snow.surf_temp = root_brent(snow.surf_temp - SNOW_DT, snow.surf_temp + SNOW_DT, eb);
      if (is_error(snow.surf_temp)) {
        if (o.TFALLBACK) {
          snow.surf_temp = out.OldTSurf;
          snow.surf_temp_fbflag = 1;
          snow.surf_temp_fbcount++;
        } else out.ok = false;
      }
    } else snow.surf_temp = NAN;       // thin pack: solved together with the ground surface (snow_melt.c:373-375)
    if (!isnan(snow.surf_temp) && !is_error(snow.surf_temp)) {
      Qnet = eb(snow.surf_temp);
      SurfaceSwq += snow.surf_water;
      Ice += snow.surf_water;
      snow.surf_water = 0.0;
      if (SurfaceSwq < -(eb.vapor_flux)) {
        eb.blowing_flux *= -(SurfaceSwq / eb.vapor_flux);
        eb.vapor_flux = -SurfaceSwq;
        eb.surface_flux = -SurfaceSwq - eb.blowing_flux;
        SurfaceSwq = 0.0;
        Ice = PackSwq;
      } else {
        SurfaceSwq += eb.vapor_flux;
        Ice += eb.vapor_flux;
      }
    }
  }

  double melt;
  double MaxLiquidWater = LIQUID_WATER_CAPACITY * SurfaceSwq;           // snow_melt.c:447-505
  if (snow.surf_water > MaxLiquidWater) { melt = snow.surf_water - MaxLiquidWater; snow.surf_water = MaxLiquidWater; }
  else melt = 0.0;
  snow.pack_water += melt;
  double PackRefreezeEnergy = snow.pack_water * LF * RHO_W;
  if (PackCC < -PackRefreezeEnergy) {
    PackSwq += snow.pack_water;
    Ice += snow.pack_water;
    snow.pack_water = 0.0;
    if (PackSwq > 0.0) {
      PackCC = PackSwq * CH_ICE * snow.pack_temp + PackRefreezeEnergy;
      snow.pack_temp = PackCC / (CH_ICE * PackSwq);
      if (snow.pack_temp > 0.) snow.pack_temp = 0.;
    } else snow.pack_temp = 0.0;
  } else {
    snow.pack_temp = 0.0;
    double DeltaPackSwq = -PackCC / (LF * RHO_W);
    snow.pack_water -= DeltaPackSwq;
    PackSwq += DeltaPackSwq;
    Ice += DeltaPackSwq;
  }
  MaxLiquidWater = LIQUID_WATER_CAPACITY * PackSwq;
  if (snow.pack_water > MaxLiquidWater) { melt = snow.pack_water - MaxLiquidWater; snow.pack_water = MaxLiquidWater; }
  else melt = 0.0;
  Ice = PackSwq + SurfaceSwq;
  if (Ice > MAX_SURFACE_SWE) {                                          // re-layer, snow_melt.c:511-533
    SurfaceCC = CH_ICE * snow.surf_temp * SurfaceSwq;
    PackCC = CH_ICE * snow.pack_temp * PackSwq;
    if (SurfaceSwq > MAX_SURFACE_SWE) {
      PackCC += SurfaceCC * (SurfaceSwq - MAX_SURFACE_SWE) / SurfaceSwq;
      SurfaceCC -= SurfaceCC * (SurfaceSwq - MAX_SURFACE_SWE) / SurfaceSwq;
      PackSwq += SurfaceSwq - MAX_SURFACE_SWE;
      SurfaceSwq -= SurfaceSwq - MAX_SURFACE_SWE;
    } else if (SurfaceSwq < MAX_SURFACE_SWE) {
      PackCC -= PackCC * (MAX_SURFACE_SWE - SurfaceSwq) / PackSwq;
      SurfaceCC += PackCC * (MAX_SURFACE_SWE - SurfaceSwq) / PackSwq;
      PackSwq -= MAX_SURFACE_SWE - SurfaceSwq;
      SurfaceSwq += MAX_SURFACE_SWE - SurfaceSwq;
    }
    snow.pack_temp = PackCC / (CH_ICE * PackSwq);
    snow.surf_temp = SurfaceCC / (CH_ICE * SurfaceSwq);
  } else {
    PackSwq = 0.0;
    PackCC = 0.0;
    snow.pack_temp = 0.0;
  }
  snow.swq = Ice + snow.pack_water + snow.surf_water;
  if (snow.swq == 0.0) { snow.surf_temp = 0.0; snow.pack_temp = 0.0; }
  snow.mass_error = (InitialSwq - snow.swq) + (RainFall + SnowFall) - melt + eb.vapor_flux;
  out.melt = melt * 1000.;
  snow.coldcontent = SurfaceCC;
  snow.vapor_flux = eb.vapor_flux * -1.;
  snow.blowing_flux = eb.blowing_flux;
  snow.surface_flux = eb.surface_flux;
  ra_used_surface = eb.ra_used_surface;
  out.NetLongSnow = eb.NetLongUnder;
  se.advection = eb.AdvectedEnergy;
  se.deltaCC = eb.DeltaColdContent;
  se.latent = eb.LatentHeat;
  se.latent_sub = eb.LatentHeatSub;
  se.sensible = eb.SensibleHeat;
  se.advected_sensible = 0.0;
  se.refreeze_energy = eb.RefreezeEnergy;
  se.error = Qnet;
  (void)melt_energy;
  return out;
}

// massrelease.c:40-93 (tail recursion as a loop)
VIC_DEV void mass_release(double& InterceptedSnow, double& TempInterceptionStorage, double& ReleasedMass, double& Drip) {
  for (;;) {
    if (InterceptedSnow > MIN_INTERCEPTION_STORAGE) {
      double Threshold = 0.10 * InterceptedSnow, MaxRelease = 0.17 * InterceptedSnow;
      if (TempInterceptionStorage >= Threshold) {
        Drip += Threshold;
        InterceptedSnow -= Threshold;
        TempInterceptionStorage -= Threshold;
        double rel = (InterceptedSnow < MIN_INTERCEPTION_STORAGE) ? 0.0 : fmin((InterceptedSnow - MIN_INTERCEPTION_STORAGE), MaxRelease);
        ReleasedMass += rel;
        InterceptedSnow -= rel;
        continue;
      }
      double TempDrip = fmin(TempInterceptionStorage, InterceptedSnow);
      Drip += TempDrip;
      InterceptedSnow -= TempDrip;
    } else {
      double TempDrip = fmin(TempInterceptionStorage, InterceptedSnow);
      Drip += TempDrip;
      InterceptedSnow -= TempDrip;
      TempInterceptionStorage = 0.0;
    }
    return;
  }
}

// Residual of the canopy energy balance (func_canopy_energy_bal.c:9-149)
struct CanopyEB {
  // inputs
  int AR;
  VegMonth vm;
  const Soil3* s3;
  const double* moist; const double* ice; const double* root;
  double delta_t, AirDens, EactAir, Press, Le, Tcanopy, Vpd, elevation, Rainfall_m;
  double Ra_snowfree, Ra_canopy, U_canopy, zref_canopy, disp_canopy, z0_canopy;
  double IntRainOrg, IntSnow, LongOverIn, LongUnderOut, NetShortOver;
  // in/out, outputs
  VegVar* vv;                  // vv->Wdew carries IntRain in m while inside snow_intercept
  double* layerevap;           // [3]
  double ra_used[2];
  double Evap, AdvectedEnergy, LatentHeat, LatentHeatSub, LongOverOut, NetLongOver, NetRadiation, RefreezeEnergy, SensibleHeat,
         VaporMassFlux;

  VIC_DEV double operator()(double Tfoliage) {
    PROF_WAVE(11); PROF_LANE(12);
    double Tmp = Tfoliage + KELVIN;
    LongOverOut = STEFAN_B * (Tmp * Tmp * Tmp * Tmp);
    NetRadiation = NetShortOver + LongOverIn + LongUnderOut - 2 * (LongOverOut);
    NetLongOver = LongOverIn - (LongOverOut);
    if (IntSnow > 0) {
      ra_used[0] = Ra_snowfree;
      ra_used[1] = Ra_canopy;
      if (AR == VIC_AR_COMBO || AR == VIC_AR_406 || AR == VIC_AR_406_LS || AR == VIC_AR_406_FULL) ra_used[1] *= 10.;
      double EsSnow = svp(Tfoliage);
      if (AR == VIC_AR_COMBO || AR == VIC_AR_410) {
        if (U_canopy > 0.0) ra_used[1] /= stability_correction(zref_canopy, disp_canopy, Tfoliage, Tcanopy, U_canopy, z0_canopy);
        else ra_used[1] = HUGE_RESIST;
      }
      VaporMassFlux = AirDens * (EPS_MW / Press) * (EactAir - EsSnow) / ra_used[1] / RHO_W;
      if (Vpd == 0.0 && VaporMassFlux < 0.0) VaporMassFlux = 0.0;
      double Ls = (677. - 0.07 * Tfoliage) * JOULESPCAL * GRAMSPKG;
      LatentHeatSub = Ls * VaporMassFlux * RHO_W;
      LatentHeat = 0;
      Evap = 0;
      vv->throughfall = 0;
      if (AR == VIC_AR_406) ra_used[1] /= 10;
    } else {
      ra_used[0] = Ra_snowfree;
      ra_used[1] = (AR == VIC_AR_406_FULL || AR == VIC_AR_410 || AR == VIC_AR_COMBO) ? Ra_canopy : Ra_snowfree;
      Evap = canopy_evap(vm, *s3, moist, ice, *vv, false, IntRainOrg * 1000., delta_t, NetRadiation, Vpd, NetShortOver, Tcanopy,
                         ra_used[1], elevation, Rainfall_m * 1000, root, layerevap);
      vv->Wdew /= 1000.;
      LatentHeat = Le * Evap * RHO_W;
      LatentHeatSub = 0;
    }
    SensibleHeat = AirDens * CP_AIR * (Tcanopy - Tfoliage) / ra_used[1];
    AdvectedEnergy = (4186.8 * Tcanopy * Rainfall_m) / (delta_t);
    double RestTerm = SensibleHeat + LatentHeat + LatentHeatSub + NetRadiation + AdvectedEnergy;
    if (IntSnow > 0) {
      RefreezeEnergy = (IntRainOrg * LF * RHO_W) / (delta_t);
      if (Tfoliage == 0.0 && RestTerm > -(RefreezeEnergy)) {
        RefreezeEnergy = -RestTerm;
        RestTerm = 0.0;
      } else RestTerm += RefreezeEnergy;
    } else RefreezeEnergy = 0;
    return RestTerm;
  }
};

// snow_intercept (snow_intercept.c:81-582), F = 1.  rainfall / snowfall in mm in/out; vv.Wdew (mm) and
// snow.snow_canopy (m) are the intercepted rain / snow.  Returns false when the foliage solve fails with TFALLBACK off.
VIC_DEVN bool snow_intercept(const Opt& o, const CellView& cv, const VegMonth& vm, const Soil3& s3, const Forcing& fc, int hidx,
                             double Dt, double Le, double LongUnderOut, double ShortOverIn, double Tcanopy, double bare_albedo,
                             const Vc& Ra, const Vc& U, const Vc& disp, const Vc& zref, const Vc& z0, double* ra_used,
                             double& rainfall, double& snowfall, double& LongUnderIn, const double* moist, const double* ice,
                             const double* root, double* layerevap, Snow& snow, SnowEnergy& se, VegVar& vv) {
  const double LAI = vm.LAI;
  double RainFall = rainfall / 1000., SnowFall = snowfall / 1000.;
  double IntRain = vv.Wdew / 1000., IntSnow = snow.snow_canopy;
  const double MaxInt = vm.Wdmax / 1000.;
  const double IntRainOrg = IntRain;
  const double InitialSnowInt = IntSnow;
  double Drip = 0.0, ReleasedMass = 0.0;
  double Tfoliage = se.Tfoliage;
  const double OldTfoliage = Tfoliage;
  se.Tfoliage_fbflag = 0;
  const double Imax1 = 4.0 * LAI_SNOW_MULTIPLIER * LAI;
  double MaxSnowInt;
  if (Tfoliage < -1.0 && Tfoliage > -3.0) MaxSnowInt = (Tfoliage * 3.0 / 2.0) + (11.0 / 2.0);
  else if (Tfoliage > -1.0) MaxSnowInt = 4.0;
  else MaxSnowInt = 1.0;
  MaxSnowInt *= LAI_SNOW_MULTIPLIER * LAI;
  double DeltaSnowInt = (1 - IntSnow / MaxSnowInt) * SnowFall;
  if (DeltaSnowInt + IntSnow > MaxSnowInt) DeltaSnowInt = MaxSnowInt - IntSnow;
  if (DeltaSnowInt < 0.0) DeltaSnowInt = 0.0;
  if (Tfoliage < -3.0 && DeltaSnowInt > 0.0 && U.v[CANOPY] > 1.0) {
    double BlownSnow = (0.2 * U.v[CANOPY] - 0.2) * DeltaSnowInt;
    if (BlownSnow >= DeltaSnowInt) BlownSnow = DeltaSnowInt;
    DeltaSnowInt -= BlownSnow;
  }
  if (IntSnow + DeltaSnowInt > Imax1) DeltaSnowInt = 0.0;
  double SnowThroughFall = (SnowFall - DeltaSnowInt) * 1. + (SnowFall) * (1 - 1.);
  if (SnowFall == 0 && IntSnow < MIN_SWQ_EB_THRES) {
    SnowThroughFall += IntSnow;
    DeltaSnowInt -= IntSnow;
  }
  IntSnow += DeltaSnowInt;
  if (IntSnow < SMALL) IntSnow = 0.0;
  double MaxWaterInt = LIQUID_WATER_CAPACITY * (IntSnow) + MaxInt;
  double RainThroughFall;
  if ((IntRain + RainFall) <= MaxWaterInt) {
    IntRain += RainFall;
    RainThroughFall = RainFall * (1 - 1.);
  } else {
    RainThroughFall = (IntRain + RainFall - MaxWaterInt) * 1. + (RainFall * (1 - 1.));
    IntRain = MaxWaterInt;
  }
  if (RainFall == 0 && IntRain < MIN_SWQ_EB_THRES) {
    RainThroughFall += IntRain;
    IntRain = 0.0;
  }
  if (IntRain + IntSnow > Imax1) {
    double Overload = (IntSnow + IntRain) - Imax1;
    double IntRainFract = IntRain / (IntRain + IntSnow);
    double IntSnowFract = IntSnow / (IntRain + IntSnow);
    IntRain = IntRain - Overload * IntRainFract;
    IntSnow = IntSnow - Overload * IntSnowFract;
    RainThroughFall = RainThroughFall + (Overload * IntRainFract) * 1.;
    SnowThroughFall = SnowThroughFall + (Overload * IntSnowFract) * 1.;
  }
  if (IntRain + IntSnow < SMALL) Tfoliage = Tcanopy;

  CanopyEB eb;
  eb.AR = o.AERO_RESIST_CANSNOW; eb.vm = vm; eb.s3 = &s3; eb.moist = moist; eb.ice = ice; eb.root = root; eb.delta_t = Dt;
  eb.AirDens = fc.v(VIC_F_DENSITY, hidx); eb.EactAir = fc.v(VIC_F_VP, hidx); eb.Press = fc.v(VIC_F_PRESSURE, hidx);
  eb.Le = Le; eb.Tcanopy = Tcanopy; eb.Vpd = fc.v(VIC_F_VPD, hidx); eb.elevation = cv.s(CP_ELEVATION); eb.Rainfall_m = RainFall;
  eb.Ra_snowfree = Ra.v[SNOW_FREE]; eb.Ra_canopy = Ra.v[CANOPY]; eb.U_canopy = U.v[CANOPY]; eb.zref_canopy = zref.v[CANOPY];
  eb.disp_canopy = disp.v[CANOPY]; eb.z0_canopy = z0.v[CANOPY];
  eb.IntRainOrg = IntRainOrg; eb.IntSnow = IntSnow; eb.LongOverIn = fc.v(VIC_F_LONGWAVE, hidx); eb.LongUnderOut = LongUnderOut;
  eb.vv = &vv; eb.layerevap = layerevap; eb.ra_used[0] = ra_used[0]; eb.ra_used[1] = ra_used[1];
  eb.RefreezeEnergy = 0; eb.VaporMassFlux = snow.canopy_vapor_flux; eb.Evap = 0;
  eb.AdvectedEnergy = se.canopy_advection; eb.LatentHeat = se.canopy_latent; eb.LatentHeatSub = se.canopy_latent_sub;
  eb.LongOverOut = LongUnderIn; eb.NetLongOver = se.NetLongOver; eb.SensibleHeat = se.canopy_sensible; eb.NetRadiation = 0;
  // while inside this routine the reference keeps IntRain (m) in veg_var_wet->Wdew
  vv.Wdew = IntRain;

  double Tupper = NAN, Tlower = NAN;
  if (IntSnow > 0 || SnowFall > 0) {
    se.AlbedoOver = cv.s(CP_NEW_SNOW_ALB);
    se.NetShortOver = (1. - se.AlbedoOver) * ShortOverIn;
    eb.NetShortOver = se.NetShortOver;
    double Qnet = eb(0.);
    if (Qnet != 0) {
      Tupper = 0;
      Tlower = (Tfoliage <= 0.) ? Tfoliage - SNOW_DT : -SNOW_DT;
    } else Tfoliage = 0.;
  } else {
    se.AlbedoOver = bare_albedo;
    se.NetShortOver = (1. - se.AlbedoOver) * ShortOverIn;
    eb.NetShortOver = se.NetShortOver;
    Tupper = Tfoliage + SNOW_DT;
    Tlower = Tfoliage - SNOW_DT;
  }
  bool ok = true;
  if (!isnan(Tupper) && !isnan(Tlower)) {
    Tfoliage = root_brent(Tlower, Tupper, eb);
    if (is_error(Tfoliage)) {
      if (o.TFALLBACK) {
        Tfoliage = OldTfoliage;
        se.Tfoliage_fbflag = 1;
        se.Tfoliage_fbcount++;
      } else ok = false;
    }
    (void)eb(Tfoliage);
  }
  IntRain = vv.Wdew;      // the no-snow residual leaves the post-evaporation storage here (func_canopy_energy_bal.c:97-106)
  if (IntSnow <= 0) RainThroughFall = vv.throughfall / 1000.;
  double RefreezeEnergy = eb.RefreezeEnergy * Dt;
  MaxWaterInt = LIQUID_WATER_CAPACITY * (IntSnow) + MaxInt;
  double VaporMassFlux = eb.VaporMassFlux * Dt;
  double TempIntStorage = snow.tmp_int_storage;
  if (Tfoliage == 0) {
    if (-(VaporMassFlux) > IntRain) { VaporMassFlux = -(IntRain); IntRain = 0.; }
    else IntRain += VaporMassFlux;
    double PotSnowMelt = (RefreezeEnergy < 0) ? fmin((-RefreezeEnergy / LF / RHO_W), IntSnow) : 0;
    if ((IntRain + PotSnowMelt) <= MaxWaterInt) {
      IntSnow -= PotSnowMelt;
      IntRain += PotSnowMelt;
    } else {
      double ExcessSnowMelt = PotSnowMelt + IntRain - MaxWaterInt;
      IntSnow -= MaxWaterInt - (IntRain);
      IntRain = MaxWaterInt;
      if (IntSnow < 0.0) IntSnow = 0.0;
      if (SnowThroughFall > 0.0 && InitialSnowInt <= MIN_INTERCEPTION_STORAGE) {
        Drip += ExcessSnowMelt;
        IntSnow -= ExcessSnowMelt;
        if (IntSnow < 0.0) IntSnow = 0.0;
      } else TempIntStorage += ExcessSnowMelt;
      mass_release(IntSnow, TempIntStorage, ReleasedMass, Drip);
    }
    MaxWaterInt = LIQUID_WATER_CAPACITY * (IntSnow) + MaxInt;
    if (IntRain > MaxWaterInt) { Drip += IntRain - MaxWaterInt; IntRain = MaxWaterInt; }
  } else {
    TempIntStorage = 0.0;
    if (-RefreezeEnergy > -(IntRain)*LF) {
      IntSnow += fabs(RefreezeEnergy) / LF;
      IntRain -= fabs(RefreezeEnergy) / LF;
      RefreezeEnergy = 0.0;
    } else {
      IntSnow += IntRain;
      IntRain = 0.0;
    }
    if (-(VaporMassFlux) > IntSnow) { VaporMassFlux = -(IntSnow); IntSnow = 0.0; }
    else IntSnow += VaporMassFlux;
  }
  if (IntSnow == 0 && IntRain > MaxInt) {
    RainThroughFall += IntRain - MaxInt;
    IntRain = MaxInt;
  }
  rainfall = (RainThroughFall + Drip) * 1000.;
  snowfall = (SnowThroughFall + ReleasedMass) * 1000.;
  vv.Wdew = IntRain * 1000.;
  snow.snow_canopy = IntSnow;
  snow.tmp_int_storage = TempIntStorage;
  snow.canopy_vapor_flux = VaporMassFlux * -1.;
  se.Tfoliage = Tfoliage;
  se.canopy_advection = eb.AdvectedEnergy;
  se.canopy_latent = eb.LatentHeat;
  se.canopy_latent_sub = eb.LatentHeatSub;
  se.canopy_refreeze = RefreezeEnergy / Dt;          // snow_intercept.c:578
  se.NetLongOver = eb.NetLongOver;
  se.canopy_sensible = eb.SensibleHeat;
  LongUnderIn = eb.LongOverOut;
  ra_used[0] = eb.ra_used[0];
  ra_used[1] = eb.ra_used[1];
  return ok;
}

}  // namespace vic
