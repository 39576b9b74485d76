// vic_glacier.hpp — glacier HRU step: surface_fluxes_glac and below (device only, gfx950).
#pragma once
#include "vic_step.hpp"

namespace vic {

// Residual of the bare-ice surface energy balance (GlacierEnergyBalance.c:15-92, latent_heat_from_glacier.c:8-51)
struct GlacierEB {
  double Dt, Ra, Z, z0_snow, AirDens, EactAir, LongSnowIn, Lv, Press, Rain, NetShortUnder, Vpd, Wind, OldTSurf, IceDepth, Tair, TGrnd;
  double ra_used_surface, AdvectedEnergy, DeltaColdContent, GroundFlux, LatentHeat, LatentHeatSub, NetLongUnder, SensibleHeat,
         vapor_flux;
  VIC_DEV double operator()(double TSurf) {
    const double Density = RHO_W;
    const double temp_IceDepth = IceDepth / 1000.;
    const double TMean = (TSurf + TGrnd) / 2, OldTMean = (OldTSurf + TGrnd) / 2;
    if (Wind > 0.0) ra_used_surface = Ra / stability_correction(Z, 0.f, TSurf, Tair, Wind, z0_snow);
    else ra_used_surface = HUGE_RESIST;
    double Tmp = TSurf + KELVIN;
    NetLongUnder = LongSnowIn - STEFAN_B * Tmp * Tmp * Tmp * Tmp;
    double NetRad = NetShortUnder + NetLongUnder;
    SensibleHeat = AirDens * CP_AIR * (Tair - TSurf) / ra_used_surface;
    double EsSnow = svp(TSurf);
    double VaporMassFlux = AirDens * (EPS_MW / Press) * (EactAir - EsSnow) / ra_used_surface;
    if (Vpd == 0.0 && VaporMassFlux < 0.0) VaporMassFlux = 0.0;
    if (TSurf >= 0.0) { LatentHeat = Lv * VaporMassFlux; LatentHeatSub = 0; }
    else {
      double Ls = (677. - 0.07 * TSurf) * JOULESPCAL * GRAMSPKG;
      LatentHeatSub = Ls * VaporMassFlux;
      LatentHeat = 0;
    }
    vapor_flux = VaporMassFlux * Dt / Density;
    AdvectedEnergy = (TSurf == 0) ? (CH_WATER * (Tair) * Rain) / (Dt) : 0.;
    DeltaColdContent = CH_ICE * temp_IceDepth * (TMean - OldTMean) / (Dt);
    GroundFlux = (GLAC_K_ICE + TSurf * (-0.0142)) * (TGrnd - TSurf) / temp_IceDepth;
    double Fbal = NetRad + SensibleHeat + LatentHeat + LatentHeatSub + AdvectedEnergy;
    double RestTerm = Fbal - DeltaColdContent + GroundFlux;
    if (TSurf == 0.0 && RestTerm >= 0.) RestTerm = 0.;
    return RestTerm;
  }
};

// the energy terms surface_fluxes_glac keeps in its single step_energy copy
struct GlacEnergy {
  double advected_sensible, advection, deltaCC, grnd_flux, latent, latent_sub, refreeze_energy, sensible, snow_flux, error,
         glacier_flux, deltaCC_glac, glacier_melt_energy, AlbedoUnder, LongUnderOut;
};

struct SnowMeltGlacOut { double melt, NetLongSnow, OldTSurf; bool ok; };

// snow_melt_glac (snow_melt_glac.c:14-420): melt stays in m; firn -> ice feeds glacier accumulation
VIC_DEV SnowMeltGlacOut snow_melt_glac(const Opt& o, double Le, double NetShortSnow, double Tgrnd, double z0_snow, double aero_resist,
                                       double& ra_used_surface, double air_temp, double delta_t, double density, double LongSnowIn,
                                       double pressure, double rainfall, double snowfall, double vp, double vpd, double wind, double z2,
                                       Snow& snow, GlacEnergy& ge, double& accumulation) {
  SnowMeltGlacOut out;
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
  double FirnToIce = 0.;
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
  if (PackSwq > 0.0) {                                                    // firn -> ice, snow_melt_glac.c:110-132
    if (snow.density > SNOW_SURF_DENSITY) {
      double zco = (CUTOFF_DENSITY - SNOW_SURF_DENSITY) * (snow.depth / 2) / (snow.density - SNOW_SURF_DENSITY);
      if (zco < snow.depth) {
        double density_zsnow = SNOW_SURF_DENSITY + 2 * (snow.density - SNOW_SURF_DENSITY);
        FirnToIce = (density_zsnow + CUTOFF_DENSITY) / (2 * RHO_W) * (snow.depth - zco);
        if (FirnToIce >= PackSwq) { FirnToIce = PackSwq; PackSwq = 0.0; snow.pack_temp = 0.0; PackCC = 0.0; }
        else PackSwq -= FirnToIce;
      }
    }
    snow.pack_temp = PackCC / (CH_ICE * PackSwq);       // 0/0 = NaN when all firn converted, as in the reference
  } else snow.pack_temp = 0.0;
  accumulation = FirnToIce;
  Ice += SnowFall;
  snow.surf_water += RainFall;

  SnowPackEB eb;
  eb.Dt = delta_t; eb.Ra = aero_resist; eb.Z = z2; eb.z0_snow = z0_snow; eb.AirDens = density; eb.EactAir = vp;
  eb.LongSnowIn = LongSnowIn; eb.Lv = Le; eb.Press = pressure; eb.Rain = RainFall; eb.NetShortUnder = NetShortSnow; eb.Vpd = vpd;
  eb.Wind = wind; eb.OldTSurf = out.OldTSurf; eb.SnowDepth = snow.depth; eb.SnowDensity = snow.density;
  eb.SurfaceLiquidWater = snow.surf_water; eb.SweSurfaceLayer = SurfaceSwq; eb.Tair = air_temp; eb.TGrnd = Tgrnd;
  eb.ra_used_surface = ra_used_surface; eb.vapor_flux = snow.vapor_flux; eb.blowing_flux = snow.blowing_flux;
  eb.surface_flux = snow.surface_flux;
  double Qnet = eb(0.0);
  if (Qnet == 0.0) {
    snow.surf_temp = 0.0;
    double SnowMelt;
    if (eb.RefreezeEnergy >= 0.0) {
      double RefrozenWater = eb.RefreezeEnergy / (LF * RHO_W) * delta_t;
      if (RefrozenWater > snow.surf_water) { RefrozenWater = snow.surf_water; eb.RefreezeEnergy = RefrozenWater * LF * RHO_W / (delta_t); }
      SurfaceSwq += RefrozenWater;
      Ice += RefrozenWater;
      snow.surf_water -= RefrozenWater;
      if (snow.surf_water < 0.0) snow.surf_water = 0.0;
      SnowMelt = 0.0;
    } else SnowMelt = fabs(eb.RefreezeEnergy) / (LF * RHO_W) * delta_t;
    if (snow.surf_water < -(eb.vapor_flux)) {
      eb.blowing_flux *= -(snow.surf_water / eb.vapor_flux);
      eb.vapor_flux = -(snow.surf_water);
      eb.surface_flux = -(snow.surf_water) - eb.blowing_flux;
      snow.surf_water = 0.0;
    } else snow.surf_water += eb.vapor_flux;
    if (SnowMelt < Ice) {
      if (SnowMelt <= PackSwq) { snow.surf_water += SnowMelt; PackSwq -= SnowMelt; Ice -= SnowMelt; }
      else { snow.surf_water += SnowMelt + snow.pack_water; snow.pack_water = 0.0; PackSwq = 0.0; Ice -= SnowMelt; SurfaceSwq = Ice; }
    } else {
      SnowMelt = Ice;
      snow.surf_water += Ice;
      SurfaceSwq = 0.0; snow.surf_temp = 0.0; PackSwq = 0.0; snow.pack_temp = 0.0; Ice = 0.0;
      eb.RefreezeEnergy = eb.RefreezeEnergy / fabs(eb.RefreezeEnergy) * SnowMelt * LF * RHO_W / (delta_t);
    }
  } else {
    snow.surf_temp = root_brent(snow.surf_temp - SNOW_DT, snow.surf_temp + SNOW_DT, eb);
    if (is_error(snow.surf_temp)) {
      if (o.TFALLBACK) { snow.surf_temp = out.OldTSurf; snow.surf_temp_fbflag = 1; snow.surf_temp_fbcount++; }
      else out.ok = false;
    }
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
      } else { SurfaceSwq += eb.vapor_flux; Ice += eb.vapor_flux; }
    }
  }
  double melt;
  double MaxLiquidWater = LIQUID_WATER_CAPACITY * SurfaceSwq;
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
  if (Ice > MAX_SURFACE_SWE) {
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
  } else { PackSwq = 0.0; PackCC = 0.0; snow.pack_temp = 0.0; }
  snow.swq = Ice + snow.pack_water + snow.surf_water;
  if (snow.swq == 0.0) { snow.surf_temp = 0.0; snow.pack_temp = 0.0; }
  snow.mass_error = (InitialSwq - snow.swq) + (RainFall + SnowFall) - melt + eb.vapor_flux;
  out.melt = melt;                                   // stays in m (snow_melt_glac.c:391)
  snow.coldcontent = SurfaceCC;
  snow.vapor_flux = eb.vapor_flux * -1.;
  snow.blowing_flux = eb.blowing_flux;
  snow.surface_flux = eb.surface_flux;
  ra_used_surface = eb.ra_used_surface;
  out.NetLongSnow = eb.NetLongUnder;
  ge.advection = eb.AdvectedEnergy; ge.deltaCC = eb.DeltaColdContent; ge.grnd_flux = eb.GroundFlux; ge.latent = eb.LatentHeat;
  ge.latent_sub = eb.LatentHeatSub; ge.sensible = eb.SensibleHeat; ge.advected_sensible = 0.0; ge.refreeze_energy = eb.RefreezeEnergy;
  ge.error = Qnet;
  return out;
}

template <int NN>
struct GlacWork {
  Glac gl;
  double NetLongUnder_prev;     // energy.NetLongUnder carried from the previous step (surface_fluxes_glac.c:343)
  double deltaH_out, fusion_out;
};

// surface_fluxes_glac (surface_fluxes_glac.c:6-614)
template <int NN>
VIC_DEV bool surface_fluxes_glac(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, const Dmy& dmy,
                                 int veg_idx, int band, double BareAlbedo, const Vc* aero_pet, const Vc& Ra, const Vc& U, const Vc& zref,
                                 const Vc& z0, const Vc& disp, const double* blow /* sigma_slope, lag_one, fetch, is_art_bare */,
                                 HruWork<NN>& w, Glac& gl, double NetLongUnder_prev, GlacEnergy& ge,
                                 double& NetLongUnder_out, double& NetShortUnder_out, double& ShortUnderIn_out) {
  Snow& snow = w.snow;
  bool ok = true;
  int N_steps = 0, UnderStory = SNOW_COVERED;
  double coverage = snow.coverage, delta_coverage = 0;
  double st_AlbedoUnder = 0, st_AtmosLatent = 0, st_AtmosLatentSub = 0, st_AtmosSensible = 0, st_LongUnderIn = 0, st_LongUnderOut = 0,
         st_NetLong = 0, st_NetShort = 0, st_ShortUnderIn = 0, st_advected_sensible = 0, st_advection = 0, st_deltaCC = 0,
         st_grnd_flux = 0, st_latent = 0, st_latent_sub = 0, st_melt_energy = 0, st_refreeze_energy = 0, st_sensible = 0,
         st_snow_flux = 0, st_deltaCC_glac = 0, st_glacier_flux = 0, st_glacier_melt_energy = 0, st_melt_glac = 0,
         st_vapor_flux_glac = 0, st_accum_glac = 0, st_melt = 0, st_vapor_flux = 0, st_blowing_flux = 0, st_surface_flux = 0,
         st_ppt = 0, st_cond_surface = 0, st_cond_overstory = 0;
  double st_pot_evap[NPET] = {0, 0, 0, 0, 0, 0};
  w.out_prec = w.out_rain = w.out_snow = 0;
  w.evap[0] = w.evap[1] = w.evap[2] = 0;
  const double NetLongAtmos_sticky = NetLongUnder_prev;   // step_energy.NetLongUnder is never written on this path

  for (int hidx = 0; hidx < o.NF; hidx++) {
    const int step_dt = o.snow_step;
    const double Tair = fc.v(VIC_F_AIR_TEMP, hidx) + cv.band(CPB_TFACTOR, band);
    const double step_prec = fc.v(VIC_F_PREC, hidx) / 1.0 * cv.band(CPB_PFACTOR, band);
    const double rainOnly = calc_rainonly(o, Tair, step_prec, cv.s(CP_MAX_SNOW_TEMP), cv.s(CP_MIN_RAIN_TEMP));
    double gc[2];
    gauge_correction(o, cv, fc, gc);
    double snowfall = gc[1] * (step_prec - rainOnly) * cv.s(CP_PADJ_S);
    double rainfall = gc[0] * rainOnly * cv.s(CP_PADJ_R);
    const double step_out_prec = snowfall + rainfall, step_out_rain = rainfall, step_out_snow = snowfall;
    const double Tgrnd = GLAC_TEMP, VPDcanopy = 0.;
    if (o.BLOWING && snow.swq > 0.) {                                   // surface_fluxes_glac.c:260-274
      const double Ls = (677. - 0.07 * snow.surf_temp) * 4.1868 * 1000.0;
      const double bf = calc_blowing_snow((double)step_dt, Tair, snow.last_snow, snow.surf_water, U.v[SNOW_COVERED], Ls,
                                          fc.v(VIC_F_DENSITY, hidx), fc.v(VIC_F_VP, hidx), z0.v[SNOW_COVERED], snow.depth, (float)blow[1],
                                          (float)blow[0], blow[3] != 0.0, (float)blow[2], disp.v[CANOPY], z0.v[CANOPY]);
      if ((int)bf == (int)ERROR_VAL) ok = false;
      snow.blowing_flux = bf * step_dt * SECPHOUR / RHO_W;
    } else snow.blowing_flux = 0.0;
    double ra_used[2] = {w.aero_resist_surface, w.aero_resist_overstory};
    snow.canopy_vapor_flux = 0; snow.vapor_flux = 0; snow.surface_flux = 0;
    double LongUnderIn = fc.v(VIC_F_LONGWAVE, hidx), ShortUnderIn = fc.v(VIC_F_SHORTWAVE, hidx);
    const double Le = (2.501e6 - 0.002361e6 * Tair);
    (void)Le;
    double NetLongSnow, NetShortSnow, step_melt, step_melt_glac, step_melt_energy = 0., step_ppt = 0.;
    const double dts = (double)step_dt * SECPHOUR;

    if (snow.swq > 0. || snowfall > 0.) {                               // solve_snow_glac.c:4-290
      snow.snow = 1;
      const double old_coverage = snow.coverage;
      const double old_swq = snow.swq;
      UnderStory = SNOW_COVERED;
      double AlbedoUnder;
      if (snow.swq > 0. && snowfall == 0.) {
        snow.last_snow++;
        snow.albedo = snow_albedo(o, cv, snowfall, snow.swq, snow.depth, snow.albedo, snow.coldcontent, (double)step_dt, snow.last_snow, snow.MELTING);
        AlbedoUnder = (coverage * snow.albedo + (1. - coverage) * BareAlbedo);
      } else {
        snow.last_snow = 0;
        snow.albedo = cv.s(CP_NEW_SNOW_ALB);
        AlbedoUnder = snow.albedo;
      }
      NetShortSnow = (1.0 - AlbedoUnder) * (ShortUnderIn);
      SnowMeltGlacOut sm = snow_melt_glac(o, Le, NetShortSnow, Tgrnd, z0.v[SNOW_COVERED], Ra.v[SNOW_COVERED], ra_used[0], Tair, dts,
                                          fc.v(VIC_F_DENSITY, hidx), LongUnderIn, fc.v(VIC_F_PRESSURE, hidx), rainfall, snowfall,
                                          fc.v(VIC_F_VP, hidx), fc.v(VIC_F_VPD, hidx), U.v[SNOW_COVERED], zref.v[SNOW_COVERED], snow, ge,
                                          gl.accumulation);
      if (!sm.ok) ok = false;
      NetLongSnow = sm.NetLongSnow;
      step_melt = sm.melt;
      step_ppt += step_melt;
      ge.AlbedoUnder = AlbedoUnder;
      if (snow.swq > 0.) {
        if (!isnan(snow.surf_temp) && snow.surf_temp <= 0) snow.density = snow_density(o, snow, snowfall, old_swq, Tair, (double)step_dt);
        else if (snow.last_snow == 0) snow.density = new_snow_density(o, Tair);
        snow.depth = 1000. * snow.swq / snow.density;
        const double lat = cv.s(CP_LAT);
        if (snow.coldcontent >= 0 && ((lat >= 0 && (dmy.day_in_year > 60 && dmy.day_in_year < 273))
                                      || (lat < 0 && (dmy.day_in_year < 60 || dmy.day_in_year > 273))))
          snow.MELTING = 1;
        else if (snow.MELTING && snowfall > TRACESNOW) snow.MELTING = 0;
        snow.coverage = 1.;
      } else snow.coverage = 0.;
      delta_coverage = old_coverage - snow.coverage;
      if (delta_coverage != 0) {
        if (old_coverage > snow.coverage) {
          coverage = old_coverage;
          step_melt_energy = (delta_coverage) * (ge.advection - ge.deltaCC + ge.latent + ge.latent_sub + ge.sensible
                                                 + ge.refreeze_energy + ge.advected_sensible);
        } else { coverage = snow.coverage; delta_coverage = 0; }
      } else if (old_coverage == 0 && snow.coverage == 0) {
        delta_coverage = 1.;
        coverage = 0.;
        step_melt_energy = (ge.advection - ge.deltaCC + ge.latent + ge.latent_sub + ge.sensible + ge.refreeze_energy + ge.advected_sensible);
      }
      const double cf = (snow.coverage + delta_coverage);
      NetLongSnow *= cf; NetShortSnow *= cf;
      ge.latent *= cf; ge.latent_sub *= cf; ge.sensible *= cf;
      if (snow.swq == 0) {
        snow.density = 0.; snow.depth = 0.; snow.surf_water = 0; snow.pack_water = 0; snow.surf_temp = 0; snow.pack_temp = 0;
        snow.coverage = 0; snow.swq_slope = 0; snow.store_snow = 1; snow.MELTING = 0;
      }
      step_melt_glac = 0.;
      gl.vapor_flux = 0.;
      ge.glacier_flux = 0.; ge.deltaCC_glac = 0.; ge.glacier_melt_energy = 0.;
      ge.snow_flux = -ge.grnd_flux;
      ge.LongUnderOut = LongUnderIn - NetLongSnow;
    } else {                                                             // solve_glacier.c:5-104, glacier_melt.c:64-222
      UnderStory = GLACIER_SURF;
      const double AlbedoUnder = BareAlbedo;
      NetShortSnow = (1.0 - AlbedoUnder) * (ShortUnderIn);
      const double RainFall = rainfall / 1000.;
      const double OldTSurf = gl.surf_temp;
      GlacierEB eb;
      eb.Dt = dts; eb.Ra = Ra.v[GLACIER_SURF]; eb.Z = zref.v[GLACIER_SURF]; eb.z0_snow = z0.v[SNOW_COVERED];
      eb.AirDens = fc.v(VIC_F_DENSITY, hidx); eb.EactAir = fc.v(VIC_F_VP, hidx); eb.LongSnowIn = LongUnderIn; eb.Lv = Le;
      eb.Press = fc.v(VIC_F_PRESSURE, hidx); eb.Rain = RainFall; eb.NetShortUnder = NetShortSnow; eb.Vpd = fc.v(VIC_F_VPD, hidx);
      eb.Wind = U.v[GLACIER_SURF]; eb.OldTSurf = OldTSurf; eb.IceDepth = cv.s(CP_GLAC_SURF_THICK); eb.Tair = Tair; eb.TGrnd = Tgrnd;
      eb.ra_used_surface = ra_used[0]; eb.vapor_flux = gl.vapor_flux;
      double Qnet = eb(0.0), melt_energy = 0., GlacMelt = 0, GlacCC = 0;
      if (Qnet == 0.0) {
        gl.surf_temp = 0.;
        melt_energy = NetShortSnow + (eb.NetLongUnder) + eb.SensibleHeat + eb.LatentHeat + eb.LatentHeatSub + eb.AdvectedEnergy - eb.DeltaColdContent;
        GlacMelt = melt_energy / (LF * RHO_W) * dts;
        GlacCC = 0.;
      } else {
        gl.surf_temp = root_brent(gl.surf_temp - SNOW_DT, gl.surf_temp + SNOW_DT, eb);
        if (is_error(gl.surf_temp)) {
          if (o.TFALLBACK) { gl.surf_temp = OldTSurf; gl.surf_temp_fbflag = 1; gl.surf_temp_fbcount++; }
          else ok = false;
        }
        if (!is_error(gl.surf_temp)) {
          Qnet = eb(gl.surf_temp);
          GlacMelt = 0.0;
          GlacCC = CH_ICE * gl.surf_temp * cv.s(CP_GLAC_SURF_THICK) / 1000.;
        }
      }
      gl.cold_content = GlacCC;
      gl.vapor_flux = eb.vapor_flux * -1.;
      ra_used[0] = eb.ra_used_surface;
      ge.advection = eb.AdvectedEnergy; ge.deltaCC_glac = eb.DeltaColdContent; ge.glacier_melt_energy = melt_energy;
      ge.grnd_flux = eb.GroundFlux; ge.latent = eb.LatentHeat; ge.latent_sub = eb.LatentHeatSub; ge.sensible = eb.SensibleHeat;
      ge.error = Qnet;
      NetLongSnow = eb.NetLongUnder;
      step_melt_glac = GlacMelt;
      step_ppt = (GlacMelt + rainfall / 1000.);
      ge.AlbedoUnder = AlbedoUnder;
      rainfall = 0;
      step_melt = 0.;
      ge.deltaCC = 0.; ge.refreeze_energy = 0.; ge.snow_flux = 0.; ge.advected_sensible = 0.;
      ge.glacier_flux = -ge.grnd_flux;
      ge.LongUnderOut = LongUnderIn - NetLongSnow;
      gl.accumulation = 0.;
    }
    const double AtmosLatent = ge.latent, AtmosLatentSub = ge.latent_sub, AtmosSensible = ge.sensible;

    double stability_factor[2], ra_s[NPET], ra_o[NPET], pe[NPET];
    if (ra_used[0] == HUGE_RESIST) stability_factor[0] = HUGE_RESIST;
    else stability_factor[0] = ra_used[0] / vsel(Ra, UnderStory);
    if (ra_used[1] == ra_used[0]) stability_factor[1] = stability_factor[0];
    else if (ra_used[1] == HUGE_RESIST) stability_factor[1] = HUGE_RESIST;
    else stability_factor[1] = ra_used[1] / Ra.v[CANOPY];
#pragma unroll
    for (int p = 0; p < NPET; p++) {
      ra_s[p] = (stability_factor[0] == HUGE_RESIST) ? HUGE_RESIST : vsel(aero_pet[p], UnderStory) * stability_factor[0];
      ra_o[p] = (stability_factor[1] == HUGE_RESIST) ? HUGE_RESIST : aero_pet[p].v[CANOPY] * stability_factor[1];
    }
    compute_pot_evap(o, vl, veg_idx, dmy.month, fc.v(VIC_F_SHORTWAVE, hidx), NetLongAtmos_sticky, Tair, VPDcanopy, cv.s(CP_ELEVATION),
                     ra_s, ra_o, pe);

    st_ppt += step_ppt;
    st_cond_surface += (ra_used[0] > 0) ? 1 / ra_used[0] : HUGE_RESIST;
    st_cond_overstory += (ra_used[1] > 0) ? 1 / ra_used[1] : HUGE_RESIST;
    st_melt += step_melt;
    st_vapor_flux += snow.vapor_flux; st_surface_flux += snow.surface_flux; st_blowing_flux += snow.blowing_flux;
    w.out_prec += step_out_prec * 1.0; w.out_rain += step_out_rain * 1.0; w.out_snow += step_out_snow * 1.0;
    st_AlbedoUnder += ge.AlbedoUnder;
    st_AtmosLatent += AtmosLatent; st_AtmosLatentSub += AtmosLatentSub; st_AtmosSensible += AtmosSensible;
    st_LongUnderIn += LongUnderIn;
    st_LongUnderOut += ge.LongUnderOut;
    st_NetLong += NetLongSnow;
    st_NetShort += NetShortSnow;
    st_ShortUnderIn += ShortUnderIn;
    st_latent += ge.latent; st_latent_sub += ge.latent_sub; st_melt_energy += step_melt_energy; st_sensible += ge.sensible;
    st_grnd_flux += ge.grnd_flux;
    st_melt_glac += step_melt_glac;
    st_vapor_flux_glac += gl.vapor_flux;
    st_accum_glac += gl.accumulation;
    st_glacier_flux += ge.glacier_flux; st_deltaCC_glac += ge.deltaCC_glac; st_glacier_melt_energy += ge.glacier_melt_energy;
    const double cf2 = (snow.coverage + delta_coverage);
    st_advected_sensible += ge.advected_sensible * cf2;
    st_advection += ge.advection * cf2;
    st_deltaCC += ge.deltaCC * cf2;
    st_snow_flux += ge.snow_flux * cf2;
    st_refreeze_energy += ge.refreeze_energy * cf2;
#pragma unroll
    for (int p = 0; p < NPET; p++) st_pot_evap[p] += pe[p];
    N_steps++;
  }

  const double N = (double)N_steps;
  gl.melt = st_melt_glac; gl.vapor_flux = st_vapor_flux_glac; gl.accumulation = st_accum_glac;
  snow.vapor_flux = st_vapor_flux; snow.blowing_flux = st_blowing_flux; snow.surface_flux = st_surface_flux;
  snow.canopy_vapor_flux = 0; snow.melt = st_melt;
  double ppt = st_ppt;
  gl.mass_balance = w.out_prec / 1000. - ppt - snow.vapor_flux - gl.vapor_flux;
  gl.ice_mass_balance = gl.accumulation - gl.melt - gl.vapor_flux;

  ge.AlbedoUnder = st_AlbedoUnder / N;
  w.AtmosLatent = st_AtmosLatent / N; w.AtmosLatentSub = st_AtmosLatentSub / N; w.AtmosSensible = st_AtmosSensible / N;
  w.LongUnderIn = st_LongUnderIn / N;
  ge.LongUnderOut = st_LongUnderOut / N;
  w.NetLongAtmos = st_NetLong / N; NetLongUnder_out = st_NetLong / N;
  w.NetShortAtmos = st_NetShort / N; NetShortUnder_out = st_NetShort / N;
  ShortUnderIn_out = st_ShortUnderIn / N;
  ge.advected_sensible = st_advected_sensible / N;
  ge.grnd_flux = st_grnd_flux / N; ge.latent = st_latent / N; ge.latent_sub = st_latent_sub / N;
  w.so.melt_energy = st_melt_energy / N;
  ge.sensible = st_sensible / N;
  ge.glacier_flux = st_glacier_flux / N; ge.deltaCC_glac = st_deltaCC_glac / N; ge.glacier_melt_energy = st_glacier_melt_energy / N;
  ge.advection = st_advection / N; ge.deltaCC = st_deltaCC / N; ge.refreeze_energy = st_refreeze_energy / N; ge.snow_flux = st_snow_flux / N;
  w.Tcanopy = 0.;
  w.vv.throughfall = 0; w.vv.canopyevap = 0;
  if (st_cond_surface > 0 && st_cond_surface < HUGE_RESIST) w.aero_resist_surface = 1 / (st_cond_surface / N);
  else if (st_cond_surface >= HUGE_RESIST) w.aero_resist_surface = 0;
  else w.aero_resist_surface = HUGE_RESIST;
  if (st_cond_overstory > 0 && st_cond_overstory < HUGE_RESIST) w.aero_resist_overstory = 1 / (st_cond_overstory / N);
  else if (st_cond_overstory >= HUGE_RESIST) w.aero_resist_overstory = 0;
  else w.aero_resist_overstory = HUGE_RESIST;
#pragma unroll
  for (int p = 0; p < NPET; p++) w.pot_evap[p] = st_pot_evap[p] / N;

  // glacier linear reservoir + runoff of the (excess-moisture-only) soil column, surface_fluxes_glac.c:580-601
  gl.inflow = ppt + 0.0;
  ppt = 0.0;                                   // cell.excess_moist is 0 after initialisation
  gl.outflow_coef = cv.s(CP_GLAC_KMIN) + cv.s(CP_GLAC_DK) * exp(-cv.s(CP_GLAC_A) * snow.swq);
  gl.water_storage += gl.inflow;
  gl.outflow = gl.outflow_coef * gl.water_storage;
  gl.water_storage -= gl.outflow;
  w.inflow = ppt;
  RunoffOut ro = runoff_step(o, cv, s3, w.moist, w.ice, w.evap, ppt);
  w.runoff = ro.runoff + (gl.outflow * 1000.);
  w.baseflow = ro.baseflow; w.asat = ro.asat;
  w.zwt = wrap_compute_zwt(cv, s3, w.moist);
  if (o.FULL_ENERGY || o.FROZEN_SOIL) distribute_node_moisture_properties<NN>(o, cv, s3, w.nd, w.moist);
  return ok;
}

}  // namespace vic
