// vic_step.hpp — one HRU step: full_energy's per-HRU body and surface_fluxes (device only, gfx950).
#pragma once
#include "vic_surface.hpp"
#include "vic_blowing.hpp"

namespace vic {

struct SolveSnowOut {
  double melt, Le, LongUnderIn, NetLongSnow, NetShortGrnd, NetShortSnow, ShortUnderIn, OldTSurf, delta_coverage, melt_energy,
         out_prec, out_rain, out_snow, ppt, rainfall, snowfall;
  int ok, pad_;
};

// solve_snow (solve_snow.c:7-544), mu = 1, SPATIAL_SNOW off.  `coverage` and `surf_atten` persist across sub-steps in
// the caller, UnderStory is in/out (NCASE = not yet decided).
VIC_DEV SolveSnowOut solve_snow(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, int hidx,
                                int veg_idx, const Dmy& dmy, bool overstory, bool is_artificial_bare, double BareAlbedo,
                                double LongUnderOut, double Tcanopy, double Tgrnd, double air_temp, double prec, double& AlbedoUnder,
                                const Vc& Ra, const Vc& U, const Vc& disp, const Vc& zref, const Vc& z0, double* ra_used,
                                double& coverage, double& surf_atten, double& snow_inflow, int& UnderStory, int dt,
                                const double* lmoist, const double* lice, const double* root, double* layerevap, Snow& snow,
                                SnowEnergy& se, VegVar& vv) {
  SolveSnowOut r;
  r.ok = true; r.melt = 0.; r.ppt = 0.; r.melt_energy = 0.; r.OldTSurf = 0.; r.delta_coverage = 0.;
  r.NetLongSnow = 0.; r.NetShortGrnd = 0.; r.NetShortSnow = 0.;
  const int month = dmy.month;
  const double rainonly = calc_rainonly(o, air_temp, prec, cv.s(CP_MAX_SNOW_TEMP), cv.s(CP_MIN_RAIN_TEMP));
  double gc[2];
  gauge_correction(o, cv, fc, gc);
  double snowfall = gc[1] * (prec - rainonly) * cv.s(CP_PADJ_S);
  double rainfall = gc[0] * rainonly * cv.s(CP_PADJ_R);
  r.out_prec = snowfall + rainfall; r.out_rain = rainfall; r.out_snow = snowfall;
  const double store_snowfall = snowfall;
  r.Le = (2.501e6 - 0.002361e6 * air_temp);
  if (UnderStory == NCASE) UnderStory = (snow.swq > 0 || snowfall > 0) ? SNOW_COVERED : SNOW_FREE;
  r.ShortUnderIn = fc.v(VIC_F_SHORTWAVE, hidx);
  r.LongUnderIn = fc.v(VIC_F_LONGWAVE, hidx);

  if (snow.swq > 0 || snowfall > 0. || (snow.snow_canopy > 0. && overstory)) {
    snow.snow = 1;
    if (!overstory) surf_atten = 1.;
    const double old_coverage = snow.coverage;
    if (!is_artificial_bare) {
      if (overstory) {
        const VegMonth vm = veg_month(vl, veg_idx, month);
        r.ShortUnderIn *= surf_atten;
        const double ShortOverIn = (1. - surf_atten) * fc.v(VIC_F_SHORTWAVE, hidx);
        PROF_T0(t_si);
        if (!snow_intercept(o, cv, vm, s3, fc, hidx, (double)dt * SECPHOUR, r.Le, LongUnderOut, ShortOverIn, Tcanopy, BareAlbedo,
                            Ra, U, disp, zref, z0, ra_used, rainfall, snowfall, r.LongUnderIn, lmoist, lice, root, layerevap, snow,
                            se, vv))
          r.ok = false;
        PROF_ADD(9, t_si);
        vv.throughfall = rainfall + snowfall;
        se.LongOverIn = fc.v(VIC_F_LONGWAVE, hidx);
      } else if (snowfall > 0. && vv.Wdew > 0.) {
        rainfall += vv.Wdew;
        vv.throughfall = rainfall + snowfall;
        vv.Wdew = 0.;
        se.NetLongOver = 0; se.LongOverIn = 0; se.Tfoliage = air_temp; se.Tfoliage_fbflag = 0;
      } else {
        vv.throughfall = rainfall + snowfall;
        se.NetLongOver = 0; se.LongOverIn = 0; se.Tfoliage = air_temp; se.Tfoliage_fbflag = 0;
      }
    } else { se.NetLongOver = 0; se.LongOverIn = 0; }

    if (snow.swq > 0.0 || snowfall > 0) {
      r.NetShortGrnd = 0.;
      snow_inflow += rainfall + snowfall;
      const double old_swq = snow.swq;
      UnderStory = SNOW_COVERED;
      if (snow.swq > 0 && store_snowfall == 0) {
        snow.last_snow++;
        snow.albedo = snow_albedo(o, cv, snowfall, snow.swq, snow.depth, snow.albedo, snow.coldcontent, (double)dt, snow.last_snow, snow.MELTING);
        AlbedoUnder = (coverage * snow.albedo + (1. - coverage) * BareAlbedo);
      } else {
        snow.last_snow = 0;
        snow.albedo = cv.s(CP_NEW_SNOW_ALB);
        AlbedoUnder = snow.albedo;
      }
      r.NetShortSnow = (1.0 - AlbedoUnder) * (r.ShortUnderIn);
      PROF_T0(t_sm);
      SnowMeltOut sm = snow_melt(o, r.Le, r.NetShortSnow, Tcanopy, Tgrnd, z0.v[SNOW_COVERED], Ra.v[SNOW_COVERED], ra_used[0], air_temp,
                                 (double)dt * SECPHOUR, fc.v(VIC_F_DENSITY, hidx), r.LongUnderIn, fc.v(VIC_F_PRESSURE, hidx), rainfall,
                                 snowfall, fc.v(VIC_F_VP, hidx), fc.v(VIC_F_VPD, hidx), U.v[SNOW_COVERED], zref.v[SNOW_COVERED], snow, se);
      PROF_ADD(10, t_sm);
      if (!sm.ok) r.ok = false;
      r.melt = sm.melt; r.NetLongSnow = sm.NetLongSnow; r.OldTSurf = sm.OldTSurf;
      r.ppt += r.melt;
      if (snow.swq > 0.) {
        if (!isnan(snow.surf_temp) && snow.surf_temp <= 0) snow.density = snow_density(o, snow, snowfall, old_swq, air_temp, (double)dt);
        else if (snow.last_snow == 0) snow.density = new_snow_density(o, air_temp);
        snow.depth = 1000. * snow.swq / snow.density;
        const double lat = cv.s(CP_LAT);
        if (snow.coldcontent >= 0 && ((lat >= 0 && (dmy.day_in_year > 60 && dmy.day_in_year < 273))
                                      || (lat < 0 && (dmy.day_in_year < 60 || dmy.day_in_year > 273))))
          snow.MELTING = 1;
        else if (snow.MELTING && snowfall > TRACESNOW) snow.MELTING = 0;
        snow.coverage = 1.;
      } else snow.coverage = 0.;
      r.delta_coverage = old_coverage - snow.coverage;
      if (r.delta_coverage != 0) {
        if (old_coverage > snow.coverage) {
          coverage = old_coverage;
          AlbedoUnder = (coverage - snow.coverage) / (1. - snow.coverage) * snow.albedo;
          AlbedoUnder += (1. - coverage) / (1. - snow.coverage) * BareAlbedo;
          r.melt_energy = (r.delta_coverage) * (se.advection - se.deltaCC + se.latent + se.latent_sub + se.sensible
                                                + se.refreeze_energy + se.advected_sensible);
        } else {
          coverage = snow.coverage;
          r.delta_coverage = 0;
        }
      } else if (old_coverage == 0 && snow.coverage == 0) {
        r.delta_coverage = 1.;
        coverage = 0.;
        r.melt_energy = (se.advection - se.deltaCC + se.latent + se.latent_sub + se.sensible + se.refreeze_energy + se.advected_sensible);
      }
      r.NetLongSnow *= (snow.coverage);
      r.NetShortSnow *= (snow.coverage);
      r.NetShortGrnd *= (snow.coverage);
      se.latent *= (snow.coverage + r.delta_coverage);
      se.latent_sub *= (snow.coverage + r.delta_coverage);
      se.sensible *= (snow.coverage + r.delta_coverage);
      if (snow.swq == 0) {
        snow.density = 0.; snow.depth = 0.; snow.surf_water = 0; snow.pack_water = 0; snow.surf_temp = 0; snow.pack_temp = 0;
        snow.coverage = 0; snow.swq_slope = 0; snow.store_snow = 1; snow.MELTING = 0;
      }
      snowfall = 0;
      rainfall = 0;
    } else {
      r.ppt += rainfall;
      se.AlbedoOver = 0.;
      AlbedoUnder = BareAlbedo;
      r.NetLongSnow = 0.; r.NetShortSnow = 0.; r.NetShortGrnd = 0.; r.delta_coverage = 0.;
      se.latent = 0.; se.latent_sub = 0.; se.sensible = 0.;
      snow.last_snow = INVALID_INT;
      snow.store_swq = 0; snow.store_coverage = 1; snow.MELTING = 0;
    }
  } else {
    UnderStory = SNOW_FREE;
    snow.snow = 0;
    se.AlbedoOver = 0.;
    AlbedoUnder = BareAlbedo;
    se.NetLongOver = 0.; se.LongOverIn = 0.; se.NetShortOver = 0.; se.ShortOverIn = 0.;
    se.latent = 0.; se.latent_sub = 0.; se.sensible = 0.;
    r.NetLongSnow = 0.; r.NetShortSnow = 0.; r.NetShortGrnd = 0.; r.delta_coverage = 0.;
    se.Tfoliage = Tcanopy;
    snow.store_swq = 0; snow.store_coverage = 1; snow.MELTING = 0; snow.last_snow = INVALID_INT;
    snow.albedo = cv.s(CP_NEW_SNOW_ALB);
  }
  r.rainfall = rainfall; r.snowfall = snowfall;
  return r;
}

// Everything one HRU carries through a step (loaded from / stored to the SoA state tables by the kernel)
template <int NN>
struct HruWork {
  double moist[3], ice[3], layer_T[3], evap[3];
  Nodes<NN> nd;
  Snow snow;
  VegVar vv;
  SnowEnergy se;
  SoilEnergy so;
  // sticky diagnostics that belong to neither side exclusively
  double Tcanopy;
  // per-step outputs
  double runoff, baseflow, asat, inflow, pot_evap[NPET], aero_resist_surface, aero_resist_overstory, rootmoist, wetness;
  Zwt zwt;
  double AtmosLatent, AtmosLatentSub, AtmosSensible, LongUnderIn, NetLongAtmos, NetShortAtmos, ShortUnderIn_avg, AlbedoOver_avg,
         AlbedoUnder_avg, LongOverIn_avg, NetLongOver_avg, NetShortOver_avg, ShortOverIn_avg;
  double out_prec, out_rain, out_snow;
  // glacier HRUs only
  Glac gl;
  double deltaCC_glac, glacier_flux, glacier_melt_energy;
};

// What of HruWork has to survive from the set-up of a sub-step (before the ground-surface root finder) to its
// bookkeeping (after it): what the set-up changed and the bookkeeping reads.  Everything else is either assigned by the
// bookkeeping before it reads it (surf_post assigns the whole soil-side energy balance, layer temperatures and ice, node
// temperatures come back from the profile solve, node moisture / ice / conductivity from distribute_node_moisture_properties,
// the per-step outputs from sf_end) or is state the step has not touched yet and is read from the state table again
// (load_untouched_state): layer moisture and ice, Wdew, last step's advected_sensible, the fallback counters.
struct WCarry {
  Snow snow;
  SnowEnergy se;
  double so_advection, aero_resist_surface, aero_resist_overstory, out_prec, out_rain, out_snow;
};
// ... and, in runs with more than one snow sub-step per step, what the sub-steps before this one have changed (layer ice and
// temperature, the fallback counters) and the step's top-layer thermal properties, which the next sub-step's set-up reads
template <int NN>
struct WCarryMulti {
  double ice[3], layer_T[3], kappa[2], Cs[2];
  int fbcount[NN + (NN & 1)];
  int Tsurf_fbcount, pad_;
};

template <int NN>
VIC_DEV void carry_out(const HruWork<NN>& w, WCarry& k) {
  k.snow = w.snow; k.se = w.se; k.so_advection = w.so.advection;
  k.aero_resist_surface = w.aero_resist_surface; k.aero_resist_overstory = w.aero_resist_overstory;
  k.out_prec = w.out_prec; k.out_rain = w.out_rain; k.out_snow = w.out_snow;
}
template <int NN>
VIC_DEV void carry_in(const WCarry& k, HruWork<NN>& w) {
  w.snow = k.snow; w.se = k.se; w.so.advection = k.so_advection;
  w.aero_resist_surface = k.aero_resist_surface; w.aero_resist_overstory = k.aero_resist_overstory;
  w.out_prec = k.out_prec; w.out_rain = k.out_rain; w.out_snow = k.out_snow;
}
template <int NN>
VIC_DEV void carry_out_multi(const HruWork<NN>& w, WCarryMulti<NN>& k) {
#pragma unroll
  for (int l = 0; l < 3; l++) { k.ice[l] = w.ice[l]; k.layer_T[l] = w.layer_T[l]; }
  k.kappa[0] = w.so.kappa[0]; k.kappa[1] = w.so.kappa[1]; k.Cs[0] = w.so.Cs[0]; k.Cs[1] = w.so.Cs[1];
#pragma unroll
  for (int n = 0; n < NN + (NN & 1); n++) k.fbcount[n] = (n < NN) ? w.nd.fbcount[n] : 0;
  k.Tsurf_fbcount = w.so.Tsurf_fbcount; k.pad_ = 0;
}
template <int NN>
VIC_DEV void carry_in_multi(const WCarryMulti<NN>& k, HruWork<NN>& w) {
#pragma unroll
  for (int l = 0; l < 3; l++) { w.ice[l] = k.ice[l]; w.layer_T[l] = k.layer_T[l]; }
  w.so.kappa[0] = k.kappa[0]; w.so.kappa[1] = k.kappa[1]; w.so.Cs[0] = k.Cs[0]; w.so.Cs[1] = k.Cs[1];
#pragma unroll
  for (int n = 0; n < NN; n++) w.nd.fbcount[n] = k.fbcount[n];
  w.so.Tsurf_fbcount = k.Tsurf_fbcount;
}

// the sub-step sums of SubLoop (all zero until the first sub-step has been booked)
// Per-step constants of one HRU: the prologue of full_energy's HRU loop (full_energy.c:216-354)
struct StepConst {
  Vc aero_pet[NPET];     // aero_pet[p].v = { snowFree, canopy, snowCovered, - } resistances of PET type p
  Vc Ra, U, disp, zref, z0;
  double surf_atten, bare_albedo, ice0, moist0, root[3];
  double sigma_slope, lag_one, fetch;      // veg_con (float values): blowing snow only
  int veg_idx, band, is_art_bare, overstory;
};

// The part of StepConst the bookkeeping after the root find reads (potential evaporation: the resistances of the six PET
// surfaces under the sub-step's understory case and in the canopy): what a run with one sub-step per step parks instead of
// the whole of StepConst (the rest comes from the HRU tables again)
struct StepConstPost { double pet_under[NPET], pet_canopy[NPET], Ra_under, Ra_canopy; };
VIC_DEV void step_const_post_out(const StepConst& C, int UnderStory, StepConstPost& q) {
#pragma unroll
  for (int p = 0; p < NPET; p++) { q.pet_under[p] = vsel(C.aero_pet[p], UnderStory); q.pet_canopy[p] = C.aero_pet[p].v[CANOPY]; }
  q.Ra_under = vsel(C.Ra, UnderStory); q.Ra_canopy = C.Ra.v[CANOPY];
}
VIC_DEV void step_const_post_in(const StepConstPost& q, int UnderStory, StepConst& C) {
#pragma unroll
  for (int k = 0; k < NCASE; k++) {
#pragma unroll
    for (int p = 0; p < NPET; p++) C.aero_pet[p].v[k] = (k == CANOPY) ? q.pet_canopy[p] : ((k == UnderStory) ? q.pet_under[p] : 0.0);
    C.Ra.v[k] = (k == CANOPY) ? q.Ra_canopy : ((k == UnderStory) ? q.Ra_under : 0.0);
    C.U.v[k] = 0; C.disp.v[k] = 0; C.zref.v[k] = 0; C.z0.v[k] = 0;
  }
  C.surf_atten = 0; C.bare_albedo = 0; C.ice0 = 0; C.moist0 = 0; C.sigma_slope = 0; C.lag_one = 0; C.fetch = 0;
}

// surface_fluxes (surface_fluxes.c:17-956) with CLOSE_ENERGY FALSE (both closure loops execute once), Ndist 1.
// The reference's iter_* / step_* struct copies collapse to in-place updates of the snow side (se, snow, vv_snow) and
// the soil side (so, nodes, layers, vv_soil); both sides start from last step's values as surface_fluxes.c:301-323 does.
// The function is cut at the ground-surface root finder: sf_begin | { sf_sub_pre | solve | sf_sub_post }* | sf_end,
// with everything that lives across the cut in SubLoop (whole step) and SubStep (one snow sub-step).
struct SubLoop {
  double snow_flux, coverage, last_snow_coverage, Tgrnd0, step_Wdew, AlbedoUnder_orig, snow_inflow, delta_coverage, surf_atten;
  VegVar vv_snow, vv_soil;
  int hidx, endhidx, step_dt, INCLUDE_SNOW, N_steps, ok;
  double st_AlbedoOver, st_AlbedoUnder, st_AtmosLatent, st_AtmosLatentSub, st_AtmosSensible, st_LongOverIn,
         st_LongUnderIn, st_LongUnderOut, st_NetLongAtmos, st_NetLongOver, st_NetLongUnder, st_NetShortAtmos,
         st_NetShortGrnd, st_NetShortOver, st_NetShortUnder, st_ShortOverIn, st_ShortUnderIn,
         st_advected_sensible, st_advection, st_canopy_advection, st_canopy_latent, st_canopy_latent_sub,
         st_canopy_sensible, st_canopy_refreeze, st_deltaCC, st_deltaH, st_fusion, st_grnd_flux,
         st_latent, st_latent_sub, st_melt_energy, st_refreeze_energy, st_sensible, st_snow_flux,
         st_canopy_vapor_flux, st_melt, st_vapor_flux, st_blowing_flux, st_surface_flux, st_canopyevap,
         st_throughfall, st_ppt, st_cond_surface, st_cond_overstory;
  double st_layerevap[3], st_pot_evap[NPET];
};

VIC_DEV void zero_substep_sums(SubLoop& L) {
  L.st_AlbedoOver = 0; L.st_AlbedoUnder = 0; L.st_AtmosLatent = 0; L.st_AtmosLatentSub = 0; L.st_AtmosSensible = 0; L.st_LongOverIn = 0;
  L.st_LongUnderIn = 0; L.st_LongUnderOut = 0; L.st_NetLongAtmos = 0; L.st_NetLongOver = 0; L.st_NetLongUnder = 0; L.st_NetShortAtmos = 0;
  L.st_NetShortGrnd = 0; L.st_NetShortOver = 0; L.st_NetShortUnder = 0; L.st_ShortOverIn = 0; L.st_ShortUnderIn = 0;
  L.st_advected_sensible = 0; L.st_advection = 0; L.st_canopy_advection = 0; L.st_canopy_latent = 0; L.st_canopy_latent_sub = 0;
  L.st_canopy_sensible = 0; L.st_canopy_refreeze = 0; L.st_deltaCC = 0; L.st_deltaH = 0; L.st_fusion = 0; L.st_grnd_flux = 0;
  L.st_latent = 0; L.st_latent_sub = 0; L.st_melt_energy = 0; L.st_refreeze_energy = 0; L.st_sensible = 0; L.st_snow_flux = 0;
  L.st_canopy_vapor_flux = 0; L.st_melt = 0; L.st_vapor_flux = 0; L.st_blowing_flux = 0; L.st_surface_flux = 0; L.st_canopyevap = 0;
  L.st_throughfall = 0; L.st_ppt = 0; L.st_cond_surface = 0; L.st_cond_overstory = 0;
#pragma unroll
  for (int l = 0; l < 3; l++) L.st_layerevap[l] = 0;
#pragma unroll
  for (int p = 0; p < NPET; p++) L.st_pot_evap[p] = 0;
}

struct SubStep {
  SolveSnowOut ss;
  double Tair, Tcanopy, VPcanopy, VPDcanopy, step_melt_energy;
  int UnderStory, hidx;
  SurfPost post;
};

template <int NN>
VIC_DEV void sf_begin(const Opt& o, const Forcing& fc, const StepConst& C, HruWork<NN>& w, SubLoop& L) {
  const int NF = o.NF, NR = o.NR;
  Snow& snow = w.snow;
  SnowEnergy& se = w.se;
  SoilEnergy& so = w.so;
  L.ok = 1;
  L.surf_atten = C.surf_atten;        // solve_snow may reset it (no overstory); the change persists over the sub-steps
  so.advection = 0; so.deltaCC = 0; so.refreeze_energy = 0;           // energy->advection / deltaCC / refreeze_energy = 0
  se.advection = 0; se.deltaCC = 0; se.refreeze_energy = 0;
  L.snow_flux = (snow.swq > 0) ? so.snow_flux : -(so.grnd_flux + so.deltaH + so.fusion);
  L.coverage = snow.coverage;
  L.vv_snow = w.vv; L.vv_soil = w.vv;
  L.vv_snow.canopyevap = 0; L.vv_soil.canopyevap = 0; L.vv_snow.throughfall = 0; L.vv_soil.throughfall = 0;
  w.evap[0] = w.evap[1] = w.evap[2] = 0;
  if (snow.swq > 0 || snow.snow_canopy > 0 || fc.flag(NR)) { L.hidx = 0; L.endhidx = NF; L.step_dt = o.snow_step; }
  else { L.hidx = NR; L.endhidx = NR + 1; L.step_dt = o.dt; }
  L.last_snow_coverage = snow.coverage;
  L.Tgrnd0 = w.nd.T[0];   // Tgrnd = energy->T[0] reads the CALLER's struct (surface_fluxes.c:423): start-of-step value
  L.step_Wdew = w.vv.Wdew;
  L.AlbedoUnder_orig = so.AlbedoUnder;     // &energy->AlbedoUnder of the caller's struct (solve_snow's AlbedoUnder argument)
  L.snow_inflow = 0;
  L.INCLUDE_SNOW = 0; L.N_steps = 0;
  L.delta_coverage = 0;
  zero_substep_sums(L);
  w.out_prec = w.out_rain = w.out_snow = 0;
}

// one snow sub-step up to the ground-surface root finder: solve_snow and the set-up of calc_surf_energy_bal
// (surface_fluxes.c:494-601)
template <int NN>
VIC_DEV void sf_sub_pre(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, const Dmy& dmy,
                        const StepConst& C, HruWork<NN>& w, SubLoop& L, SubStep& P, SurfEB& eb, SurfSolve& sv) {
  Snow& snow = w.snow;
  SnowEnergy& se = w.se;
  SoilEnergy& so = w.so;
  const int hidx = L.hidx;
  const double Tair = fc.v(VIC_F_AIR_TEMP, hidx) + cv.band(CPB_TFACTOR, C.band);
  const double step_prec = fc.v(VIC_F_PREC, hidx) / 1.0 * cv.band(CPB_PFACTOR, C.band);
  const double Tcanopy = Tair;
  const double VPcanopy = fc.v(VIC_F_VP, hidx), VPDcanopy = fc.v(VIC_F_VPD, hidx);
  // mass flux of blowing snow (surface_fluxes.c:439-453): once per sub-step, before the snow pack's energy balance
  if (!C.overstory && o.BLOWING && snow.swq > 0.) {
    const double Ls = (677. - 0.07 * snow.surf_temp) * 4.1868 * 1000.0;
    const double bf = calc_blowing_snow((double)L.step_dt, Tair, snow.last_snow, snow.surf_water, C.U.v[SNOW_COVERED], Ls,
                                        fc.v(VIC_F_DENSITY, hidx), fc.v(VIC_F_VP, hidx), C.z0.v[SNOW_COVERED], snow.depth, (float)C.lag_one,
                                        (float)C.sigma_slope, C.is_art_bare, (float)C.fetch, C.disp.v[CANOPY], C.z0.v[CANOPY]);
    if ((int)bf == (int)ERROR_VAL) L.ok = 0;
    snow.blowing_flux = bf * L.step_dt * SECPHOUR / RHO_W;
  } else snow.blowing_flux = 0.0;
  int UnderStory = NCASE;
  // snow_grnd_flux = -snow_flux is overwritten inside snow_melt
  // per-iteration resets (surface_fluxes.c:501-532)
  L.vv_snow.Wdew = L.step_Wdew; L.vv_soil.Wdew = L.step_Wdew;
  L.vv_snow.canopyevap = 0; L.vv_soil.canopyevap = 0;
  double layerevap[3] = {0, 0, 0};
  double ra_used[2] = {w.aero_resist_surface, w.aero_resist_overstory};
  snow.canopy_vapor_flux = 0; snow.vapor_flux = 0; snow.surface_flux = 0;
  const double LongUnderOut = so.LongUnderOut;
  const double step_snow_surf_temp = snow.surf_temp, step_snow_depth = snow.depth;

  PROF_T0(t_ss);
  SolveSnowOut ss = solve_snow(o, cv, vl, s3, fc, hidx, C.veg_idx, dmy, C.overstory != 0, C.is_art_bare != 0, C.bare_albedo, LongUnderOut,
                               Tcanopy, L.Tgrnd0, Tair, step_prec, L.AlbedoUnder_orig, C.Ra, C.U, C.disp, C.zref, C.z0, ra_used, L.coverage,
                               L.surf_atten, L.snow_inflow, UnderStory, L.step_dt, w.moist, w.ice, C.root, layerevap, snow, se, L.vv_snow);
  PROF_ADD(2, t_ss);
  PROF_WAVE(2);
  if (!ss.ok) L.ok = 0;
  L.delta_coverage = ss.delta_coverage;
  double step_melt_energy = ss.melt_energy;

  if (isnan(snow.surf_temp) && snow.swq > 0) {                       // surface_fluxes.c:553-560 (UNSTABLE_SNOW is never set)
    L.INCLUDE_SNOW = UnderStory + 1;
    so.advection = se.advection;
    snow.surf_temp = step_snow_surf_temp;
    step_melt_energy = 0;
  } else L.INCLUDE_SNOW = 0;

  P.ss = ss; P.Tair = Tair; P.Tcanopy = Tcanopy; P.VPcanopy = VPcanopy; P.VPDcanopy = VPDcanopy;
  P.step_melt_energy = step_melt_energy; P.UnderStory = UnderStory; P.hidx = hidx;

  surf_setup<NN>(o, cv, vl, s3, fc, hidx, C.veg_idx, dmy.month, C.is_art_bare != 0, C.overstory != 0, ss.Le, ss.LongUnderIn,
                 ss.NetLongSnow, ss.NetShortGrnd, ss.NetShortSnow, ss.OldTSurf, ss.ShortUnderIn, snow.albedo, se.latent, se.latent_sub,
                 se.sensible, Tcanopy, VPDcanopy, VPcanopy, L.delta_coverage, C.ice0, step_melt_energy, C.moist0, snow.coverage,
                 (step_snow_depth + snow.depth) / 2., C.bare_albedo, L.surf_atten, C.Ra, C.U, C.disp, C.zref, C.z0, ra_used, ss.melt, ss.ppt,
                 ss.rainfall, C.root, L.INCLUDE_SNOW, UnderStory, L.step_dt, w.moist, w.ice, layerevap, w.nd, so, snow, L.vv_soil, eb,
                 P.post, sv);
}

// the rest of the sub-step: calc_surf_energy_bal's bookkeeping, potential evaporation, sub-step sums
// (surface_fluxes.c:601-816).  Tprof/cntprof/fbmask: the soil profile of the final evaluation (finite-difference path).
template <int NN>
VIC_DEV void sf_sub_post(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, const Dmy& dmy,
                         const StepConst& C, HruWork<NN>& w, SubLoop& L, const SubStep& P, const SurfEB& eb, const SurfSolve& sv,
                         const double* Tprof, const int* cntprof, unsigned fbmask) {
  Snow& snow = w.snow;
  SnowEnergy& se = w.se;
  SoilEnergy& so = w.so;
  const SolveSnowOut& ss = P.ss;
  const int hidx = P.hidx, UnderStory = P.UnderStory;
  const int INCLUDE_SNOW = L.INCLUDE_SNOW;
  const double delta_coverage = L.delta_coverage;
  const double step_melt_energy = P.step_melt_energy;
  double layerevap[3], ra_used[2];
  SurfOut sf = surf_post<NN>(o, cv, s3, P.post, eb, sv, Tprof, cntprof, fbmask, w.moist, w.ice, w.layer_T, layerevap, ra_used, w.nd,
                             so, snow, L.vv_soil);
  if (!sf.ok) L.ok = 0;
  double step_melt = sf.melt, step_ppt = sf.ppt;
  if (INCLUDE_SNOW) step_ppt += step_melt;

  const double AtmosLatent = so.latent, AtmosLatentSub = so.latent_sub, AtmosSensible = so.sensible;
  const double NetLongAtmos = so.NetLongUnder, NetShortAtmos = so.NetShortUnder;
  w.Tcanopy = P.Tcanopy;

  // potential evaporation, surface_fluxes.c:658-693
  double stability_factor[2], ra_s[NPET], ra_o[NPET], pe[NPET];
  if (ra_used[0] == HUGE_RESIST) stability_factor[0] = HUGE_RESIST;
  else stability_factor[0] = ra_used[0] / vsel(C.Ra, UnderStory);
  if (ra_used[1] == ra_used[0]) stability_factor[1] = stability_factor[0];
  else if (ra_used[1] == HUGE_RESIST) stability_factor[1] = HUGE_RESIST;
  else stability_factor[1] = ra_used[1] / C.Ra.v[CANOPY];
#pragma unroll
  for (int p = 0; p < NPET; p++) {
    ra_s[p] = (stability_factor[0] == HUGE_RESIST) ? HUGE_RESIST : vsel(C.aero_pet[p], UnderStory) * stability_factor[0];
    ra_o[p] = (stability_factor[1] == HUGE_RESIST) ? HUGE_RESIST : C.aero_pet[p].v[CANOPY] * stability_factor[1];
  }
  PROF_T0(t_pe);
  compute_pot_evap(o, vl, C.veg_idx, dmy.month, fc.v(VIC_F_SHORTWAVE, hidx), NetLongAtmos, P.Tair, P.VPDcanopy, cv.s(CP_ELEVATION), ra_s,
                   ra_o, pe);
  PROF_ADD(5, t_pe);

  // store sub-step, surface_fluxes.c:699-816
  if (!C.is_art_bare) {
    if (snow.snow) { L.st_throughfall += L.vv_snow.throughfall; L.st_canopyevap += L.vv_snow.canopyevap; L.vv_soil.Wdew = L.vv_snow.Wdew; }
    else { L.st_throughfall += L.vv_soil.throughfall; L.st_canopyevap += L.vv_soil.canopyevap; L.vv_snow.Wdew = L.vv_soil.Wdew; }
    L.step_Wdew = L.vv_soil.Wdew;
  }
#pragma unroll
  for (int l = 0; l < 3; l++) L.st_layerevap[l] += layerevap[l];
  L.st_ppt += step_ppt;
  L.st_cond_surface += (ra_used[0] > 0) ? 1 / ra_used[0] : HUGE_RESIST;
  L.st_cond_overstory += (ra_used[1] > 0) ? 1 / ra_used[1] : HUGE_RESIST;
  if (!C.is_art_bare) L.st_canopy_vapor_flux += snow.canopy_vapor_flux;
  L.st_melt += step_melt;
  L.st_vapor_flux += snow.vapor_flux;
  L.st_surface_flux += snow.surface_flux;
  L.st_blowing_flux += snow.blowing_flux;
  w.out_prec += ss.out_prec * 1.0; w.out_rain += ss.out_rain * 1.0; w.out_snow += ss.out_snow * 1.0;
  if (INCLUDE_SNOW) {
    se.advected_sensible = so.advected_sensible;   // never written on the soil side: last step's average
    se.advection = so.advection; se.deltaCC = so.deltaCC; se.latent = so.latent; se.latent_sub = so.latent_sub;
    se.refreeze_energy = so.refreeze_energy; se.sensible = so.sensible; se.snow_flux = so.snow_flux;
  }
  L.st_AlbedoOver += se.AlbedoOver;
  L.st_AlbedoUnder += so.AlbedoUnder;
  L.st_AtmosLatent += AtmosLatent; L.st_AtmosLatentSub += AtmosLatentSub; L.st_AtmosSensible += AtmosSensible;
  L.st_LongOverIn += se.LongOverIn;
  L.st_LongUnderIn += ss.LongUnderIn;
  L.st_LongUnderOut += so.LongUnderOut;
  L.st_NetLongAtmos += NetLongAtmos;
  L.st_NetLongOver += se.NetLongOver;
  L.st_NetLongUnder += so.NetLongUnder;
  L.st_NetShortAtmos += NetShortAtmos;
  L.st_NetShortGrnd += ss.NetShortGrnd;
  L.st_NetShortOver += se.NetShortOver;
  L.st_NetShortUnder += so.NetShortUnder;
  L.st_ShortOverIn += se.ShortOverIn;
  L.st_ShortUnderIn += 0.0;   // soil_energy.ShortUnderIn is never assigned for non-glacier HRUs: it keeps the 0 of
                              // initialize_model_state.c:278 through surface_fluxes.c:781,859
  L.st_canopy_advection += se.canopy_advection; L.st_canopy_latent += se.canopy_latent; L.st_canopy_latent_sub += se.canopy_latent_sub;
  L.st_canopy_sensible += se.canopy_sensible; L.st_canopy_refreeze += se.canopy_refreeze;
  L.st_deltaH += so.deltaH; L.st_fusion += so.fusion; L.st_grnd_flux += so.grnd_flux; L.st_latent += so.latent;
  L.st_latent_sub += so.latent_sub; L.st_melt_energy += step_melt_energy; L.st_sensible += so.sensible;
  if (snow.swq == 0 && INCLUDE_SNOW) {
    if (L.last_snow_coverage == 0) L.last_snow_coverage = 1;           // pointer test always true, SURVEY.md Appendix C #5
    L.st_advected_sensible += se.advected_sensible * L.last_snow_coverage;
    L.st_advection += se.advection * L.last_snow_coverage;
    L.st_deltaCC += se.deltaCC * L.last_snow_coverage;
    L.st_snow_flux += so.snow_flux * L.last_snow_coverage;
    L.st_refreeze_energy += se.refreeze_energy * L.last_snow_coverage;
  } else if (snow.snow || INCLUDE_SNOW) {
    const double cf = (snow.coverage + delta_coverage);
    L.st_advected_sensible += se.advected_sensible * cf;
    L.st_advection += se.advection * cf;
    L.st_deltaCC += se.deltaCC * cf;
    L.st_snow_flux += so.snow_flux * cf;
    L.st_refreeze_energy += se.refreeze_energy * cf;
  }
#pragma unroll
  for (int p = 0; p < NPET; p++) L.st_pot_evap[p] += pe[p];
  L.N_steps++;
  L.hidx += 1;
}

// step averages and runoff (surface_fluxes.c:818-948)
template <int NN>
VIC_DEV bool sf_end(const Opt& o, const CellView& cv, const Soil3& s3, const StepConst& C, HruWork<NN>& w, SubLoop& L) {
  Snow& snow = w.snow;
  SnowEnergy& se = w.se;
  SoilEnergy& so = w.so;
  const double N = (double)L.N_steps;
  snow.vapor_flux = L.st_vapor_flux; snow.blowing_flux = L.st_blowing_flux; snow.surface_flux = L.st_surface_flux;
  snow.canopy_vapor_flux = L.st_canopy_vapor_flux; snow.melt = L.st_melt;
  double ppt = L.st_ppt;

  // *energy = soil_energy, then the step averages (surface_fluxes.c:842-881)
  w.AlbedoOver_avg = L.st_AlbedoOver / N;
  so.AlbedoUnder = L.st_AlbedoUnder / N;
  w.AtmosLatent = L.st_AtmosLatent / N; w.AtmosLatentSub = L.st_AtmosLatentSub / N; w.AtmosSensible = L.st_AtmosSensible / N;
  w.LongOverIn_avg = L.st_LongOverIn / N;
  w.LongUnderIn = L.st_LongUnderIn / N;
  so.LongUnderOut = L.st_LongUnderOut / N;
  w.NetLongAtmos = L.st_NetLongAtmos / N;
  w.NetLongOver_avg = L.st_NetLongOver / N;
  so.NetLongUnder = L.st_NetLongUnder / N;
  w.NetShortAtmos = L.st_NetShortAtmos / N;
  so.NetShortGrnd = L.st_NetShortGrnd / N;
  w.NetShortOver_avg = L.st_NetShortOver / N;
  so.NetShortUnder = L.st_NetShortUnder / N;
  w.ShortOverIn_avg = L.st_ShortOverIn / N;
  w.ShortUnderIn_avg = L.st_ShortUnderIn / N;
  so.advected_sensible = L.st_advected_sensible / N;
  se.canopy_advection = L.st_canopy_advection / N; se.canopy_latent = L.st_canopy_latent / N;
  se.canopy_latent_sub = L.st_canopy_latent_sub / N; se.canopy_refreeze = L.st_canopy_refreeze / N;
  se.canopy_sensible = L.st_canopy_sensible / N;
  so.deltaH = L.st_deltaH / N; so.fusion = L.st_fusion / N; so.grnd_flux = L.st_grnd_flux / N; so.latent = L.st_latent / N;
  so.latent_sub = L.st_latent_sub / N; so.melt_energy = L.st_melt_energy / N; so.sensible = L.st_sensible / N;
  if (snow.snow || L.INCLUDE_SNOW) {
    so.advection = L.st_advection / N; so.deltaCC = L.st_deltaCC / N; so.refreeze_energy = L.st_refreeze_energy / N;
    so.snow_flux = L.st_snow_flux / N;
  }

  if (!C.is_art_bare) {
    w.vv.throughfall = L.st_throughfall;
    w.vv.canopyevap = L.st_canopyevap;
    w.vv.Wdew = snow.snow ? L.vv_snow.Wdew : L.vv_soil.Wdew;
  }
#pragma unroll
  for (int l = 0; l < 3; l++) w.evap[l] = L.st_layerevap[l];
  if (L.st_cond_surface > 0 && L.st_cond_surface < HUGE_RESIST) w.aero_resist_surface = 1 / (L.st_cond_surface / N);
  else if (L.st_cond_surface >= HUGE_RESIST) w.aero_resist_surface = 0;
  else w.aero_resist_surface = HUGE_RESIST;
  if (L.st_cond_overstory > 0 && L.st_cond_overstory < HUGE_RESIST) w.aero_resist_overstory = 1 / (L.st_cond_overstory / N);
  else if (L.st_cond_overstory >= HUGE_RESIST) w.aero_resist_overstory = 0;
  else w.aero_resist_overstory = HUGE_RESIST;
#pragma unroll
  for (int p = 0; p < NPET; p++) w.pot_evap[p] = L.st_pot_evap[p] / N;

  // runoff, surface_fluxes.c:941-948 (excess_moist is 0 after initialisation)
  w.inflow = ppt;
  PROF_T0(t_ro);
  RunoffOut ro = runoff_step(o, cv, s3, w.moist, w.ice, w.evap, ppt);
  w.runoff = ro.runoff; w.baseflow = ro.baseflow; w.asat = ro.asat;
  PROF_ADD(6, t_ro);
  PROF_T0(t_zw);
  w.zwt = wrap_compute_zwt(cv, s3, w.moist);
  if (o.FULL_ENERGY || o.FROZEN_SOIL) distribute_node_moisture_properties<NN>(o, cv, s3, w.nd, w.moist);
  PROF_ADD(7, t_zw);
  return L.ok != 0;
}

// the whole of surface_fluxes in one lane (QUICK_FLUX: no soil-profile solve, the root finder is a short lane loop)
template <int NN>
VIC_DEV bool surface_fluxes(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, const Dmy& dmy,
                            const StepConst& C, HruWork<NN>& w) {
  SubLoop L;
  sf_begin<NN>(o, fc, C, w, L);
  do {
    SubStep P;
    SurfEB eb;
    SurfSolve sv;
    sf_sub_pre<NN>(o, cv, vl, s3, fc, dmy, C, w, L, P, eb, sv);
    PROF_T0(t_sf);
    while (sv.stage != SurfSolve::DONE) {
      const double fx = eb.eval(o, s3, sv.x, 0., 0.);
      surf_solve_consume(o, sv, eb, eb, fx);
    }
    PROF_ADD(3, t_sf);
    sf_sub_post<NN>(o, cv, vl, s3, fc, dmy, C, w, L, P, eb, sv, nullptr, nullptr, 0u);
  } while (L.hidx < L.endhidx);
  return sf_end<NN>(o, cv, s3, C, w, L);
}

}  // namespace vic
