// vic_surface.hpp — ground-surface energy balance, soil temperature profile, solve_snow (device only, gfx950).
#pragma once
#include "vic_snow.hpp"

namespace vic {

// estimate_T1.c:8-47
VIC_DEV double estimate_T1(double Ts, double T1_old, double T2, double D1, double D2, double kappa1, double kappa2, double Cs2,
                           double dp, double delta_t) {
  double C1 = Cs2 * dp / D2 * (1. - exp(-D2 / dp));
  double C2 = -(1. - exp(D1 / dp)) * exp(-D2 / dp);
  double C3 = kappa1 / D1 - kappa2 / D1 + kappa2 / D1 * exp(-D1 / dp);
  return (kappa1 / 2. / D1 / D2 * (Ts) + C1 / delta_t * T1_old + (2. * C2 - 1. + exp(-D1 / dp)) * kappa2 / 2. / D1 / D2 * T2)
         / (C1 / delta_t + kappa2 / D1 / D2 * C2 + C3 / 2. / D2);
}

// ------------------------------------------------------------------------------------------------
// Finite-difference soil temperature profile: what the ground-surface balance hands to the profile-solve kernel
// (vic_profile.hpp).  One record per node, PREC doubles, records of one HRU contiguous ("item block").
//
// The explicit scheme (frozen_soil.c:161-213, 380-468) updates node j from its neighbours TL = T[j+1], TU = T[j-1]:
//   unfrozen node:  T = N / S                                  with N = A*T0 + B*(TL-TU) + C*TL + D*TU + E*(0-ice0), S = A+C+D
//   frozen node:    root of  -A(T-T0) + B(TL-TU) + C(TL-T) - D(T-TU) + E(ice(T)-ice0)  (soil_thermal_eqn.c:47-131)
//                   which is, term by term,  N - S*T + E*ice(T)   with the SAME N and S,
//   ice(T) = clip(moist - u(T)), u(T) = max_moist * (-Lf*T/273.16/(g*bubble/100))^(-2/(expt-3)) clipped to max_moist
//   (maximum_unfrozen_water, soil_conduction.c:830-863); EXP_TRANS: N and S of frozen_soil.c:194-212 instead.
// So a record holds what both forms need, with everything that is constant over the Brent solve on Tsurf folded once
// per sub-step here instead of once per node visit there:
//   [0] T0     previous-step temperature (node 0: replaced by the trial surface temperature at solve time)
//   [1] A*T0   (node 0: frozen_on flag)
//   [2] B  [3] C  [4] D  [5] E*(0-ice0)  [6] S = A+C+D  (EXP_TRANS: A+2C)        -> N and the closed-form update keep the
//                                                                                  reference's operation order exactly
//   [7] G = E*max_moist*kappa^Y, kappa = Lf/273.16/(g*bubble/100)   [8] Y = -2/(expt-3)       -> E*u(T) = G*(-T)^Y
//   [9] E*moist  [10] E*max_moist                                                               (clip levels of E*u, E*ice)
// max_moist/bubble/expt are the node arrays ("fixed") or the LAYER arrays indexed by node as shipped ("compat",
// frozen_soil.c:218-221: max_moist in mm for j < 3, then the node arrays shifted by 3).
// A-D depend only on kappa, Cs, the node geometry and dt, i.e. they are the same for every residual evaluation of one
// Brent solve on Tsurf (upstream keeps them in static arrays for that reason; SURVEY.md Finding 1.1).
// ------------------------------------------------------------------------------------------------
constexpr int PREC = 11;
enum { PR_T0 = 0, PR_AT0, PR_B, PR_C, PR_D, PR_EI, PR_S, PR_G, PR_Y, PR_EMOIST, PR_EMM };

template <int NN>
VIC_DEV void profile_item_store(const Opt& o, const CellView& cv, const Soil3& s3, const Nodes<NN>& nd, double deltat, bool frozen_on,
                                double* __restrict__ blk) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  const double Dp = cv.s(CP_DP);
  const double Bexp = o.EXP_TRANS ? log(Dp + 1.) / (double)(Nn - 1) : 0.0;
#pragma unroll
  for (int j = 0; j < NN; j++) {
    if (j >= Nn) continue;
    double* r = blk + j * PREC;
    r[PR_T0] = nd.T[j];
    if (j == 0) { r[PR_AT0] = frozen_on ? 1.0 : 0.0; continue; }
    if (!(j < Nn - 1 || o.NOFLUX)) continue;
    const double kup = (j < Nn - 1) ? nd.kappa[(j + 1 < NN) ? j + 1 : j] : nd.kappa[j];
    double A, C, D, E;
    if (!o.EXP_TRANS) {
      const double al = cv.node(CPN_ALPHA, j - 1), be = cv.node(CPN_BETA, j - 1), ga = cv.node(CPN_GAMMA, j - 1);
      A = nd.Cs[j] * al * al;
      C = 2 * deltat * nd.kappa[j] * al / ga;
      D = 2 * deltat * nd.kappa[j] * al / be;
      E = ICE_DENSITY * LF * al * al;
      r[PR_S] = A + C + D;
    } else {
      const double z1 = cv.node(CPN_ZSUM, j) + 1;
      A = 4 * Bexp * Bexp * nd.Cs[j] * z1 * z1;
      C = 4 * deltat * nd.kappa[j];
      D = 2 * deltat * nd.kappa[j] * Bexp;
      E = 4 * Bexp * Bexp * ICE_DENSITY * LF * z1 * z1;
      r[PR_S] = A + 2. * C;
    }
    r[PR_AT0] = A * nd.T[j];
    r[PR_B] = (kup - nd.kappa[j - 1]) * deltat;
    r[PR_C] = C; r[PR_D] = D;
    r[PR_EI] = E * (0. - nd.ice[j]);
    double mm, bub, ex;
    if (o.frozen_compat) {
      if (j < 3) { mm = s3.max_moist[j]; bub = cv.lay(CPL_BUBBLE, j); ex = cv.lay(CPL_EXPT, j); }
      else { mm = cv.node(CPN_MAX_MOIST, j - 3); bub = cv.node(CPN_BUBBLE, j - 3); ex = cv.node(CPN_EXPT, j - 3); }
    } else { mm = cv.node(CPN_MAX_MOIST, j); bub = cv.node(CPN_BUBBLE, j); ex = cv.node(CPN_EXPT, j); }
    const double Y = -(2.0 / (ex - 3.0));
    const double kap = LF / 273.16 / (9.81 * bub / 100.);
    r[PR_Y] = Y;
    r[PR_G] = frozen_on ? E * mm * pow_pos(kap, Y) : 0.0;
    r[PR_EMOIST] = E * nd.moist[j];
    r[PR_EMM] = E * mm;
  }
}

// Residual of the ground-surface energy balance (func_surf_energy_bal.c:9-403).  Plain data, so that the evaluation
// kernel can park it in HBM between the rounds of the Brent iteration on Tsurf: Const is written once per sub-step,
// Mut ("last evaluation wins", the reference passes these by pointer) after every evaluation.
// The residual's inputs that depend on the cell, the vegetation class and the forcing only: not parked, every kernel that
// needs a SurfEB fills them from their (cache-resident) tables (surf_cell_fill) -- 16 words less context per evaluation.
struct SurfEBCell {
  VegMonth vm;
  double D1, D2, bubble, dp, expt, max_moist, elevation, b_infilt, resid0, atmos_density, atmos_pressure;
};
// The members are grouped by who reads them (the finite-difference pipeline parks and fetches a group only for the HRUs
// whose evaluations use it -- EBG_* class bits, decided once per root find in surf_setup's caller):
struct SurfEBConst {
  // [post] also read by the bookkeeping after the root find (surf_post)
  int VEG, frozen_on, INCLUDE_SNOW, SNOWING, overstory, hidx;      // hidx: the forcing sub-step of this root find
  double T2;                      // bottom of the column: estimate_T1 (QUICK_FLUX); surf_post when neither FULL_ENERGY nor frozen soil
  // [always] every evaluation
  double delta_t, Cs1, T1_old, Ts_old, kappa1, kappa2;
  double NetShortBare, NetShortGrnd, NetShortSnow, Tair, LongBareIn, surf_atten, vpd;
  double Le, Advection, melt_energy, snow_coverage;
  double U_under, zref_under, disp_under, z0_under, ra_under;
  // [frozen] frozen_on
  double ice0, moist;
  // [snowcov] snow_coverage > 0 and not INCLUDE_SNOW (with SurfEBMut::Tsnow_surf)
  double kappa_snow;
  // [incl] INCLUDE_SNOW, the thin snowpack solved together with the ground surface
  double LongSnowIn, vp, OldTSurf, snow_density, snow_swq, snow_water;
  // [evap] not SNOWING: arno_evap (layer 0) or canopy_evap
  double lmoist[3], lice[3];
  // [canopy] VEG and not SNOWING: canopy_evap / transpiration (with SurfEBMut::ra_used[1])
  double Wdew, rainfall, root[3];
  // [quick] QUICK_FLUX only (estimate_T1): never parked
  double Cs2;
};
enum { EBG_FROZEN = 1, EBG_SNOWCOV = 2, EBG_INCL = 4, EBG_EVAP = 8, EBG_CANOPY = 16 };
struct SurfEBMut {
  // [feed] carried from one evaluation to the next -- only with INCLUDE_SNOW (the mass fluxes of latent_heat_from_snow are
  // in/out)
  double vapor_flux, blowing_flux, surface_flux;
  // [keep] inputs of an evaluation that it may also assign: whichever of them a root find assigns at all, every evaluation
  // assigns before it uses it (the branch is decided by the constant flags), so an evaluation reads the values the set-up
  // parked and only the final evaluation's values are ever written back
  double deltaCC, NetLongSnow, fusion;          // read by every evaluation (inputs unless INCLUDE_SNOW / frozen_on)
  double Tsnow_surf;                            // read with [snowcov]
  double ra_used[2];                            // [1] read with [canopy]; [0] is an output
  VegVar vv;
  double layerevap[3];
  double refreeze_energy;
  // [out] assigned by every evaluation before any use: only the final evaluation's values are ever read
  double Tnew2;                   // Tnew_node[2] of the last evaluation
  double NetLongBare, T1, deltaH, grnd_flux, latent_heat, latent_heat_sub, sensible_heat, snow_flux, error;
};

struct SurfEB : SurfEBCell, SurfEBConst, SurfEBMut {
  // T1_fd / T2_fd: nodes 1 and 2 of the finite-difference profile solved for this Ts (ignored with QUICK_FLUX)
  VIC_DEV double eval(const Opt& o, const Soil3& s3, double Ts, double T1_fd, double T2_fd) {
    PROF_WAVE(7); PROF_LANE(8);
    const double TMean = Ts;
    const double Tmp = TMean + KELVIN;
    // (through locals: two members assigned on alternative paths end up behind a selected address, i.e. in scratch)
    double Tsn = Tsnow_surf, sflux = 0;
    if (snow_coverage > 0 && !INCLUDE_SNOW) sflux = (kappa_snow * (Tsn - TMean));
    else if (INCLUDE_SNOW) Tsn = TMean;
    snow_flux = sflux; Tsnow_surf = Tsn;
    const double att = (snow_coverage + (1. - snow_coverage) * surf_atten);
    if (o.QUICK_FLUX) {
      T1 = estimate_T1(TMean, T1_old, T2, D1, D2, kappa1, kappa2, Cs2, dp, delta_t);
      if (o.GRND_FLUX_TYPE == VIC_GF_406) grnd_flux = att * (kappa1 / D1 * ((T1) - TMean));
      else grnd_flux = att * (kappa1 / D1 * ((T1) - TMean) + (kappa2 / D2 * (1. - exp(-D1 / dp)) * (T2 - (T1)))) / 2.;
    } else {
      T1 = T1_fd;                                              // solve_T_profile with T_node[0] = TMean (func_surf_energy_bal.c:190)
      Tnew2 = T2_fd;
      if (o.GRND_FLUX_TYPE == VIC_GF_406) grnd_flux = att * (kappa1 / D1 * ((T1) - TMean));
      else grnd_flux = att * (kappa1 / D1 * ((T1) - TMean) + (kappa2 / D2 * (Tnew2 - (T1)))) / 2.;
    }
    if (o.GRND_FLUX_TYPE == VIC_GF_FULL) deltaH = att * (Cs1 * ((Ts_old + T1_old) - (TMean + T1)) * D1 / delta_t / 2.);
    else deltaH = (Cs1 * ((Ts_old + T1_old) - (TMean + T1)) * D1 / delta_t / 2.);
    if (frozen_on) {
      double ice;
      if ((TMean + T1) / 2. < 0.) {
        ice = moist - maximum_unfrozen_water((TMean + T1) / 2., max_moist, bubble, expt);
        if (ice < 0.) ice = 0.;
      } else ice = 0.;
      if (o.GRND_FLUX_TYPE == VIC_GF_FULL) fusion = att * (-ICE_DENSITY * LF * (ice0 - ice) * D1 / delta_t);
      else fusion = (-ICE_DENSITY * LF * (ice0 - ice) * D1 / delta_t);
    }
    if (INCLUDE_SNOW) {
      if (TMean > 0) deltaCC = CH_ICE * (snow_swq - snow_water) * (0 - OldTSurf) / delta_t;
      else deltaCC = CH_ICE * (snow_swq - snow_water) * (TMean - OldTSurf) / delta_t;
      refreeze_energy = (snow_water * LF * snow_density) / delta_t;
      deltaCC *= snow_coverage;
      refreeze_energy *= snow_coverage;
    }
    const double LongBareOut = STEFAN_B * Tmp * Tmp * Tmp * Tmp;
    if (INCLUDE_SNOW) NetLongSnow = (LongSnowIn - snow_coverage * LongBareOut);
    NetLongBare = (LongBareIn - (1. - snow_coverage) * LongBareOut);
    const double NetBareRad = (NetShortBare + (NetLongBare) + grnd_flux + deltaH + fusion);

    if (U_under > 0.0 && overstory && SNOWING) ra_used[0] = ra_under / stability_correction(zref_under, 0.f, TMean, Tair, U_under, z0_under);
    else if (U_under > 0.0) ra_used[0] = ra_under / stability_correction(zref_under, disp_under, TMean, Tair, U_under, z0_under);
    else ra_used[0] = HUGE_RESIST;

    double Evap;
    if (VEG && !SNOWING && vm.LAI > 0) {
      Evap = canopy_evap(vm, s3, lmoist, lice, vv, true, Wdew, delta_t, NetBareRad, vpd, NetShortBare, Tair, ra_used[1], elevation,
                         rainfall, root, layerevap);
    } else if (!SNOWING) {
      double e0 = layerevap[0];
      Evap = arno_evap(lmoist[0], lice[0], NetBareRad, Tair, vpd, s3.depth[0], max_moist * s3.depth[0] * 1000., elevation, b_infilt,
                       ra_used[0], delta_t, resid0, e0);
      layerevap[0] = e0;
    } else Evap = 0.;

    latent_heat = -RHO_W * Le * Evap;
    latent_heat_sub = 0.;
    if (INCLUDE_SNOW) {
      double VaporMassFlux = vapor_flux * ICE_DENSITY / delta_t;
      double BlowingMassFlux = blowing_flux * ICE_DENSITY / delta_t;
      double SurfaceMassFlux = surface_flux * ICE_DENSITY / delta_t;
      double tl, tls;
      latent_heat_from_snow(atmos_density, vp, Le, atmos_pressure, ra_used[0], TMean, vpd, tl, tls, VaporMassFlux, BlowingMassFlux,
                            SurfaceMassFlux);
      latent_heat += tl * snow_coverage;
      latent_heat_sub = tls * snow_coverage;
      vapor_flux = VaporMassFlux * delta_t / ICE_DENSITY;
      blowing_flux = BlowingMassFlux * delta_t / ICE_DENSITY;
      surface_flux = SurfaceMassFlux * delta_t / ICE_DENSITY;
    } else latent_heat *= (1. - snow_coverage);

    if (snow_coverage < 1 || INCLUDE_SNOW) {
      sensible_heat = atmos_density * CP_AIR * (Tair - (TMean)) / ra_used[0];
      if (!INCLUDE_SNOW) (sensible_heat) *= (1. - snow_coverage);
    } else sensible_heat = 0.;

    double err = (NetBareRad + NetShortGrnd + NetShortSnow + 1. * (NetLongSnow)) + sensible_heat + (latent_heat + latent_heat_sub)
                 + sflux * snow_coverage + melt_energy + Advection - deltaCC;
    if (INCLUDE_SNOW) {
      if (Tsn == 0.0 && err > -(refreeze_energy)) {
        refreeze_energy = -err;
        err = 0.0;
      } else err += refreeze_energy;
    }
    error = err;
    return err;
  }
};

// What calc_surf_energy_bal keeps from its set-up for the bookkeeping after the solve
struct SurfPost {
  double NetLongSnow, NetShortGrnd, NetShortSnow, SnowAlbedo, SnowLatent, SnowLatentSub, SnowSensible, delta_coverage, snow_coverage,
         BareAlbedo, LongUnderIn, melt_energy, rainfall, NetShortBare, TmpNetShortSnow, melt_in, ppt_in, delta_t;
  int INCLUDE_SNOW, is_artificial_bare;
};

// The Brent iteration on Tsurf (calc_surf_energy_bal.c:430-487) and the final evaluation at the root (:489-506), one
// residual evaluation per call of surf_solve_consume: the same code runs inside a lane's loop (QUICK_FLUX) and across
// launches of the evaluation kernel (finite-difference profile).
struct SurfSolve {
  enum { ROOT = 0, FINAL = 1, DONE = 2, ROOT_QUICK = 3 };   // ROOT_QUICK: QUICK_SOLVE's first iteration, on the shortened column
  Brent br;
  double x, Tsurf, snow_surf_temp, Ts_old, error;
  int stage, fbflag, fbcount, ok;
  int final_slot, on_record;   // finite-difference pipeline: the profile record of the final evaluation's solve, and
                               // whether that solve is already on record (no profile solve before the final evaluation)
};

VIC_DEV void surf_solve_begin(const Opt& o, SurfSolve& sv, double T0, double Tair, bool INCLUDE_SNOW, double snow_surf_temp) {
  sv.Ts_old = T0; sv.snow_surf_temp = snow_surf_temp; sv.fbflag = 0; sv.fbcount = 0; sv.ok = 1; sv.error = 0; sv.Tsurf = 0;
  sv.final_slot = 0; sv.on_record = 0;
  if (o.FULL_ENERGY) {
    double T_lower, T_upper;
    if (INCLUDE_SNOW) { T_lower = T0 - SURF_DT; T_upper = 0.; }
    else { T_lower = 0.5 * (T0 + Tair) - SURF_DT; T_upper = 0.5 * (T0 + Tair) + SURF_DT; }
    sv.br.start(T_lower, T_upper);
    sv.stage = (o.QUICK_SOLVE && !o.QUICK_FLUX) ? SurfSolve::ROOT_QUICK : SurfSolve::ROOT; sv.x = sv.br.x;
  } else {
    sv.br.start(0, 0);
    sv.Tsurf = Tair; sv.x = Tair; sv.stage = SurfSolve::FINAL;
  }
}

// ec: the residual's constant inputs (QUICK_SOLVE's second iteration starts from the same bracket as the first)
VIC_DEV void surf_solve_consume(const Opt& o, SurfSolve& sv, SurfEBMut& m, const SurfEBConst& ec, double fx) {
  if (sv.stage == SurfSolve::ROOT || sv.stage == SurfSolve::ROOT_QUICK) {
    sv.br.advance(fx);
    if (sv.br.phase == Brent::DONE) {
      double Tsurf = sv.br.result;
      if (is_error(Tsurf)) {
        if (o.TFALLBACK) { Tsurf = sv.Ts_old; sv.fbflag = 1; sv.fbcount++; }
        else sv.ok = 0;
      }
      if (sv.stage == SurfSolve::ROOT_QUICK && sv.ok && sv.Ts_old * Tsurf < 0) {
        // the surface changes sign: iterate again on the whole column (calc_surf_energy_bal.c:400-480), a fresh object
        double T_lower, T_upper;
        if (ec.INCLUDE_SNOW) { T_lower = sv.Ts_old - SURF_DT; T_upper = 0.; }
        else { T_lower = 0.5 * (sv.Ts_old + ec.Tair) - SURF_DT; T_upper = 0.5 * (sv.Ts_old + ec.Tair) + SURF_DT; }
        sv.br.start(T_lower, T_upper);
        sv.stage = SurfSolve::ROOT; sv.x = sv.br.x;
        m.Tsnow_surf = sv.snow_surf_temp;
      } else {
        sv.Tsurf = Tsurf; sv.x = Tsurf; sv.stage = SurfSolve::FINAL;
        m.Tsnow_surf = sv.snow_surf_temp;        // the final evaluation starts from the stored pack temperature
      }
    } else sv.x = sv.br.x;
  } else {
    sv.error = fx;
    if (fx == ERROR_VAL) sv.ok = 0;
    sv.stage = SurfSolve::DONE;
  }
}

// calc_surf_energy_bal.c:7-428: everything before the root finder.  lmoist/lice/layerevap are the three soil layers of
// the current sub-step.
// the SurfEBCell part, from the tables (the expressions of calc_surf_energy_bal.c:225-262 as surf_setup had them)
VIC_DEV void surf_cell_fill(SurfEBCell& c, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, int hidx, int veg_idx,
                            int month) {
  c.vm = veg_month(vl, veg_idx, month);
  c.D1 = cv.node(CPN_ZSUM, 1) - cv.node(CPN_ZSUM, 0); c.D2 = cv.node(CPN_ZSUM, 2) - cv.node(CPN_ZSUM, 1);
  c.bubble = cv.lay(CPL_BUBBLE, 0); c.dp = cv.s(CP_DP); c.expt = cv.lay(CPL_EXPT, 0);
  c.max_moist = s3.max_moist[0] / (s3.depth[0] * 1000.);
  c.elevation = cv.s(CP_ELEVATION); c.b_infilt = cv.s(CP_B_INFILT); c.resid0 = s3.resid_moist[0];
  c.atmos_density = fc.v(VIC_F_DENSITY, hidx); c.atmos_pressure = fc.v(VIC_F_PRESSURE, hidx);
}

template <int NN>
VIC_DEV void surf_setup(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, int hidx, int veg_idx,
                        int month, bool is_artificial_bare, bool overstory, double Le, double LongUnderIn, double NetLongSnow,
                        double NetShortGrnd, double NetShortSnow, double OldTSurf, double ShortUnderIn, double SnowAlbedo,
                        double SnowLatent, double SnowLatentSub, double SnowSensible, double Tair, double VPDcanopy, double VPcanopy,
                        double delta_coverage, double ice0, double melt_energy, double moist0, double snow_coverage,
                        double snow_depth_avg, double BareAlbedo, double surf_atten, const Vc& Ra, const Vc& U, const Vc& disp,
                        const Vc& zref, const Vc& z0, const double* ra_used, double melt_in, double ppt_in, double rainfall,
                        const double* root, int INCLUDE_SNOW, int UnderStory, int dt, const double* lmoist, const double* lice,
                        const double* layerevap, const Nodes<NN>& nd, const SoilEnergy& e, const Snow& snow, const VegVar& vv,
                        SurfEB& eb, SurfPost& P, SurfSolve& sv) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  surf_cell_fill(eb, cv, vl, s3, fc, hidx, veg_idx, month);
  const VegMonth vm = eb.vm;
  const bool frozen_on = (cv.s(CP_FS_ACTIVE) != 0.0) && o.FROZEN_SOIL;
  const double delta_t = (double)dt * 3600.;
  const double kappa_snow = (snow.depth > 0.) ? K_SNOW * (snow.density) * (snow.density) / snow_depth_avg : 0;
  const double NetShortBare = (ShortUnderIn * (1. - (snow_coverage + delta_coverage)) * (1. - BareAlbedo)
                               + ShortUnderIn * (delta_coverage) * (1. - SnowAlbedo));
  const double LongBareIn = (1. - snow_coverage) * LongUnderIn;
  double TmpNetLongSnow, TmpNetShortSnow, LongSnowIn;
  if (INCLUDE_SNOW || snow.swq == 0) { TmpNetLongSnow = NetLongSnow; TmpNetShortSnow = NetShortSnow; LongSnowIn = snow_coverage * LongUnderIn; }
  else { TmpNetShortSnow = 0.; TmpNetLongSnow = 0.; LongSnowIn = 0.; }

  P.NetLongSnow = NetLongSnow; P.NetShortGrnd = NetShortGrnd; P.NetShortSnow = NetShortSnow; P.SnowAlbedo = SnowAlbedo;
  P.SnowLatent = SnowLatent; P.SnowLatentSub = SnowLatentSub; P.SnowSensible = SnowSensible; P.delta_coverage = delta_coverage;
  P.snow_coverage = snow_coverage; P.BareAlbedo = BareAlbedo; P.LongUnderIn = LongUnderIn; P.melt_energy = melt_energy;
  P.rainfall = rainfall; P.NetShortBare = NetShortBare; P.TmpNetShortSnow = TmpNetShortSnow; P.melt_in = melt_in; P.ppt_in = ppt_in;
  P.delta_t = delta_t; P.INCLUDE_SNOW = INCLUDE_SNOW; P.is_artificial_bare = is_artificial_bare;

  eb.VEG = (!is_artificial_bare) && (vm.LAI > 0.0);
  eb.frozen_on = frozen_on; eb.INCLUDE_SNOW = INCLUDE_SNOW != 0; eb.SNOWING = snow.snow != 0; eb.overstory = overstory; eb.hidx = hidx;
  eb.delta_t = delta_t; eb.Cs1 = e.Cs[0]; eb.Cs2 = e.Cs[1];
  eb.T1_old = nd.T[1]; eb.T2 = nd.T[Nn - 1 < NN ? Nn - 1 : NN - 1]; eb.Ts_old = nd.T[0];
  eb.ice0 = ice0;
  eb.kappa1 = e.kappa[0]; eb.kappa2 = e.kappa[1];
  eb.moist = moist0;
  eb.NetShortBare = NetShortBare; eb.NetShortGrnd = NetShortGrnd; eb.NetShortSnow = TmpNetShortSnow; eb.Tair = Tair;
  eb.LongBareIn = LongBareIn; eb.LongSnowIn = LongSnowIn; eb.surf_atten = surf_atten; eb.vp = VPcanopy; eb.vpd = VPDcanopy;
  eb.Wdew = vv.Wdew; eb.rainfall = rainfall; eb.Le = Le; eb.Advection = e.advection; eb.OldTSurf = OldTSurf;
  eb.kappa_snow = kappa_snow; eb.melt_energy = melt_energy; eb.snow_coverage = snow_coverage; eb.snow_density = snow.density;
  eb.snow_swq = snow.swq; eb.snow_water = snow.surf_water;
  eb.U_under = vsel(U, UnderStory); eb.zref_under = vsel(zref, UnderStory); eb.disp_under = vsel(disp, UnderStory);
  eb.z0_under = z0.v[UnderStory]; eb.ra_under = vsel(Ra, UnderStory);
#pragma unroll
  for (int l = 0; l < 3; l++) { eb.lmoist[l] = lmoist[l]; eb.lice[l] = lice[l]; eb.root[l] = root[l]; eb.layerevap[l] = layerevap[l]; }
  eb.Tsnow_surf = snow.surf_temp;
  eb.Tnew2 = 0;
  eb.ra_used[0] = ra_used[0]; eb.ra_used[1] = ra_used[1];
  eb.vv = vv;
  eb.deltaCC = e.deltaCC; eb.refreeze_energy = e.refreeze_energy; eb.vapor_flux = snow.vapor_flux;
  eb.blowing_flux = snow.blowing_flux; eb.surface_flux = snow.surface_flux;
  eb.NetLongBare = 0; eb.NetLongSnow = TmpNetLongSnow; eb.T1 = 0; eb.deltaH = e.deltaH; eb.fusion = e.fusion;
  eb.grnd_flux = e.grnd_flux; eb.latent_heat = e.latent; eb.latent_heat_sub = e.latent_sub; eb.sensible_heat = e.sensible;
  eb.snow_flux = e.snow_flux; eb.error = e.error;

  surf_solve_begin(o, sv, nd.T[0], Tair, INCLUDE_SNOW != 0, snow.surf_temp);
}

struct SurfOut { double Tsurf, melt, ppt; bool ok; };

// calc_surf_energy_bal.c:489-692: bookkeeping after the final evaluation.  Tnew/cntnew/fbmask: the finite-difference
// profile of the final evaluation (unused with QUICK_FLUX).  lice/lT: layer ice and temperature out.
template <int NN>
VIC_DEV SurfOut surf_post(const Opt& o, const CellView& cv, const Soil3& s3, const SurfPost& P, const SurfEB& eb, const SurfSolve& sv,
                          const double* Tprof, const int* cntprof, unsigned fbmask, const double* lmoist, double* lice, double* lT,
                          double* layerevap, double* ra_used, Nodes<NN>& nd, SoilEnergy& e, Snow& snow, VegVar& vv) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  SurfOut out;
  out.ok = sv.ok != 0; out.melt = P.melt_in; out.ppt = P.ppt_in;
  const double Tsurf = sv.Tsurf;
  const bool frozen_on = eb.frozen_on != 0;
  const bool INCLUDE_SNOW = P.INCLUDE_SNOW != 0;
  const double delta_t = P.delta_t;
  e.error = sv.error;
  e.deltaCC = eb.deltaCC; e.refreeze_energy = eb.refreeze_energy; e.deltaH = eb.deltaH; e.fusion = eb.fusion;
  e.grnd_flux = eb.grnd_flux; e.latent = eb.latent_heat; e.latent_sub = eb.latent_heat_sub; e.sensible = eb.sensible_heat;
  e.snow_flux = eb.snow_flux;
  snow.vapor_flux = eb.vapor_flux; snow.blowing_flux = eb.blowing_flux; snow.surface_flux = eb.surface_flux;
  ra_used[0] = eb.ra_used[0]; ra_used[1] = eb.ra_used[1];
  vv = eb.vv;
#pragma unroll
  for (int l = 0; l < 3; l++) layerevap[l] = eb.layerevap[l];
  const double TmpNetLongSnow = eb.NetLongSnow;
  const double NetLongBare = eb.NetLongBare;

  double Tnew[NN];
  int cntnew[NN];
#pragma unroll
  for (int n = 0; n < NN; n++) { Tnew[n] = 0; cntnew[n] = 0; }
  if (!o.QUICK_FLUX) {
#pragma unroll
    for (int n = 0; n < NN; n++)
      if (n < Nn) { Tnew[n] = Tprof[n]; cntnew[n] = cntprof[n]; }
  }
  if (o.QUICK_FLUX || !(o.FULL_ENERGY || frozen_on)) {
    Tnew[0] = Tsurf;
    Tnew[1] = eb.T1;
    Tnew[2] = eb.T2;
  }
  // calc_layer_average_thermal_props (frozen_soil.c:12-103)
  if (frozen_on) find_0_degree_fronts<NN>(o, cv, e, Tnew);
  else e.Nfrost = 0;
#pragma unroll
  for (int n = 0; n < NN; n++) nd.T[n] = Tnew[n];
  e.frozen = (e.Nfrost > 0) ? 1 : 0;
  if (o.QUICK_FLUX) estimate_layer_ice_content_quick_flux(o, cv, s3, nd.T[0], nd.T[1], lmoist, lice, lT);
  else if (!estimate_layer_ice_content<NN>(o, cv, s3, nd.T, lmoist, lice, lT)) out.ok = false;

  if (!snow.snow && !INCLUDE_SNOW) {                                     // calc_surf_energy_bal.c:527-546
    if (!P.is_artificial_bare) {
      if (eb.vm.LAI <= 0.0) { vv.throughfall = P.rainfall; out.ppt = vv.throughfall; }
      else out.ppt = vv.throughfall;
    } else out.ppt = P.rainfall;
  }
  e.NetShortGrnd = P.NetShortGrnd;
  if (INCLUDE_SNOW) {
    e.NetLongUnder = NetLongBare + TmpNetLongSnow;
    e.NetShortUnder = P.NetShortBare + P.TmpNetShortSnow + P.NetShortGrnd;
  } else {
    e.NetLongUnder = NetLongBare + P.NetLongSnow;
    e.NetShortUnder = P.NetShortBare + P.NetShortSnow + P.NetShortGrnd;
    e.latent = (P.SnowLatent + e.latent);
    e.latent_sub = (P.SnowLatentSub + e.latent_sub);
    e.sensible = (P.SnowSensible + e.sensible);
  }
  e.LongUnderOut = P.LongUnderIn - e.NetLongUnder;
  e.AlbedoUnder = ((1. - (P.snow_coverage + P.delta_coverage)) * P.BareAlbedo + (P.snow_coverage + P.delta_coverage) * P.SnowAlbedo);
  e.melt_energy = P.melt_energy;
  e.Tsurf = (snow.coverage * snow.surf_temp + (1. - snow.coverage) * Tsurf);

  if (INCLUDE_SNOW) {                                                    // thin snowpack, calc_surf_energy_bal.c:589-679
    if (-(snow.vapor_flux) > snow.swq) {
      snow.blowing_flux *= -(snow.swq / snow.vapor_flux);
      snow.vapor_flux = -(snow.swq);
      snow.surface_flux = snow.vapor_flux - snow.blowing_flux;
    }
    snow.swq += snow.vapor_flux;
    snow.surf_water += snow.vapor_flux;
    snow.surf_water = (snow.surf_water < 0) ? 0. : snow.surf_water;
    if (e.refreeze_energy >= 0.0) {
      double refrozen_water = e.refreeze_energy / (LF * RHO_W) * delta_t;
      if (refrozen_water > snow.surf_water) {
        refrozen_water = snow.surf_water;
        e.refreeze_energy = refrozen_water * LF * RHO_W / delta_t;
      }
      snow.surf_water -= refrozen_water;
      if (snow.surf_water < 0.0) snow.surf_water = 0.0;
      out.melt = 0.0;
    } else {
      out.melt = fabs(e.refreeze_energy) / (LF * RHO_W) * delta_t;
      snow.swq -= out.melt;
      if (snow.swq < 0) { out.melt += snow.swq; snow.swq = 0; }
    }
    if (snow.swq > 0) {
      snow.surf_temp = (Tsurf > 0) ? 0 : Tsurf;
      snow.coldcontent = CH_ICE * snow.surf_temp * snow.swq;
      snow.depth = 1000. * snow.swq / snow.density;
      snow.coverage = 1.;
      if (isnan(snow.surf_temp) || snow.surf_temp > 0) e.snow_flux = (e.grnd_flux + e.deltaH + e.fusion);
    } else {
      snow.density = 0.; snow.depth = 0.; snow.surf_water = 0; snow.pack_water = 0; snow.surf_temp = 0; snow.pack_temp = 0;
      snow.coverage = 0;
    }
    snow.vapor_flux *= -1;
  }
  e.Tsurf_fbflag = sv.fbflag;
  e.Tsurf_fbcount += sv.fbcount;
#pragma unroll
  for (int n = 0; n < NN; n++) { nd.fbflag[n] = (fbmask >> n) & 1u; nd.fbcount[n] += cntnew[n]; }
  out.Tsurf = Tsurf;
  return out;
}

}  // namespace vic
