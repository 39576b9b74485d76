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

// residual of one frozen node (soil_thermal_eqn.c:8-131)
struct SoilThermalEqn {
  double TL, TU, T0, moist, max_moist, bubble, expt, ice0, A, B, C, D, E;
  int EXP_TRANS, node;
  VIC_DEV double operator()(double T) const {
    double ice;
    if (T < 0.) {
      ice = moist - maximum_unfrozen_water(T, max_moist, bubble, expt);
      if (ice < 0.) ice = 0.;
      if (ice > max_moist) ice = max_moist;
    } else ice = 0.;
    double value, flux_term1, flux_term2;
    if (!EXP_TRANS) {
      value = -A * (T - T0) + B * (TL - TU) + C * (TL - T) - D * (T - TU) + E * (ice - ice0);
      flux_term1 = B * (TL - TU);
      flux_term2 = C * (TL - T) - D * (T - TU);
      if (node == 1 && fabs(TL - TU) > 5. && (T < TL && T < TU) && (flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2))
        value = -A * (T - T0) + C * (TL - T) - D * (T - TU) + E * (ice - ice0);
    } else {
      value = -A * (T - T0) + B * (TL - TU) + C * (TL - 2. * T + TU) - D * (TL - TU) + E * (ice - ice0);
      flux_term1 = B * (TL - TU);
      flux_term2 = C * (TL - 2. * T + TU) - D * (TL - TU);
      if (node == 1 && fabs(TL - TU) > 5. && (T < TL && T < TU) && (flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2))
        value = -A * (T - T0) + C * (TL - 2. * T + TU) - D * (TL - TU) + E * (ice - ice0);
    }
    return value;
  }
};

// ------------------------------------------------------------------------------------------------
// Finite-difference soil temperature profile (solve_T_profile + calc_soil_thermal_fluxes, frozen_soil.c:105-225,
// 305-505), laid out for a 64-lane wavefront.
//
// Node columns live in LDS as [node][lane] (ds_read/ds_write_b64 with a per-lane node index are bank-conflict
// free: the two 32-lane halves never collide), so every lane can be at ITS OWN node: the reference's loop nest
//      sweeps (<=1000) { nodes j { closed-form update | Brent (<=1000 residual evaluations with a pow) } }
// is flattened into ONE wave loop in which each lane carries its own (sweep, node, Brent) state and advances by one
// unit of work per trip.  A lane whose node is unfrozen moves to the next node while its neighbours iterate their
// Brent; a lane that has converged waits only for the slowest lane's TOTAL work instead of for the slowest lane at
// every node of every sweep.  Per lane the sequence of operations, and therefore the result, is the reference's.
//
// A-D of the explicit scheme (frozen_soil.c:161-213) depend only on kappa, Cs, the node geometry and dt, i.e. they are
// the same for every residual evaluation of one Brent solve on Tsurf (upstream keeps them in static arrays for that
// reason; SURVEY.md Finding 1.1), so they are built once per calc_surf_energy_bal call.  EI = E*(0-ice) serves the
// closed-form update; E, ice, moist and the freezing-curve parameters of a frozen node are fetched when its Brent
// starts (in compat mode the LAYER arrays max_moist(mm)/bubble/expt followed by the node arrays: Finding 1.2).
// ------------------------------------------------------------------------------------------------
constexpr int PROF_NARR = 7;   // T, T0, A, B, C, D, EI

template <int NN>
constexpr size_t prof_lds_bytes() { return NN > 3 ? (size_t)(PROF_NARR * 8 + 4) * NN * 64 : 0; }

template <int NN>
struct ProfLds {
  double* base;   // [PROF_NARR][NN][64] doubles of this wave
  int* fbc;       // [NN][64] fallback counters of the current solve
  int lane;
  VIC_DEV double& T(int j) const { return base[(0 * NN + j) * 64 + lane]; }
  VIC_DEV double& T0(int j) const { return base[(1 * NN + j) * 64 + lane]; }
  VIC_DEV double& A(int j) const { return base[(2 * NN + j) * 64 + lane]; }
  VIC_DEV double& B(int j) const { return base[(3 * NN + j) * 64 + lane]; }
  VIC_DEV double& C(int j) const { return base[(4 * NN + j) * 64 + lane]; }
  VIC_DEV double& D(int j) const { return base[(5 * NN + j) * 64 + lane]; }
  VIC_DEV double& EI(int j) const { return base[(6 * NN + j) * 64 + lane]; }
  VIC_DEV int& cnt(int j) const { return fbc[j * 64 + lane]; }
};

template <int NN>
VIC_DEV double profile_E(const Opt& o, const CellView& cv, int j, int Nn, double Dp) {
  if (!o.EXP_TRANS) {
    double al = cv.node(CPN_ALPHA, j - 1);
    return ICE_DENSITY * LF * al * al;
  }
  const double Bexp = log(Dp + 1.) / (double)(Nn - 1);
  double z1 = cv.node(CPN_ZSUM, j) + 1;
  return 4 * Bexp * Bexp * ICE_DENSITY * LF * z1 * z1;
}

template <int NN>
VIC_DEV void profile_coefficients(const Opt& o, const CellView& cv, const Nodes<NN>& nd, double deltat, double Dp, const ProfLds<NN>& S) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  const double Bexp = o.EXP_TRANS ? log(Dp + 1.) / (double)(Nn - 1) : 0.0;
#pragma unroll
  for (int j = 0; j < NN; j++) {
    if (j < Nn) S.T0(j) = nd.T[j];
    if (j >= 1 && (j < Nn - 1 || (o.NOFLUX && j == Nn - 1))) {
      const double kup = (j < Nn - 1) ? nd.kappa[(j + 1 < NN) ? j + 1 : j] : nd.kappa[j];
      double E;
      if (!o.EXP_TRANS) {
        double al = cv.node(CPN_ALPHA, j - 1), be = cv.node(CPN_BETA, j - 1), ga = cv.node(CPN_GAMMA, j - 1);
        S.A(j) = nd.Cs[j] * al * al;
        S.B(j) = (kup - nd.kappa[j - 1]) * deltat;
        S.C(j) = 2 * deltat * nd.kappa[j] * al / ga;
        S.D(j) = 2 * deltat * nd.kappa[j] * al / be;
        E = ICE_DENSITY * LF * al * al;
      } else {
        double z1 = cv.node(CPN_ZSUM, j) + 1;
        S.A(j) = 4 * Bexp * Bexp * nd.Cs[j] * z1 * z1;
        S.B(j) = (kup - nd.kappa[j - 1]) * deltat;
        S.C(j) = 4 * deltat * nd.kappa[j];
        S.D(j) = 2 * deltat * nd.kappa[j] * Bexp;
        E = 4 * Bexp * Bexp * ICE_DENSITY * LF * z1 * z1;
      }
      S.EI(j) = E * (0. - nd.ice[j]);
    }
  }
}

// One profile solve; S.T0(0) holds the trial surface temperature.  Results: S.T(j), S.cnt(j), fbmask.
template <int NN>
VIC_DEV bool solve_T_profile(const Opt& o, bool frozen_on, const CellView& cv, const Soil3& s3, const Nodes<NN>& nd,
                             const ProfLds<NN>& S, unsigned& fbmask) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  const int MAXIT = 1000;
  const double threshold = 1.e-2;
  const double Dp = cv.s(CP_DP);
#pragma unroll
  for (int j = 0; j < NN; j++)
    if (j < Nn) { S.T(j) = S.T0(j); S.cnt(j) = 0; }
  fbmask = 0;
  const int jlast = o.NOFLUX ? Nn : Nn - 1;     // exclusive upper node of a sweep
  bool ok = true, converged = false;
  bool done = (jlast <= 1);
  if (done) converged = true;
  int it = 1, j = 1;
  double maxdiff = threshold, oldT = 0;
  bool in_brent = false;
  Brent br;
  SoilThermalEqn eq;
  br.phase = Brent::DONE;
  while (!done) {
    PROF_WAVE(4); PROF_LANE(5);
    bool node_done = false;
    double newT = 0;
    if (!in_brent) {
      oldT = S.T(j);
      const bool bottom = (j == Nn - 1);        // only reached with NOFLUX (frozen_soil.c:423-464)
      const double Tdn = bottom ? oldT : S.T(j + 1), Tup = S.T(j - 1);
      if (oldT >= 0 || !frozen_on) {
        const double A = S.A(j), B = S.B(j), C = S.C(j), D = S.D(j);
        if (!o.EXP_TRANS) newT = (A * S.T0(j) + B * (Tdn - Tup) + C * Tdn + D * Tup + S.EI(j)) / (A + C + D);
        else newT = (A * S.T0(j) + B * (Tdn - Tup) + C * (Tdn + Tup) - D * (Tdn - Tup) + S.EI(j)) / (A + 2. * C);
        node_done = true;
      } else {
        eq.TL = Tdn; eq.TU = Tup; eq.T0 = S.T0(j); eq.moist = nd.moist[j]; eq.ice0 = nd.ice[j];
        eq.A = S.A(j); eq.B = S.B(j); eq.C = S.C(j); eq.D = S.D(j); eq.E = profile_E<NN>(o, cv, j, Nn, Dp);
        eq.EXP_TRANS = o.EXP_TRANS; eq.node = j;
        if (o.frozen_compat) {
          if (j < 3) { eq.max_moist = s3.max_moist[j]; eq.bubble = cv.lay(CPL_BUBBLE, j); eq.expt = cv.lay(CPL_EXPT, j); }
          else { eq.max_moist = cv.node(CPN_MAX_MOIST, j - 3); eq.bubble = cv.node(CPN_BUBBLE, j - 3); eq.expt = cv.node(CPN_EXPT, j - 3); }
        } else { eq.max_moist = cv.node(CPN_MAX_MOIST, j); eq.bubble = cv.node(CPN_BUBBLE, j); eq.expt = cv.node(CPN_EXPT, j); }
        br.start(eq.T0 - SOIL_DT, eq.T0 + SOIL_DT);
        in_brent = true;
      }
    }
    if (in_brent) {
      PROF_LANE(6);
      const double fx = eq(br.x);
      br.advance(fx);
      if (br.phase == Brent::DONE) {
        double r = br.result;
        if (is_error(r)) {
          if (o.TFALLBACK) { r = eq.T0; fbmask |= (1u << j); S.cnt(j) += 1; }
          else { ok = false; done = true; }
        }
        newT = r;
        in_brent = false;
        node_done = true;
      }
    }
    if (node_done) {
      S.T(j) = newT;
      const double diff = fabs(oldT - newT);
      if (diff > maxdiff) maxdiff = diff;
      j++;
      if (j >= jlast) {                           // end of a Gauss-Seidel sweep (frozen_soil.c:466)
        if (maxdiff <= threshold) { converged = true; done = true; }
        else if (it >= MAXIT) done = true;
        else { it++; j = 1; maxdiff = threshold; }
      }
    }
  }
  if (!ok) return false;
  if (o.TFALLBACK) {            // cold-nose hack, frozen_soil.c:470-484 (sic: Tlast[j+1] - T[j]); Tlast == T0
#pragma unroll 1
    for (int k = 1; k < Nn - 1; k++) {
      const double Tk = S.T(k), Tm = S.T(k - 1), Tp = S.T(k + 1), Lk = S.T0(k), Lm = S.T0(k - 1), Lp = S.T0(k + 1);
      if (Lm - Lk > 0 && Lp - Tk > 0 && (Tm - Tk) - (Lm - Lk) > 0 && (Tp - Tk) - (Lp - Lk) > 0) {
        S.T(k) = 0.5 * (Tm + Tp);
        fbmask |= (1u << k);
        S.cnt(k) += 1;
      }
    }
  }
  if (!converged) {
    if (o.TFALLBACK) {
#pragma unroll 1
      for (int k = 0; k < Nn; k++) { S.T(k) = S.T0(k); S.cnt(k) += 1; }
      fbmask |= (Nn >= 32) ? 0xFFFFFFFFu : ((1u << Nn) - 1u);
    } else return false;
  }
  return true;
}

// Residual of the ground-surface energy balance (func_surf_energy_bal.c:9-403)
template <int NN>
struct SurfEB {
  // constant inputs
  const Opt* o; const CellView* cv; const Soil3* s3; const Nodes<NN>* nd;
  ProfLds<NN> S;
  VegMonth vm;
  bool VEG, frozen_on, INCLUDE_SNOW, SNOWING, overstory;
  double delta_t, Cs1, Cs2, D1, D2, T1_old, T2, Ts_old, bubble, dp, expt, ice0, kappa1, kappa2, max_moist, moist, elevation,
         b_infilt, resid0;
  double NetShortBare, NetShortGrnd, NetShortSnow, Tair, atmos_density, atmos_pressure, LongBareIn, LongSnowIn, surf_atten, vp, vpd;
  double Wdew, rainfall, Le, Advection, OldTSurf, kappa_snow, melt_energy, snow_coverage, snow_density, snow_swq, snow_water;
  double U_under, zref_under, disp_under, z0_under, ra_under;
  const double* lmoist; const double* lice; const double* root;
  // state mutated by evaluations ("last evaluation wins")
  double Tsnow_surf;
  double Tnew2;                   // Tnew_node[2] of the last evaluation
  unsigned fbmask;                // T_fbflag bits of the last profile solve
  double ra_used[2];
  VegVar* vv;
  double* layerevap;              // [3]
  double deltaCC, refreeze_energy, vapor_flux, blowing_flux, surface_flux;
  double NetLongBare, NetLongSnow, T1, deltaH, fusion, grnd_flux, latent_heat, latent_heat_sub, sensible_heat, snow_flux, error;

  VIC_DEV double operator()(double Ts) {
    PROF_WAVE(7); PROF_LANE(8);
    const double TMean = Ts;
    const double Tmp = TMean + KELVIN;
    if (snow_coverage > 0 && !INCLUDE_SNOW) snow_flux = (kappa_snow * (Tsnow_surf - TMean));
    else if (INCLUDE_SNOW) { snow_flux = 0; Tsnow_surf = TMean; }
    else snow_flux = 0;
    const double att = (snow_coverage + (1. - snow_coverage) * surf_atten);
    if (o->QUICK_FLUX) {
      T1 = estimate_T1(TMean, T1_old, T2, D1, D2, kappa1, kappa2, Cs2, dp, delta_t);
      if (o->GRND_FLUX_TYPE == VIC_GF_406) grnd_flux = att * (kappa1 / D1 * ((T1) - TMean));
      else grnd_flux = att * (kappa1 / D1 * ((T1) - TMean) + (kappa2 / D2 * (1. - exp(-D1 / dp)) * (T2 - (T1)))) / 2.;
    } else {
      if constexpr (NN > 3) {
        PROF_T0(t_prof);
        S.T0(0) = TMean;                                        // T_node[0] = TMean (func_surf_energy_bal.c:190)
        if (!solve_T_profile<NN>(*o, frozen_on, *cv, *s3, *nd, S, fbmask)) return ERROR_VAL;
        PROF_ADD(4, t_prof);
        T1 = S.T(1);
        Tnew2 = S.T(2);
      }
      if (o->GRND_FLUX_TYPE == VIC_GF_406) grnd_flux = att * (kappa1 / D1 * ((T1) - TMean));
      else grnd_flux = att * (kappa1 / D1 * ((T1) - TMean) + (kappa2 / D2 * (Tnew2 - (T1)))) / 2.;
    }
    if (o->GRND_FLUX_TYPE == VIC_GF_FULL) deltaH = att * (Cs1 * ((Ts_old + T1_old) - (TMean + T1)) * D1 / delta_t / 2.);
    else deltaH = (Cs1 * ((Ts_old + T1_old) - (TMean + T1)) * D1 / delta_t / 2.);
    if (frozen_on) {
      double ice;
      if ((TMean + T1) / 2. < 0.) {
        ice = moist - maximum_unfrozen_water((TMean + T1) / 2., max_moist, bubble, expt);
        if (ice < 0.) ice = 0.;
      } else ice = 0.;
      if (o->GRND_FLUX_TYPE == VIC_GF_FULL) fusion = att * (-ICE_DENSITY * LF * (ice0 - ice) * D1 / delta_t);
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
      Evap = canopy_evap(vm, *s3, lmoist, lice, *vv, true, Wdew, delta_t, NetBareRad, vpd, NetShortBare, Tair, ra_used[1], elevation,
                         rainfall, root, layerevap);
    } else if (!SNOWING) {
      double e0 = layerevap[0];
      Evap = arno_evap(lmoist[0], lice[0], NetBareRad, Tair, vpd, s3->depth[0], max_moist * s3->depth[0] * 1000., elevation, b_infilt,
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
                 + snow_flux * snow_coverage + melt_energy + Advection - deltaCC;
    if (INCLUDE_SNOW) {
      if (Tsnow_surf == 0.0 && err > -(refreeze_energy)) {
        refreeze_energy = -err;
        err = 0.0;
      } else err += refreeze_energy;
    }
    error = err;
    return err;
  }
};

struct SurfOut { double Tsurf, melt, ppt; bool ok; };

// calc_surf_energy_bal (calc_surf_energy_bal.c:7-692).  lmoist/lice/lT/layerevap are the three soil layers of the
// current sub-step (moist in, ice/T/evap out).  melt/ppt are in/out.
template <int NN>
VIC_DEV SurfOut calc_surf_energy_bal(const Opt& o, const CellView& cv, const VegLib& vl, const Soil3& s3, const Forcing& fc, int hidx,
                                     int veg_idx, int month, bool is_artificial_bare, bool overstory, double Le, double LongUnderIn,
                                     double NetLongSnow, double NetShortGrnd, double NetShortSnow, double OldTSurf, double ShortUnderIn,
                                     double SnowAlbedo, double SnowLatent, double SnowLatentSub, double SnowSensible, double Tair,
                                     double VPDcanopy, double VPcanopy, double delta_coverage, double ice0, double melt_energy,
                                     double moist0, double snow_coverage, double snow_depth_avg, double BareAlbedo, double surf_atten,
                                     const Vc& Ra, const Vc& U, const Vc& disp, const Vc& zref, const Vc& z0, double* ra_used,
                                     double melt_in, double ppt_in, double rainfall, const double* root, int INCLUDE_SNOW, int UnderStory,
                                     int dt, const double* lmoist, double* lice, double* lT, double* layerevap, Nodes<NN>& nd,
                                     SoilEnergy& e, Snow& snow, VegVar& vv) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  SurfOut out;
  out.ok = true; out.melt = melt_in; out.ppt = ppt_in;
  const VegMonth vm = veg_month(vl, veg_idx, month);
  const bool frozen_on = (cv.s(CP_FS_ACTIVE) != 0.0) && o.FROZEN_SOIL;
  const double delta_t = (double)dt * 3600.;
  const double Ts_old = nd.T[0];
  const double kappa_snow = (snow.depth > 0.) ? K_SNOW * (snow.density) * (snow.density) / snow_depth_avg : 0;
  const double NetShortBare = (ShortUnderIn * (1. - (snow_coverage + delta_coverage)) * (1. - BareAlbedo)
                               + ShortUnderIn * (delta_coverage) * (1. - SnowAlbedo));
  const double LongBareIn = (1. - snow_coverage) * LongUnderIn;
  double TmpNetLongSnow, TmpNetShortSnow, LongSnowIn;
  if (INCLUDE_SNOW || snow.swq == 0) { TmpNetLongSnow = NetLongSnow; TmpNetShortSnow = NetShortSnow; LongSnowIn = snow_coverage * LongUnderIn; }
  else { TmpNetShortSnow = 0.; TmpNetLongSnow = 0.; LongSnowIn = 0.; }

  SurfEB<NN> eb;
  if constexpr (NN > 3) {
    // one LDS slab per wave (block = one wave): node columns of the profile solver
    extern __shared__ double vic_dyn_lds[];       // prof_lds_bytes<NN>() bytes, sized by the launch
    eb.S.base = vic_dyn_lds; eb.S.fbc = reinterpret_cast<int*>(vic_dyn_lds + PROF_NARR * NN * 64); eb.S.lane = threadIdx.x & 63;
    if (!o.QUICK_FLUX) profile_coefficients<NN>(o, cv, nd, delta_t, cv.s(CP_DP), eb.S);
  } else { eb.S.base = nullptr; eb.S.fbc = nullptr; eb.S.lane = 0; }
  eb.o = &o; eb.cv = &cv; eb.s3 = &s3; eb.nd = &nd; eb.vm = vm;
  eb.VEG = (!is_artificial_bare) && (vm.LAI > 0.0);
  eb.frozen_on = frozen_on; eb.INCLUDE_SNOW = INCLUDE_SNOW != 0; eb.SNOWING = snow.snow != 0; eb.overstory = overstory;
  eb.delta_t = delta_t; eb.Cs1 = e.Cs[0]; eb.Cs2 = e.Cs[1];
  eb.D1 = cv.node(CPN_ZSUM, 1) - cv.node(CPN_ZSUM, 0); eb.D2 = cv.node(CPN_ZSUM, 2) - cv.node(CPN_ZSUM, 1);
  eb.T1_old = nd.T[1]; eb.T2 = nd.T[Nn - 1 < NN ? Nn - 1 : NN - 1]; eb.Ts_old = Ts_old;
  eb.bubble = cv.lay(CPL_BUBBLE, 0); eb.dp = cv.s(CP_DP); eb.expt = cv.lay(CPL_EXPT, 0); eb.ice0 = ice0;
  eb.kappa1 = e.kappa[0]; eb.kappa2 = e.kappa[1];
  eb.max_moist = s3.max_moist[0] / (s3.depth[0] * 1000.); eb.moist = moist0;
  eb.elevation = cv.s(CP_ELEVATION); eb.b_infilt = cv.s(CP_B_INFILT); eb.resid0 = s3.resid_moist[0];
  eb.NetShortBare = NetShortBare; eb.NetShortGrnd = NetShortGrnd; eb.NetShortSnow = TmpNetShortSnow; eb.Tair = Tair;
  eb.atmos_density = fc.v(VIC_F_DENSITY, hidx); eb.atmos_pressure = fc.v(VIC_F_PRESSURE, hidx);
  eb.LongBareIn = LongBareIn; eb.LongSnowIn = LongSnowIn; eb.surf_atten = surf_atten; eb.vp = VPcanopy; eb.vpd = VPDcanopy;
  eb.Wdew = vv.Wdew; eb.rainfall = rainfall; eb.Le = Le; eb.Advection = e.advection; eb.OldTSurf = OldTSurf;
  eb.kappa_snow = kappa_snow; eb.melt_energy = melt_energy; eb.snow_coverage = snow_coverage; eb.snow_density = snow.density;
  eb.snow_swq = snow.swq; eb.snow_water = snow.surf_water;
  eb.U_under = U.v[UnderStory]; eb.zref_under = zref.v[UnderStory]; eb.disp_under = disp.v[UnderStory];
  eb.z0_under = z0.v[UnderStory]; eb.ra_under = Ra.v[UnderStory];
  eb.lmoist = lmoist; eb.lice = lice; eb.root = root;
  eb.Tsnow_surf = snow.surf_temp;
  eb.Tnew2 = 0; eb.fbmask = 0;
  eb.ra_used[0] = ra_used[0]; eb.ra_used[1] = ra_used[1];
  eb.vv = &vv; eb.layerevap = layerevap;
  eb.deltaCC = e.deltaCC; eb.refreeze_energy = e.refreeze_energy; eb.vapor_flux = snow.vapor_flux;
  eb.blowing_flux = snow.blowing_flux; eb.surface_flux = snow.surface_flux;
  eb.NetLongBare = 0; eb.NetLongSnow = TmpNetLongSnow; eb.T1 = 0; eb.deltaH = e.deltaH; eb.fusion = e.fusion;
  eb.grnd_flux = e.grnd_flux; eb.latent_heat = e.latent; eb.latent_heat_sub = e.latent_sub; eb.sensible_heat = e.sensible;
  eb.snow_flux = e.snow_flux; eb.error = e.error;

  double Tsurf;
  int Tsurf_fbflag = 0, Tsurf_fbcount = 0;
  if (o.FULL_ENERGY) {
    double T_lower, T_upper;
    if (INCLUDE_SNOW) { T_lower = nd.T[0] - SURF_DT; T_upper = 0.; }
    else { T_lower = 0.5 * (nd.T[0] + Tair) - SURF_DT; T_upper = 0.5 * (nd.T[0] + Tair) + SURF_DT; }
    Tsurf = root_brent(T_lower, T_upper, eb);
    if (is_error(Tsurf)) {
      if (o.TFALLBACK) { Tsurf = Ts_old; Tsurf_fbflag = 1; Tsurf_fbcount++; }
      else out.ok = false;
    }
  } else Tsurf = Tair;

  eb.Tsnow_surf = snow.surf_temp;        // the final evaluation runs on a fresh object (calc_surf_energy_bal.c:489-506)
  double error = eb(Tsurf);
  if (error == ERROR_VAL) out.ok = false;
  e.error = error;
  e.deltaCC = eb.deltaCC; e.refreeze_energy = eb.refreeze_energy; e.deltaH = eb.deltaH; e.fusion = eb.fusion;
  e.grnd_flux = eb.grnd_flux; e.latent = eb.latent_heat; e.latent_sub = eb.latent_heat_sub; e.sensible = eb.sensible_heat;
  e.snow_flux = eb.snow_flux;
  snow.vapor_flux = eb.vapor_flux; snow.blowing_flux = eb.blowing_flux; snow.surface_flux = eb.surface_flux;
  ra_used[0] = eb.ra_used[0]; ra_used[1] = eb.ra_used[1];
  TmpNetLongSnow = eb.NetLongSnow;
  const double NetLongBare = eb.NetLongBare;

  double Tnew[NN];
  int cntnew[NN];
#pragma unroll
  for (int n = 0; n < NN; n++) { Tnew[n] = 0; cntnew[n] = 0; }
  if constexpr (NN > 3) {
    if (!o.QUICK_FLUX) {
#pragma unroll
      for (int n = 0; n < NN; n++)
        if (n < Nn) { Tnew[n] = eb.S.T(n); cntnew[n] = eb.S.cnt(n); }
    }
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
    if (!is_artificial_bare) {
      if (vm.LAI <= 0.0) { vv.throughfall = rainfall; out.ppt = vv.throughfall; }
      else out.ppt = vv.throughfall;
    } else out.ppt = rainfall;
  }
  e.NetShortGrnd = NetShortGrnd;
  if (INCLUDE_SNOW) {
    e.NetLongUnder = NetLongBare + TmpNetLongSnow;
    e.NetShortUnder = NetShortBare + TmpNetShortSnow + NetShortGrnd;
  } else {
    e.NetLongUnder = NetLongBare + NetLongSnow;
    e.NetShortUnder = NetShortBare + NetShortSnow + NetShortGrnd;
    e.latent = (SnowLatent + e.latent);
    e.latent_sub = (SnowLatentSub + e.latent_sub);
    e.sensible = (SnowSensible + e.sensible);
  }
  e.LongUnderOut = LongUnderIn - e.NetLongUnder;
  e.AlbedoUnder = ((1. - (snow_coverage + delta_coverage)) * BareAlbedo + (snow_coverage + delta_coverage) * SnowAlbedo);
  e.melt_energy = melt_energy;
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
  e.Tsurf_fbflag = Tsurf_fbflag;
  e.Tsurf_fbcount += Tsurf_fbcount;
#pragma unroll
  for (int n = 0; n < NN; n++) { nd.fbflag[n] = (eb.fbmask >> n) & 1u; nd.fbcount[n] += cntnew[n]; }
  out.Tsurf = Tsurf;
  return out;
}

}  // namespace vic
