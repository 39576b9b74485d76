// vic_soil.hpp — soil thermal bookkeeping, runoff/baseflow and evapotranspiration (device only, gfx950).
#pragma once
#include "vic_math.hpp"

namespace vic {

// compute_soil_layer_thermal_properties (soil_conduction.c:725-773) for the top two layers, which is all
// prepare_full_energy.c:83-89 keeps
VIC_DEV void top_layer_thermal_properties(const CellView& cv, const Soil3& s3, const double* moist, const double* ice,
                                          double* kappa2, double* Cs2) {
#pragma unroll
  for (int l = 0; l < 2; l++) {
    double m = moist[l] / s3.depth[l] / 1000;
    double ic = ice[l] / s3.depth[l] / 1000;
    double bd = cv.lay(CPL_BULK_DENSITY, l), sd = cv.lay(CPL_SOIL_DENSITY, l), org = cv.lay(CPL_ORGANIC, l);
    const SoilKLayer kc{cv.x(CPX_KDRY, l), cv.x(CPX_KSP, l), cv.x(CPX_KWP, l), cv.x(CPX_POROSITY, l)};
    kappa2[l] = soil_conductivity_pre(m, m - ic, kc);
    Cs2[l] = volumetric_heat_capacity(bd / sd, m - ic, ic, org);
  }
}

// distribute_node_moisture_properties (soil_conduction.c:304-440)
template <int NN>
VIC_DEV void distribute_node_moisture_properties(const Opt& o, const CellView& cv, const Soil3& s3, Nodes<NN>& nd,
                                                 const double* moist) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  int l = 0;
  bool past_bottom = false;
  double Lsum = 0.;
  const bool fs = (cv.s(CP_FS_ACTIVE) != 0.0) && o.FROZEN_SOIL;
  // per-layer constants of the loop body, selected by the run-time layer index below
  double lbd[3], lsd[3], lorg[3], lKdry[3], lKsP[3], lKwP[3], lpor[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    lbd[q] = cv.lay(CPL_BULK_DENSITY, q); lsd[q] = cv.lay(CPL_SOIL_DENSITY, q); lorg[q] = cv.lay(CPL_ORGANIC, q);
    lKdry[q] = cv.x(CPX_KDRY, q); lKsP[q] = cv.x(CPX_KSP, q); lKwP[q] = cv.x(CPX_KWP, q); lpor[q] = cv.x(CPX_POROSITY, q);
  }
#pragma unroll
  for (int n = 0; n < NN; n++) {
    if (n < Nn) {
      double z = cv.node(CPN_ZSUM, n), mmn = cv.node(CPN_MAX_MOIST, n);
      // l is a run-time layer index: select instead of indexing (a dynamic index would pin s3 / the HRU struct to memory)
      double dl = sel3(s3.depth, l);
      if (z == Lsum + dl && n != 0 && l != 2) nd.moist[n] = (sel3(moist, l) / dl + sel3(moist, l + 1) / sel3(s3.depth, l + 1)) / 1000 / 2.;
      else nd.moist[n] = sel3(moist, l) / dl / 1000;
      if (nd.moist[n] - mmn > 0) nd.moist[n] = mmn;
      const double bd = sel3(lbd, l), sd = sel3(lsd, l), org = sel3(lorg, l);
      const SoilKLayer kc{sel3(lKdry, l), sel3(lKsP, l), sel3(lKwP, l), sel3(lpor, l)};
      if (nd.T[n] < 0 && fs) {
        nd.ice[n] = nd.moist[n] - maximum_unfrozen_water(nd.T[n], mmn, cv.node(CPN_BUBBLE, n), cv.node(CPN_EXPT, n));
        if (nd.ice[n] < 0) nd.ice[n] = 0;
        nd.kappa[n] = soil_conductivity_pre(nd.moist[n], nd.moist[n] - nd.ice[n], kc);
      } else {
        nd.ice[n] = 0;
        nd.kappa[n] = soil_conductivity_pre(nd.moist[n], nd.moist[n], kc);
      }
      nd.Cs[n] = volumetric_heat_capacity(bd / sd, nd.moist[n] - nd.ice[n], nd.ice[n], org);
      if (z > Lsum + dl && !past_bottom) {
        Lsum += dl;
        l++;
        if (l == 3) { past_bottom = true; l = 2; }
      }
    }
  }
}

// estimate_layer_ice_content (soil_conduction.c:444-614), one frost sub-area.  Returns false when the thermal nodes do
// not reach below the bottom soil layer (:526-529).
template <int NN>
VIC_DEV bool estimate_layer_ice_content(const Opt& o, const CellView& cv, const Soil3& s3, const double* T,
                                        const double* moist, double* layer_ice, double* layer_T) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  // everything the layer / node walk indexes at run time is copied into small local arrays first: dynamic indexing
  // through the argument pointers would pin the caller's whole HRU struct to scratch memory
  double Lsum[4], Z[NN], Tl[NN], ml[3], mml[3], dl[3], bubl[3], exl[3], outI[3], outT[3];
  Lsum[0] = 0;
#pragma unroll
  for (int l = 1; l <= 3; l++) Lsum[l] = s3.depth[l - 1] + Lsum[l - 1];
#pragma unroll
  for (int n = 0; n < NN; n++) { Z[n] = (n < Nn) ? cv.node(CPN_ZSUM, n) : 0.0; Tl[n] = T[n]; }
#pragma unroll
  for (int l = 0; l < 3; l++) {
    ml[l] = moist[l]; mml[l] = s3.max_moist[l]; dl[l] = s3.depth[l]; bubl[l] = cv.lay(CPL_BUBBLE, l); exl[l] = cv.lay(CPL_EXPT, l);
    outI[l] = 0; outT[l] = 0;
  }
  const bool fs = o.FROZEN_SOIL && (cv.s(CP_FS_ACTIVE) != 0.0);
  int lfail = 3;
#pragma unroll 1
  for (int l = 0; l < 3; l++) {
    double accT = 0., accI = 0.;
    int min_n = Nn - 2;
    while (Lsum[l] < Z[min_n] && min_n > 0) min_n--;
    int max_n = 1;
    while (max_n < Nn && Lsum[l + 1] > Z[max_n]) max_n++;
    if (max_n >= Nn) { lfail = l; break; }      // the failing layer keeps the zeros of :512-519, later layers are not touched
    const double mm = mml[l], bub = bubl[l], ex = exl[l];
    // walk the bracketing nodes once, carrying (z, T, ice) of the previous point: the trapezoid sums of :591-600
    double pz = 0, pT = 0, pI = 0;
    for (int n = min_n; n <= max_n; n++) {
      double tz, tT;
      if (n == min_n) {
        tT = (Z[min_n] < Lsum[l]) ? linear_interp(Lsum[l], Z[min_n], Z[min_n + 1], Tl[min_n], Tl[min_n + 1]) : Tl[min_n];
        tz = Lsum[l];
      } else if (n == max_n) {
        tT = (Z[max_n] > Lsum[l + 1]) ? linear_interp(Lsum[l + 1], Z[max_n - 1], Z[max_n], Tl[max_n - 1], Tl[max_n]) : Tl[max_n];
        tz = Lsum[l + 1];
      } else { tT = Tl[n]; tz = Z[n]; }
      double tI = 0;
      if (fs) {
        tI = ml[l] - maximum_unfrozen_water(tT, mm, bub, ex);
        if (tI < 0) tI = 0.;
      }
      if (n > min_n) {
        accI += (tz - pz) * (tI + pI) / 2.;
        accT += (tz - pz) * (tT + pT) / 2.;
      }
      pz = tz; pT = tT; pI = tI;
    }
    outI[l] = accI / dl[l];
    outT[l] = accT / dl[l];
  }
#pragma unroll
  for (int l = 0; l < 3; l++)
    if (l <= lfail) { layer_ice[l] = outI[l]; layer_T[l] = outT[l]; }
  return lfail == 3;
}

// estimate_layer_ice_content_quick_flux (soil_conduction.c:617-723)
VIC_DEV void estimate_layer_ice_content_quick_flux(const Opt& o, const CellView& cv, const Soil3& s3, double Tsurf, double T1,
                                                   const double* moist, double* layer_ice, double* layer_T) {
  double Lsum[4];
  Lsum[0] = 0;
#pragma unroll
  for (int l = 1; l <= 3; l++) Lsum[l] = s3.depth[l - 1] + Lsum[l - 1];
  const double avg_temp = cv.s(CP_AVG_TEMP), dp = cv.s(CP_DP);
  layer_T[0] = 0.5 * (Tsurf + T1);
#pragma unroll
  for (int l = 1; l < 3; l++)
    layer_T[l] = avg_temp - dp / (s3.depth[l]) * (T1 - avg_temp) * (exp(-(Lsum[l + 1] - Lsum[1]) / dp) - exp(-(Lsum[l] - Lsum[1]) / dp));
  const bool fs = o.FROZEN_SOIL && (cv.s(CP_FS_ACTIVE) != 0.0);
#pragma unroll
  for (int l = 0; l < 3; l++) {
    layer_ice[l] = 0;
    if (fs) {
      layer_ice[l] = moist[l] - maximum_unfrozen_water(layer_T[l], s3.max_moist[l], cv.lay(CPL_BUBBLE, l), cv.lay(CPL_EXPT, l));
      if (layer_ice[l] < 0) layer_ice[l] = 0;
      if (layer_ice[l] > moist[l]) layer_ice[l] = moist[l];
    }
  }
}

// find_0_degree_fronts (soil_conduction.c:775-828)
template <int NN>
VIC_DEV void find_0_degree_fronts(const Opt& o, const CellView& cv, SoilEnergy& e, const double* T) {
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  int Nthaw = 0, Nfrost = 0;
  double td[3] = {NAN, NAN, NAN}, fd[3] = {NAN, NAN, NAN};
  for (int n = Nn - 2; n >= 0; n--) {
    if (T[n] > 0 && T[n + 1] <= 0 && Nthaw < 3) {
      double v = linear_interp(0, T[n], T[n + 1], cv.node(CPN_ZSUM, n), cv.node(CPN_ZSUM, n + 1));
      if (Nthaw == 0) td[0] = v; else if (Nthaw == 1) td[1] = v; else td[2] = v;
      Nthaw++;
    } else if (T[n] < 0 && T[n + 1] >= 0 && Nfrost < 3) {
      double v = linear_interp(0, T[n], T[n + 1], cv.node(CPN_ZSUM, n), cv.node(CPN_ZSUM, n + 1));
      if (Nfrost == 0) fd[0] = v; else if (Nfrost == 1) fd[1] = v; else fd[2] = v;
      Nfrost++;
    }
  }
#pragma unroll
  for (int f = 0; f < 3; f++) { e.tdepth[f] = td[f]; e.fdepth[f] = fd[f]; }
  e.Nthaw = Nthaw;
  e.Nfrost = Nfrost;
}

// ------------------------------------------------------------------------------------------------ water table (compute_zwt.c)
VIC_DEV double compute_zwt(const CellView& cv, int l, double moist) {
  double zwt = NAN;
  int i = VIC_MAX_ZWTVMOIST - 1;
  while (i >= 1 && moist > cv.zwt_moist(l, i)) i--;
  if (i == VIC_MAX_ZWTVMOIST - 1) {
    double mi = cv.zwt_moist(l, i);
    if (moist < mi) zwt = NAN;
    else if (moist == mi) zwt = cv.zwt_zwt(l, i);
  } else {
    double z1 = cv.zwt_zwt(l, i + 1), z0 = cv.zwt_zwt(l, i), m1 = cv.zwt_moist(l, i + 1), m0 = cv.zwt_moist(l, i);
    zwt = z1 + (z0 - z1) * (moist - m1) / (m0 - m1);
  }
  return zwt;
}

struct Zwt { double zwt, zwt2, zwt3, lz[3]; };   // lz: layer[l].zwt
// wrap_compute_zwt (compute_zwt.c:49-112)
VIC_DEV Zwt wrap_compute_zwt(const CellView& cv, const Soil3& s3, const double* moist) {
  Zwt r;
  double lz[3];
  double total_depth = 0;
#pragma unroll
  for (int l = 0; l < 3; l++) total_depth += s3.depth[l];
#pragma unroll
  for (int l = 0; l < 3; l++) lz[l] = compute_zwt(cv, l, moist[l]);
  if (isnan(lz[2])) lz[2] = -total_depth * 100;
  int l = 2;
  double tmp_depth = total_depth;
  // while (l >= 0 && max_moist[l] - moist[l] <= SMALL) { tmp_depth -= depth[l]; l--; } with static indices
#pragma unroll
  for (int k = 2; k >= 0; k--)
    if (l == k && s3.max_moist[k] - moist[k] <= SMALL) { tmp_depth -= s3.depth[k]; l--; }
  if (l < 0) r.zwt = 0;
  else if (l < 2) {
    double z = (l == 0) ? lz[0] : lz[1];
    if (!isnan(z)) r.zwt = z; else r.zwt = -tmp_depth * 100;
  } else r.zwt = lz[2];
  double tm = moist[0] + moist[1];
  r.zwt2 = compute_zwt(cv, 3, tm);
  if (isnan(r.zwt2)) r.zwt2 = lz[2];
  tm = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) tm += moist[k];
  r.zwt3 = compute_zwt(cv, 4, tm);
  if (isnan(r.zwt3)) r.zwt3 = -total_depth * 100;
#pragma unroll
  for (int k = 0; k < 3; k++) r.lz[k] = lz[k];
  return r;
}

// ------------------------------------------------------------------------------------------------ runoff.c
// compute_runoff_and_asat (runoff.c:773-813)
VIC_DEV void compute_runoff_and_asat(const Soil3& s3, double b_infilt, const double* moist, double inflow, double& A, double& runoff) {
  double top_moist = moist[0] + moist[1];
  double top_max_moist = s3.max_moist[0] + s3.max_moist[1];
  if (top_moist > top_max_moist) top_moist = top_max_moist;
  double ex = b_infilt / (1.0 + b_infilt);
  A = 1.0 - pow((1.0 - top_moist / top_max_moist), ex);
  double max_infil = (1.0 + b_infilt) * top_max_moist;
  double i_0 = max_infil * (1.0 - pow((1.0 - A), (1.0 / b_infilt)));
  if (inflow == 0.0) runoff = 0.0;
  else if (max_infil == 0.0) runoff = inflow;
  else if ((i_0 + inflow) > max_infil) runoff = inflow - top_max_moist + top_moist;
  else {
    double basis = 1.0 - (i_0 + inflow) / max_infil;
    runoff = (inflow - top_max_moist + top_moist + top_max_moist * pow(basis, 1.0 * (1.0 + b_infilt)));
  }
  if (runoff < 0.) runoff = 0.;
}

struct RunoffOut { double runoff, baseflow, asat; };

// runoff (runoff.c:7-771), Ndist 1, one frost area.  moist[] in/out (mm), ice[] in, evap[] in/out (mm/step).
VIC_DEV RunoffOut runoff_step(const Opt& o, const CellView& cv, const Soil3& s3, double* moist, const double* ice_in, double* evap_io,
                              double ppt) {
  const int dt = o.dt;
  double resid[3], liq[3], ice[3], maxm[3], Ksat[3], expt[2], Q12[2], evap[3], tmpm[3];
  const double b_infilt = cv.s(CP_B_INFILT);
  const double Dsmax = cv.s(CP_DSMAX) / 24., Ds = cv.s(CP_DS), Ws = cv.s(CP_WS), cexp = cv.s(CP_C);
  double A, runoff, baseflow = 0, tmp_runoff;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    resid[l] = s3.resid_moist[l] * s3.depth[l] * 1000.;
    evap[l] = evap_io[l] / (double)dt;
    Ksat[l] = cv.lay(CPL_KSAT, l) / 24.;
    liq[l] = moist[l] - ice_in[l];
    ice[l] = ice_in[l];
    maxm[l] = s3.max_moist[l];
    tmpm[l] = liq[l] + ice[l];
  }
  expt[0] = cv.lay(CPL_EXPT, 0); expt[1] = cv.lay(CPL_EXPT, 1);
  double inflow = ppt;
  compute_runoff_and_asat(s3, b_infilt, tmpm, inflow, A, runoff);
  const double tmp_dt_runoff = runoff / (double)dt;
  const double dt_inflow = inflow / (double)dt;
  for (int ts = 0; ts < dt; ts++) {                                      // hourly sub-steps, runoff.c:451-700
    inflow = dt_inflow;
#pragma unroll
    for (int l = 0; l < 2; l++) {
      double tmp_liq = liq[l] - evap[l];
      if (tmp_liq < resid[l]) tmp_liq = resid[l];
      if (liq[l] > resid[l]) Q12[l] = Ksat[l] * pow(((tmp_liq - resid[l]) / (maxm[l] - resid[l])), expt[l]);
      else Q12[l] = 0.;
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
      double dt_runoff = (l == 0) ? tmp_dt_runoff : 0;
      double tmp_inflow = 0.;
      liq[l] = liq[l] + (inflow - dt_runoff) - (Q12[l] + evap[l]);
      if ((liq[l] + ice[l]) > maxm[l]) {
        tmp_inflow = (liq[l] + ice[l]) - maxm[l];
        liq[l] = maxm[l] - ice[l];
        if (l == 0) { Q12[l] += tmp_inflow; tmp_inflow = 0; }
        else {
          // spill upward into layer 0, then to runoff (the while loop of runoff.c:571-595 for l == 1)
          liq[0] += tmp_inflow;
          if ((liq[0] + ice[0]) > maxm[0]) {
            tmp_inflow = ((liq[0] + ice[0]) - maxm[0]);
            liq[0] = maxm[0] - ice[0];
          } else tmp_inflow = 0;
          if (tmp_inflow > 0) { runoff += tmp_inflow; tmp_inflow = 0; }
        }
      }
      if ((liq[l] + ice[l]) < resid[l]) {
        Q12[l] += (liq[l] + ice[l]) - resid[l];
        liq[l] = resid[l] - ice[l];
      }
      inflow = (Q12[l] + tmp_inflow);
      Q12[l] += tmp_inflow;
    }
    // ARNO baseflow, runoff.c:622-698
    double rel_moist = (liq[2] - resid[2]) / (maxm[2] - resid[2]);
    double frac = Dsmax * Ds / Ws;
    double dt_baseflow = frac * rel_moist;
    if (rel_moist > Ws) {
      frac = (rel_moist - Ws) / (1 - Ws);
      dt_baseflow += Dsmax * (1 - Ds / Ws) * pow(frac, cexp);
    }
    if (dt_baseflow < 0) dt_baseflow = 0;
    liq[2] += Q12[1] - (evap[2] + dt_baseflow);
    if ((liq[2] + ice[2]) < resid[2]) {
      dt_baseflow += (liq[2] + ice[2]) - resid[2];
      liq[2] = resid[2] - ice[2];
    }
    if ((liq[2] + ice[2]) > maxm[2]) {
      double tmp_moist = ((liq[2] + ice[2]) - maxm[2]);
      liq[2] = maxm[2] - ice[2];
#pragma unroll
      for (int tl = 1; tl >= 0; tl--) {
        if (tmp_moist > 0) {
          liq[tl] += tmp_moist;
          if ((liq[tl] + ice[tl]) > maxm[tl]) {
            tmp_moist = ((liq[tl] + ice[tl]) - maxm[tl]);
            liq[tl] = maxm[tl] - ice[tl];
          } else tmp_moist = 0;
        }
      }
      if (tmp_moist > 0) { runoff += tmp_moist; tmp_moist = 0; }
    }
    baseflow += dt_baseflow;
  }
  if (baseflow < 0) {            // runoff.c:707-710: negative baseflow comes out of the bottom layer's evap
    evap_io[2] += baseflow;
    baseflow = 0;
  }
#pragma unroll
  for (int l = 0; l < 3; l++) tmpm[l] = liq[l] + ice[l];
  compute_runoff_and_asat(s3, b_infilt, tmpm, 0, A, tmp_runoff);
#pragma unroll
  for (int l = 0; l < 3; l++) moist[l] = liq[l] + ice[l];
  RunoffOut r;
  r.asat = A; r.runoff = runoff; r.baseflow = baseflow;
  return r;
}

// ------------------------------------------------------------------------------------------------ evapotranspiration

struct VegMonth { double LAI, Wdmax, rmin, rarc; float RGL; };
VIC_DEV VegMonth veg_month(const VegLib& vl, int idx, int month) {
  VegMonth v;
  v.LAI = vl.f(idx, VL_LAI + month - 1); v.Wdmax = vl.f(idx, VL_WDMAX + month - 1);
  v.rmin = vl.f(idx, VL_RMIN); v.rarc = vl.f(idx, VL_RARC); v.RGL = (float)vl.f(idx, VL_RGL);
  return v;
}

// transpiration (canopy_evap.c:218-442)
VIC_DEV void transpiration(const VegMonth& vm, const Soil3& s3, const double* moist, const double* ice, const PenmanBase& pb,
                           double rad, double vpd, double net_short, double air_temp, double ra, double f, double delta_t,
                           double Wdew, const double* root, double* layerevap) {
  double avail[3], moist1 = 0.0, Wcr1 = 0.0, gsm_inv, rc, evap;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if (root[i] > 0.) { avail[i] = moist[i] - ice[i]; moist1 += avail[i]; Wcr1 += s3.Wcr[i]; }
    else avail[i] = 0.;
  }
  double moist2 = moist[2] - ice[2];
  avail[2] = moist2;
  const double wet = (1.0 - f * pow((Wdew / vm.Wdmax), (2.0 / 3.0)));
  if ((moist1 >= Wcr1 && moist2 >= s3.Wcr[2] && Wcr1 > 0.) || (moist1 >= Wcr1 && (1 - root[2]) >= 0.5)
      || (moist2 >= s3.Wcr[2] && root[2] >= 0.5)) {
    gsm_inv = 1.0;
    rc = calc_rc(vm.rmin, net_short, vm.RGL, air_temp, vpd, vm.LAI, gsm_inv, false);
    evap = penman_eval(pb, rad, vpd, ra, rc, vm.rarc) * delta_t / SEC_PER_DAY * wet;
    double root_sum = 1.0, spare_evap = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (avail[i] >= s3.Wcr[i]) layerevap[i] = evap * root[i];
      else {
        if (avail[i] >= s3.Wpwp[i]) gsm_inv = (avail[i] - s3.Wpwp[i]) / (s3.Wcr[i] - s3.Wpwp[i]);
        else gsm_inv = 0.0;
        layerevap[i] = evap * gsm_inv * root[i];
        root_sum -= root[i];
        spare_evap = evap * root[i] * (1.0 - gsm_inv);
      }
    }
    if (spare_evap > 0.0) {
#pragma unroll
      for (int i = 0; i < 3; i++)
        if (avail[i] >= s3.Wcr[i]) layerevap[i] += root[i] * spare_evap / root_sum;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (avail[i] >= s3.Wcr[i]) gsm_inv = 1.0;
      else if (avail[i] >= s3.Wpwp[i]) gsm_inv = (avail[i] - s3.Wpwp[i]) / (s3.Wcr[i] - s3.Wpwp[i]);
      else gsm_inv = 0.0;
      if (gsm_inv > 0.0) {
        rc = calc_rc(vm.rmin, net_short, vm.RGL, air_temp, vpd, vm.LAI, gsm_inv, false);
        layerevap[i] = penman_eval(pb, rad, vpd, ra, rc, vm.rarc) * delta_t / SEC_PER_DAY * root[i] * wet;
      } else layerevap[i] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (ice[i] > 0) {
      if (ice[i] >= s3.Wpwp[i]) { if (layerevap[i] > avail[i]) layerevap[i] = avail[i]; }
      else { if (layerevap[i] > moist[i] - s3.Wpwp[i]) layerevap[i] = moist[i] - s3.Wpwp[i]; }
    } else {
      if (layerevap[i] > moist[i] - s3.Wpwp[i]) layerevap[i] = moist[i] - s3.Wpwp[i];
    }
    if (layerevap[i] < 0.0) layerevap[i] = 0.0;
  }
}

// canopy_evap (canopy_evap.c:46-212), Ndist 1 / mu 1.  Writes layerevap[3] (zeros unless calc_evap) and vv.
VIC_DEV double canopy_evap(const VegMonth& vm, const Soil3& s3, const double* moist, const double* ice, VegVar& vv, bool calc_evap,
                           double Wdew_in, double delta_t, double rad, double vpd, double net_short, double air_temp, double ra,
                           double elevation, double ppt, const double* root, double* layerevap) {
  double throughfall = 0, tmp_Wdew = Wdew_in, f;
  const double Wdew_org = tmp_Wdew;
  const PenmanBase pb = penman_base(air_temp, elevation);
  if (tmp_Wdew > vm.Wdmax) { throughfall = tmp_Wdew - vm.Wdmax; tmp_Wdew = vm.Wdmax; }
  double rc = calc_rc(0.0, net_short, vm.RGL, air_temp, vpd, vm.LAI, 1.0, false);
  double canopyevap = pow((tmp_Wdew / vm.Wdmax), (2.0 / 3.0)) * penman_eval(pb, rad, vpd, ra, rc, vm.rarc) * delta_t / SEC_PER_DAY;
  if (canopyevap > 0.0 && delta_t == SEC_PER_DAY) f = fmin(1.0, ((tmp_Wdew + ppt) / canopyevap));
  else if (canopyevap > 0.0) f = fmin(1.0, ((tmp_Wdew) / canopyevap));
  else f = 1.0;
  canopyevap *= f;
  tmp_Wdew += ppt - canopyevap;
  if (tmp_Wdew < 0.0) tmp_Wdew = 0.0;
  if (tmp_Wdew > vm.Wdmax) { throughfall += tmp_Wdew - vm.Wdmax; tmp_Wdew = vm.Wdmax; }
  layerevap[0] = layerevap[1] = layerevap[2] = 0;
  if (calc_evap)
    transpiration(vm, s3, moist, ice, pb, rad, vpd, net_short, air_temp, ra, f, delta_t, Wdew_org, root, layerevap);
  vv.canopyevap = canopyevap;
  vv.throughfall = throughfall;
  vv.Wdew = tmp_Wdew;
  double tmp_Evap = canopyevap;
#pragma unroll
  for (int i = 0; i < 3; i++) tmp_Evap += layerevap[i];
  return tmp_Evap * 1.0 / (1000. * delta_t);
}

// arno_evap (arno_evap.c:61-228), Ndist 1 / mu 1.  Returns Evap (m/s) or ERROR_VAL; writes evap0 (mm/step).
VIC_DEV double arno_evap(double moist0, double ice0, double rad, double air_temp, double vpd, double depth1, double max_moist,
                         double elevation, double b_infilt, double ra, double delta_t, double moist_resid, double& evap0) {
  double moist = moist0 - ice0;
  if (moist > max_moist) moist = max_moist;
  double Epot = penman(air_temp, elevation, rad, vpd, ra, 0.0, 0.0) * delta_t / SEC_PER_DAY;
  double max_infil = (1.0 + b_infilt) * max_moist;
  double tmp, ratio, evap;
  if (b_infilt == -1.0) tmp = max_infil;
  else {
    ratio = 1.0 - (moist) / (max_moist);
    if (ratio > 1.0 || ratio < 0.0) return ERROR_VAL;
    ratio = pow(ratio, (1.0 / (b_infilt + 1.0)));
    tmp = max_infil * (1.0 - ratio);
  }
  if (tmp >= max_infil) evap = Epot;
  else {
    ratio = tmp / max_infil;
    ratio = 1.0 - ratio;
    if (ratio > 1.0 || ratio < 0.0) return ERROR_VAL;
    if (ratio != 0.0) ratio = pow(ratio, b_infilt);
    double as = 1 - ratio;
    ratio = pow(ratio, (1.0 / b_infilt));
    // 30-term power series of the ARNO beta function (arno_evap.c:185-193); the reference rebuilds ratio^n with an
    // inner multiply loop, which is the same product chain as the running power below
    double dummy = 1.0, pw = 1.0;
    for (int num_term = 1; num_term <= 30; num_term++) {
      pw = (num_term == 1) ? ratio : pw * ratio;
      dummy += b_infilt * pw / (b_infilt + num_term);
    }
    double beta_asp = as + (1.0 - as) * (1.0 - ratio) * dummy;
    evap = Epot * beta_asp;
  }
  if (evap > 0.0) {
    if (moist > moist_resid * depth1 * 1000.) {
      if (evap > moist - moist_resid * depth1 * 1000.) evap = moist - moist_resid * depth1 * 1000.;
    } else evap = 0.0;
  }
  evap0 = evap;
  return evap / 1000. / delta_t * 1.0;
}

// compute_pot_evap (compute_pot_evap.c:8-78) including the stale net_short of SURVEY.md Appendix C #4
VIC_DEV void compute_pot_evap(const Opt& o, const VegLib& vl, int veg_idx, int month, double shortwave, double net_longwave,
                              double tair, double vpd, double elevation, const double* ra_surface, const double* ra_overstory,
                              double* pot_evap) {
  const PenmanBase pb = penman_base(tair, elevation);
  const bool over = vl.f(veg_idx, VL_OVERSTORY) != 0.0;
  double net_short = 0.0;
#pragma unroll
  for (int i = 0; i < NPET; i++) {
    const int idx = (i < NPET_NON_NAT) ? o.nveg_types + i : veg_idx;
    double rs = vl.f(idx, VL_RMIN), rarc = vl.f(idx, VL_RARC), lai = vl.f(idx, VL_LAI + month - 1), albedo = vl.f(idx, VL_ALBEDO + month - 1);
    float RGL = (float)vl.f(idx, VL_RGL);
    if (i == PET_VEGNOCR) rs = 0;
    const bool ref_crop = (i == 2 || i == 3);
    double rc = calc_rc(rs, net_short, RGL, tair, vpd, lai, 1.0, ref_crop);
    double ra = (i < NPET_NON_NAT || !over) ? ra_surface[i] : ra_overstory[i];
    net_short = (1.0 - albedo) * shortwave;
    double net_rad = net_short + net_longwave;
    pot_evap[i] = penman_eval(pb, net_rad, vpd, ra, rc, rarc) * o.dt / 24.0;
  }
}

}  // namespace vic
