/*
 * orc_base.c — TEST INFRASTRUCTURE (CPU oracle): leaf physics, root finder, soil
 * thermal properties, runoff and evapotranspiration.  Each function cites the
 * reference file:line whose algorithm it restates.
 */
#include "orc.h"

/* ------------------------------------------------------------------ vapour pressure, Penman */

/* svp.c:7-24: saturation vapour pressure (Pa), ice correction below 0 C */
double orc_svp(double T) {
  double s = 0.61078 * exp((17.269 * T) / (237.3 + T));
  if (T < 0) s *= 1.0 + .00972 * T + .000042 * T * T;
  return s * 1000.;
}

/* svp.c:26-34 */
double orc_svp_slope(double T) {
  return (17.269 * 237.3) / ((237.3 + T) * (237.3 + T)) * orc_svp(T);
}

/* penman.c:44-95: Jarvis-type canopy resistance */
double orc_calc_rc(double rs, double net_short, float RGL, double tair, double vpd, double lai, double gsm_inv, int ref_crop) {
  const double CLOSURE = 4000, RSMAX = 5000, VPDMINFACTOR = 0.1;
  double rc;
  if (rs == 0) rc = 0;
  else if (lai == 0) rc = ORC_HUGE_RESIST;
  else if (ref_crop) rc = rs / (lai * 0.5);
  else {
    double dayf, tf, vf;
    if (rs > 0.) {
      double f = net_short / RGL;
      dayf = (1. + f) / (f + rs / RSMAX);
    } else dayf = 1.;
    tf = .08 * tair - 0.0016 * tair * tair;
    tf = (tf <= 0.0) ? 1e-10 : tf;
    vf = 1 - vpd / CLOSURE;
    vf = (vf < VPDMINFACTOR) ? VPDMINFACTOR : vf;
    rc = rs / (lai * gsm_inv * tf * vf) * dayf;
    rc = (rc > RSMAX) ? RSMAX : rc;
  }
  return rc;
}

/* penman.c:96-145: Penman-Monteith, mm/day */
double orc_penman(double tair, double elevation, double rad, double vpd, double ra, double rc, double rarc) {
  double slope = orc_svp_slope(tair);
  double h = 287 / 9.81 * ((tair + 273.15) + 0.5 * (double)elevation * -0.006);
  double pz = 101300 * exp(-(double)elevation / h);
  double lv = 2501000 - 2361 * tair;
  double gamma = 1628.6 * pz / lv;
  double r_air = 0.003486 * pz / (275 + tair);
  double evap = (slope * rad + r_air * 1013 * vpd / ra) / (lv * (slope + gamma * (1 + (rc + rarc) / ra))) * ORC_SEC_PER_DAY;
  if (vpd >= 0.0 && evap < 0.0) evap = 0.0;
  return evap;
}

/* StabilityCorrection.c:44-81: Richardson-number correction of the aerodynamic resistance */
double orc_stability_correction(double Z, double d, double TSurf, double Tair, double Wind, double Z0) {
  double corr = 1.0;
  const double RiCr = 0.2;
  if (TSurf != Tair) {
    double Ri = ORC_G * (Tair - TSurf) * (Z - d) / (((Tair + 273.15) + (TSurf + 273.15)) / 2.0 * Wind * Wind);
    double RiLimit = (Tair + 273.15) / (((Tair + 273.15) + (TSurf + 273.15)) / 2.0 * (log((Z - d) / Z0) + 5));
    if (Ri > RiLimit) Ri = RiLimit;
    if (Ri > 0.0) corr = (1 - Ri / RiCr) * (1 - Ri / RiCr);
    else {
      if (Ri < -0.5) Ri = -0.5;
      corr = sqrt(1 - 16 * Ri);
    }
  }
  return corr;
}

/* ------------------------------------------------------------------ root_brent.c:97-337 */
double orc_root_brent(double lower, double upper, orc_fn f, void *ctx) {
  return orc_root_brent_tol(lower, upper, f, ctx, 3e-8, 1e-7);                      /* root_brent.c:32-36 */
}

double orc_root_brent_tol(double lower, double upper, orc_fn f, void *ctx, double MACHEPS, double TTOL) {
  const int MAXTRIES = 5, MAXITER = 1000;
  const double TSTEP = 10;
  double a = lower, b = upper, c = 0, d = 0, e = 0, fa, fb, fc, m, p, q, r, s, tol;
  double last_bad = 0, last_good = 0;
  int which_err = 0, i, j;

  fa = f(a, ctx);
  fb = f(b, ctx);
  if (fa == ORC_ERROR && fb == ORC_ERROR) return ORC_ERROR;                        /* :129-132 */
  if (fa == ORC_ERROR || fb == ORC_ERROR) {                                         /* :136-177 */
    if (fa == ORC_ERROR) { which_err = -1; last_bad = a; last_good = b; }
    else { which_err = 1; last_good = a; last_bad = b; }
    c = 0.5 * (last_bad + last_good);
    fc = f(c, ctx);
    j = 0;
    while (fc == ORC_ERROR && j < MAXITER) {
      last_bad = c;
      c = 0.5 * (last_bad + last_good);
      fc = f(c, ctx);
      j++;
    }
    if (fc == ORC_ERROR) return ORC_ERROR;
    if (which_err == -1) { a = c; fa = fc; } else { b = c; fb = fc; }
  }
  j = 0;
  while ((fa * fb) >= 0 && j < MAXTRIES) {                                          /* :183-243 */
    if (which_err == 0) {
      a -= TSTEP; b += TSTEP;
      fa = f(a, ctx); fb = f(b, ctx);
    } else {
      if (which_err == -1) {
        b += TSTEP; fb = f(b, ctx);
        if (fb == ORC_ERROR) return ORC_ERROR;
        last_good = a;
      } else {
        a -= TSTEP; fa = f(a, ctx);
        if (fa == ORC_ERROR) return ORC_ERROR;
        last_good = b;
      }
      c = 0.5 * (last_good + last_bad);
      fc = f(c, ctx);
      i = 0;
      while (fc == ORC_ERROR && i < MAXITER) {
        last_bad = c;
        c = 0.5 * (last_bad + last_good);
        fc = f(c, ctx);
        i++;
      }
      if (fc == ORC_ERROR) return ORC_ERROR;
      if (which_err == -1) { a = c; fa = fc; } else { b = c; fb = fc; }
    }
    j++;
  }
  if ((fa * fb) >= 0) return ORC_ERROR;                                             /* :244-248 */

  fc = fb;
  for (i = 0; i < MAXITER; i++) {                                                   /* :256-332 */
    if (fb * fc > 0) { c = a; fc = fa; d = b - a; e = d; }
    if (fabs(fc) < fabs(fb)) { a = b; b = c; c = a; fa = fb; fb = fc; fc = fa; }
    tol = 2 * MACHEPS * fabs(b) + TTOL;
    m = 0.5 * (c - b);
    if (fabs(m) <= tol || fb == 0) return b;
    if (fabs(e) < tol || fabs(fa) <= fabs(fb)) { d = m; e = d; }
    else {
      s = fb / fa;
      if (a == c) { p = 2 * m * s; q = 1 - s; }
      else {
        q = fa / fc; r = fb / fc;
        p = s * (2 * m * q * (q - r) - (b - a) * (r - 1));
        q = (q - 1) * (r - 1) * (s - 1);
      }
      if (p > 0) q = -q; else p = -p;
      s = e; e = d;
      if ((2 * p) < (3 * m * q - fabs(tol * q)) && p < fabs(0.5 * s * q)) d = p / q;
      else { d = m; e = d; }
    }
    a = b; fa = fb;
    b += (fabs(d) > tol) ? d : ((m > 0) ? tol : -tol);
    fb = f(b, ctx);
    if (fb == ORC_ERROR) return ORC_ERROR;
  }
  return ORC_ERROR;
}

/* ------------------------------------------------------------------ aerodynamics */

/* calc_veg_params.c:26-41 */
double orc_calc_veg_height(double displacement, double L) {
  double X = ORC_COEF_DRAG * L;
  return displacement / (1.1 * log(1 + pow(X, 0.25)));
}

/* CalcAerodynamic.c:64-271 */
int orc_calc_aerodynamic(int overstory, double Height, double Trunk, double Z0_SNOW, double Z0_SOIL, double n,
                         orc_vc *ra, orc_vc *U, orc_vc *disp, orc_vc *zref, orc_vc *z0) {
  const double K2 = ORC_VON_K * ORC_VON_K;
  double tmp_wind = U->v[ORC_SNOW_FREE];
  int k;
  if (!overstory) {
    double Z0_Lower = z0->v[ORC_SNOW_FREE], d_Lower = disp->v[ORC_SNOW_FREE];
    U->v[ORC_SNOW_FREE] = log((2. + Z0_Lower) / Z0_Lower) / log((zref->v[ORC_SNOW_FREE] - d_Lower) / Z0_Lower);
    ra->v[ORC_SNOW_FREE] = log((2. + Z0_Lower) / Z0_Lower) * log((zref->v[ORC_SNOW_FREE] - d_Lower) / Z0_Lower) / K2;
    zref->v[ORC_CANOPY] = zref->v[ORC_SNOW_FREE];
    z0->v[ORC_CANOPY] = z0->v[ORC_SNOW_FREE];
    disp->v[ORC_CANOPY] = disp->v[ORC_SNOW_FREE];
    U->v[ORC_CANOPY] = U->v[ORC_SNOW_FREE];
    ra->v[ORC_CANOPY] = ra->v[ORC_SNOW_FREE];
    zref->v[ORC_SNOW_COVERED] = zref->v[ORC_SNOW_FREE];
    z0->v[ORC_SNOW_COVERED] = Z0_SNOW;
    disp->v[ORC_SNOW_COVERED] = 0.;
    U->v[ORC_SNOW_COVERED] = log((2. + Z0_SNOW) / Z0_SNOW) / log(zref->v[ORC_SNOW_COVERED] / Z0_SNOW);
    ra->v[ORC_SNOW_COVERED] = log((2. + Z0_SNOW) / Z0_SNOW) * log(zref->v[ORC_SNOW_COVERED] / Z0_SNOW) / K2;
    zref->v[ORC_SNOW_COVERED] = 2. + Z0_SNOW;
    zref->v[ORC_GLACIER_SURF] = zref->v[ORC_SNOW_FREE];
    z0->v[ORC_GLACIER_SURF] = Z0_Lower;
    disp->v[ORC_GLACIER_SURF] = 0.;
    U->v[ORC_GLACIER_SURF] = log((2. + Z0_Lower) / Z0_Lower) / log(zref->v[ORC_GLACIER_SURF] / Z0_Lower);
    ra->v[ORC_GLACIER_SURF] = log((2. + Z0_Lower) / Z0_Lower) * log(zref->v[ORC_GLACIER_SURF] / Z0_Lower) / K2;
    zref->v[ORC_GLACIER_SURF] = 2. + Z0_Lower;
  } else {
    double Z0_Upper = z0->v[ORC_SNOW_FREE], d_Upper = disp->v[ORC_SNOW_FREE];
    double Z0_Lower = Z0_SOIL, d_Lower = 0;
    double Zw = 1.5 * Height - 0.5 * d_Upper;
    double Zt = Trunk * Height;
    double Uw, Uh, Ut, zr = zref->v[ORC_SNOW_FREE];
    if (Zt < (Z0_Lower + d_Lower)) return -1;                                      /* :214-217 */
    ra->v[ORC_CANOPY] = log((zr - d_Upper) / Z0_Upper) / K2
        * (Height / (n * (Zw - d_Upper)) * (exp(n * (1 - (d_Upper + Z0_Upper) / Height)) - 1)
           + (Zw - Height) / (Zw - d_Upper) + log((zr - d_Upper) / (Zw - d_Upper)));
    Uw = log((Zw - d_Upper) / Z0_Upper) / log((zr - d_Upper) / Z0_Upper);
    Uh = Uw - (1 - (Height - d_Upper) / (Zw - d_Upper)) / log((zr - d_Upper) / Z0_Upper);
    U->v[ORC_CANOPY] = Uh * exp(n * ((Z0_Upper + d_Upper) / Height - 1.));
    Ut = Uh * exp(n * (Zt / Height - 1.));
    U->v[ORC_SNOW_FREE] = Ut * log((2. + Z0_Lower) / Z0_Lower) / log(Zt / Z0_Lower);
    ra->v[ORC_SNOW_FREE] = log((2. + Z0_Lower) / Z0_Lower) * log(Zt / Z0_Lower) / (K2 * Ut);
    if (Zt > (2. + Z0_SNOW)) {
      U->v[ORC_SNOW_COVERED] = Ut * log((2. + Z0_SNOW) / Z0_SNOW) / log(Zt / Z0_SNOW);
      ra->v[ORC_SNOW_COVERED] = log((2. + Z0_SNOW) / Z0_SNOW) * log(Zt / Z0_SNOW) / (K2 * Ut);
    } else if (Height > (2. + Z0_SNOW)) {
      U->v[ORC_SNOW_COVERED] = Uh * exp(n * ((2. + Z0_SNOW) / Height - 1.));
      ra->v[ORC_SNOW_COVERED] = log(Zt / Z0_SNOW) * log(Zt / Z0_SNOW) / (K2 * Ut)
          + Height * log((zr - d_Upper) / Z0_Upper) / (n * K2 * (Zw - d_Upper))
            * (exp(n * (1 - Zt / Height)) - exp(n * (1 - (Z0_SNOW + 2.) / Height)));
    } else {
      U->v[ORC_SNOW_COVERED] = Uh;
      ra->v[ORC_SNOW_COVERED] = log(Zt / Z0_SNOW) * log(Zt / Z0_SNOW) / (K2 * Ut)
          + Height * log((zr - d_Upper) / Z0_Upper) / (n * K2 * (Zw - d_Upper)) * (exp(n * (1 - Zt / Height)) - 1);
    }
    zref->v[ORC_CANOPY] = zref->v[ORC_SNOW_FREE];
    z0->v[ORC_CANOPY] = z0->v[ORC_SNOW_FREE];
    disp->v[ORC_CANOPY] = disp->v[ORC_SNOW_FREE];
    zref->v[ORC_SNOW_FREE] = 2. + Z0_Lower;
    z0->v[ORC_SNOW_FREE] = Z0_Lower;
    disp->v[ORC_SNOW_FREE] = d_Lower;
    zref->v[ORC_SNOW_COVERED] = 2. + Z0_SNOW;
    z0->v[ORC_SNOW_COVERED] = Z0_SNOW;
    disp->v[ORC_SNOW_COVERED] = 0.;
    zref->v[ORC_GLACIER_SURF] = 2. + Z0_Lower;
    z0->v[ORC_GLACIER_SURF] = Z0_Lower;
    disp->v[ORC_GLACIER_SURF] = 0.;
  }
  if (tmp_wind > 0.) {                                                              /* :246-269 */
    U->v[ORC_SNOW_FREE] *= tmp_wind;
    ra->v[ORC_SNOW_FREE] /= tmp_wind;
    for (k = 1; k < ORC_NCASE; k++)
      if (!isnan(U->v[k])) { U->v[k] *= tmp_wind; ra->v[k] /= tmp_wind; }
  } else {
    U->v[ORC_SNOW_FREE] *= tmp_wind;
    ra->v[ORC_SNOW_FREE] = ORC_HUGE_RESIST;
    for (k = 1; k < ORC_NCASE; k++) {
      if (!isnan(U->v[k])) U->v[k] *= tmp_wind;
      ra->v[k] = ORC_HUGE_RESIST;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ soil thermal properties */

/* soil_conduction.c:7-105 (Johansen) */
double orc_soil_conductivity(double moist, double Wu, double soil_dens_min, double bulk_dens_min, double quartz,
                             double soil_density, double bulk_density, double organic) {
  const double Ki = 2.2, Kw = 0.57, Kdry_org = 0.05, Ks_org = 0.25;
  double Kdry_min = (0.135 * bulk_dens_min + 64.7) / (soil_dens_min - 0.947 * bulk_dens_min);
  double Kdry = (1 - organic) * Kdry_min + organic * Kdry_org;
  double K;
  if (moist > 0.) {
    double porosity = 1.0 - bulk_density / soil_density;
    double Sr = moist / porosity;
    double Ks_min, Ks, Ksat, Ke;
    if (quartz < .2) Ks_min = pow(7.7, quartz) * pow(3.0, 1.0 - quartz);
    else Ks_min = pow(7.7, quartz) * pow(2.2, 1.0 - quartz);
    Ks = (1 - organic) * Ks_min + organic * Ks_org;
    if (Wu == moist) {
      Ksat = pow(Ks, 1.0 - porosity) * pow(Kw, porosity);
      Ke = 0.7 * log10(Sr) + 1.0;
    } else {
      Ksat = pow(Ks, 1.0 - porosity) * pow(Ki, porosity - Wu) * pow(Kw, Wu);
      Ke = Sr;
    }
    K = (Ksat - Kdry) * Ke + Kdry;
    if (K < Kdry) K = Kdry;
  } else K = Kdry;
  return K;
}

/* soil_conduction.c:108-139 */
double orc_volumetric_heat_capacity(double soil_fract, double water_fract, double ice_fract, double organic_fract) {
  double Cs = 2.0e6 * soil_fract * (1 - organic_fract);
  Cs += 2.7e6 * soil_fract * organic_fract;
  Cs += 4.2e6 * water_fract;
  Cs += 1.9e6 * ice_fract;
  Cs += 1.3e3 * (1. - (soil_fract + water_fract + ice_fract));
  return Cs;
}

/* soil_conduction.c:830-863 */
double orc_maximum_unfrozen_water(double T, double max_moist, double bubble, double expt) {
  double u;
  if (T <= 0) {
    u = max_moist * pow((-ORC_LF * T) / 273.16 / (9.81 * bubble / 100.), -(2.0 / (expt - 3.0)));
    if (u > max_moist) u = max_moist;
    if (u < 0) u = 0;
  } else u = max_moist;
  return u;
}

/* linear_interp (vicNl.h helper, used at soil_conduction.c:533,541,812) */
double orc_linear_interp(double x, double lx, double ux, double ly, double uy) {
  return (x - lx) / (ux - lx) * (uy - ly) + ly;
}

/* soil_conduction.c:725-773 */
void orc_layer_thermal_properties(orc_layer *layer, const orc_soil *sc) {
  int l;
  for (l = 0; l < 3; l++) {
    double moist = layer[l].moist / sc->depth[l] / 1000;
    double ice = layer[l].ice / sc->depth[l] / 1000;
    layer[l].kappa = orc_soil_conductivity(moist, moist - ice, sc->soil_dens_min[l], sc->bulk_dens_min[l], sc->quartz[l],
                                           sc->soil_density[l], sc->bulk_density[l], sc->organic[l]);
    layer[l].Cs = orc_volumetric_heat_capacity(sc->bulk_density[l] / sc->soil_density[l], moist - ice, ice, sc->organic[l]);
  }
}

/* soil_conduction.c:304-440 */
int orc_distribute_node_moisture_properties(const orc_model *m, orc_energy *e, const orc_soil *sc, const double *moist) {
  const int Nn = m->opt.Nnode;
  int n, l = 0, past_bottom = 0;
  double Lsum = 0.;
  for (n = 0; n < Nn; n++) {
    if (sc->Zsum_node[n] == Lsum + sc->depth[l] && n != 0 && l != 2) {
      e->moist[n] = (moist[l] / sc->depth[l] + moist[l + 1] / sc->depth[l + 1]) / 1000 / 2.;
    } else {
      e->moist[n] = moist[l] / sc->depth[l] / 1000;
    }
    if (e->moist[n] - sc->max_moist_node[n] > 0) e->moist[n] = sc->max_moist_node[n];
    if (e->T[n] < 0 && (sc->FS_ACTIVE && m->opt.FROZEN_SOIL)) {
      e->ice[n] = e->moist[n] - orc_maximum_unfrozen_water(e->T[n], sc->max_moist_node[n], sc->bubble_node[n], sc->expt_node[n]);
      if (e->ice[n] < 0) e->ice[n] = 0;
      e->kappa_node[n] = orc_soil_conductivity(e->moist[n], e->moist[n] - e->ice[n], sc->soil_dens_min[l], sc->bulk_dens_min[l],
                                               sc->quartz[l], sc->soil_density[l], sc->bulk_density[l], sc->organic[l]);
    } else {
      e->ice[n] = 0;
      e->kappa_node[n] = orc_soil_conductivity(e->moist[n], e->moist[n], sc->soil_dens_min[l], sc->bulk_dens_min[l],
                                               sc->quartz[l], sc->soil_density[l], sc->bulk_density[l], sc->organic[l]);
    }
    e->Cs_node[n] = orc_volumetric_heat_capacity(sc->bulk_density[l] / sc->soil_density[l], e->moist[n] - e->ice[n], e->ice[n],
                                                 sc->organic[l]);
    if (sc->Zsum_node[n] > Lsum + sc->depth[l] && !past_bottom) {
      Lsum += sc->depth[l];
      l++;
      if (l == 3) { past_bottom = 1; l = 2; }
    }
  }
  return 0;
}

/* soil_conduction.c:444-614 (SPATIAL_FROST off: one frost area) */
int orc_estimate_layer_ice_content(const orc_model *m, orc_layer *layer, const double *T, const orc_soil *sc) {
  const int Nn = m->opt.Nnode;
  double Lsum[4], tmpT[VIC_MAX_NODES], tmpZ[VIC_MAX_NODES], tmp_ice[VIC_MAX_NODES];
  int l, n, min_n, max_n;
  Lsum[0] = 0;
  for (l = 1; l <= 3; l++) Lsum[l] = sc->depth[l - 1] + Lsum[l - 1];
  for (l = 0; l < 3; l++) {
    layer[l].T = 0.;
    layer[l].ice = 0.;
    min_n = Nn - 2;
    while (Lsum[l] < sc->Zsum_node[min_n] && min_n > 0) min_n--;
    max_n = 1;
    while (Lsum[l + 1] > sc->Zsum_node[max_n] && max_n < Nn) max_n++;
    if (max_n >= Nn) return -1;                                                     /* :526-529 */
    if (sc->Zsum_node[min_n] < Lsum[l])
      tmpT[min_n] = orc_linear_interp(Lsum[l], sc->Zsum_node[min_n], sc->Zsum_node[min_n + 1], T[min_n], T[min_n + 1]);
    else tmpT[min_n] = T[min_n];
    tmpZ[min_n] = Lsum[l];
    for (n = min_n + 1; n < max_n; n++) { tmpT[n] = T[n]; tmpZ[n] = sc->Zsum_node[n]; }
    if (sc->Zsum_node[max_n] > Lsum[l + 1])
      tmpT[max_n] = orc_linear_interp(Lsum[l + 1], sc->Zsum_node[max_n - 1], sc->Zsum_node[max_n], T[max_n - 1], T[max_n]);
    else tmpT[max_n] = T[max_n];
    tmpZ[max_n] = Lsum[l + 1];
    if (m->opt.FROZEN_SOIL && sc->FS_ACTIVE) {
      for (n = min_n; n <= max_n; n++) {
        tmp_ice[n] = layer[l].moist - orc_maximum_unfrozen_water(tmpT[n], sc->max_moist[l], sc->bubble[l], sc->expt[l]);
        if (tmp_ice[n] < 0) tmp_ice[n] = 0.;
      }
    } else {
      for (n = min_n; n <= max_n; n++) tmp_ice[n] = 0;
    }
    for (n = min_n; n < max_n; n++) {
      layer[l].ice += (tmpZ[n + 1] - tmpZ[n]) * (tmp_ice[n + 1] + tmp_ice[n]) / 2.;
      layer[l].T += (tmpZ[n + 1] - tmpZ[n]) * (tmpT[n + 1] + tmpT[n]) / 2.;
    }
    layer[l].ice /= sc->depth[l];
    layer[l].T /= sc->depth[l];
  }
  return 0;
}

/* soil_conduction.c:617-723 */
void orc_estimate_layer_ice_content_quick_flux(const orc_model *m, orc_layer *layer, double Tsurf, double T1, const orc_soil *sc) {
  double Lsum[4];
  int l;
  Lsum[0] = 0;
  for (l = 1; l <= 3; l++) Lsum[l] = sc->depth[l - 1] + Lsum[l - 1];
  layer[0].T = 0.5 * (Tsurf + T1);
  for (l = 1; l < 3; l++)
    layer[l].T = sc->avg_temp - sc->dp / (sc->depth[l]) * (T1 - sc->avg_temp)
                 * (exp(-(Lsum[l + 1] - Lsum[1]) / sc->dp) - exp(-(Lsum[l] - Lsum[1]) / sc->dp));
  for (l = 0; l < 3; l++) {
    layer[l].ice = 0;
    if (m->opt.FROZEN_SOIL && sc->FS_ACTIVE) {
      layer[l].ice = layer[l].moist - orc_maximum_unfrozen_water(layer[l].T, sc->max_moist[l], sc->bubble[l], sc->expt[l]);
      if (layer[l].ice < 0) layer[l].ice = 0;
      if (layer[l].ice > layer[l].moist) layer[l].ice = layer[l].moist;
    }
  }
}

/* soil_conduction.c:775-828 */
void orc_find_0_degree_fronts(orc_energy *e, const double *Zsum, const double *T, int Nnodes) {
  int n, f, Nthaw = 0, Nfrost = 0;
  double td[3], fd[3];
  for (f = 0; f < 3; f++) { fd[f] = NAN; td[f] = NAN; }
  for (n = Nnodes - 2; n >= 0; n--) {
    if (T[n] > 0 && T[n + 1] <= 0 && Nthaw < 3) {
      td[Nthaw] = orc_linear_interp(0, T[n], T[n + 1], Zsum[n], Zsum[n + 1]);
      Nthaw++;
    } else if (T[n] < 0 && T[n + 1] >= 0 && Nfrost < 3) {
      fd[Nfrost] = orc_linear_interp(0, T[n], T[n + 1], Zsum[n], Zsum[n + 1]);
      Nfrost++;
    }
  }
  for (f = 0; f < 3; f++) { e->tdepth[f] = td[f]; e->fdepth[f] = fd[f]; }
  e->Nthaw = Nthaw;
  e->Nfrost = Nfrost;
}

/* ------------------------------------------------------------------ water table: compute_zwt.c:7-112 */
static double orc_compute_zwt(const orc_soil *sc, int l, double moist) {
  double zwt = NAN;
  int i = VIC_MAX_ZWTVMOIST - 1;
  while (i >= 1 && moist > sc->zwt_moist[l][i]) i--;
  if (i == VIC_MAX_ZWTVMOIST - 1) {
    if (moist < sc->zwt_moist[l][i]) zwt = NAN;
    else if (moist == sc->zwt_moist[l][i]) zwt = sc->zwt_zwt[l][i];
  } else {
    zwt = sc->zwt_zwt[l][i + 1] + (sc->zwt_zwt[l][i] - sc->zwt_zwt[l][i + 1]) * (moist - sc->zwt_moist[l][i + 1])
          / (sc->zwt_moist[l][i] - sc->zwt_moist[l][i + 1]);
  }
  return zwt;
}

void orc_wrap_compute_zwt(const orc_soil *sc, orc_hru *h) {
  int l;
  double total_depth = 0, tmp_depth, tmp_moist;
  for (l = 0; l < 3; l++) total_depth += sc->depth[l];
  for (l = 0; l < 3; l++) h->layer[l].zwt = orc_compute_zwt(sc, l, h->layer[l].moist);
  if (isnan(h->layer[2].zwt)) h->layer[2].zwt = -total_depth * 100;
  l = 2;
  tmp_depth = total_depth;
  while (l >= 0 && sc->max_moist[l] - h->layer[l].moist <= ORC_SMALL) { tmp_depth -= sc->depth[l]; l--; }
  if (l < 0) h->zwt = 0;
  else if (l < 2) {
    if (!isnan(h->layer[l].zwt)) h->zwt = h->layer[l].zwt;
    else h->zwt = -tmp_depth * 100;
  } else h->zwt = h->layer[l].zwt;
  tmp_moist = 0;
  for (l = 0; l < 2; l++) tmp_moist += h->layer[l].moist;
  h->zwt2 = orc_compute_zwt(sc, 3, tmp_moist);
  if (isnan(h->zwt2)) h->zwt2 = h->layer[2].zwt;
  tmp_moist = 0;
  for (l = 0; l < 3; l++) tmp_moist += h->layer[l].moist;
  h->zwt3 = orc_compute_zwt(sc, 4, tmp_moist);
  if (isnan(h->zwt3)) h->zwt3 = -total_depth * 100;
}

/* ------------------------------------------------------------------ runoff.c */

/* runoff.c:773-813 (Wood et al. 1992 eqs 1, 3a, 3b) */
void orc_compute_runoff_and_asat(const orc_soil *sc, const double *moist, double inflow, double *A, double *runoff) {
  double top_moist = 0., top_max_moist = 0., ex, max_infil, i_0, basis;
  int l;
  for (l = 0; l < 2; l++) { top_moist += moist[l]; top_max_moist += sc->max_moist[l]; }
  if (top_moist > top_max_moist) top_moist = top_max_moist;
  ex = sc->b_infilt / (1.0 + sc->b_infilt);
  *A = 1.0 - pow((1.0 - top_moist / top_max_moist), ex);
  max_infil = (1.0 + sc->b_infilt) * top_max_moist;
  i_0 = max_infil * (1.0 - pow((1.0 - *A), (1.0 / sc->b_infilt)));
  if (inflow == 0.0) *runoff = 0.0;
  else if (max_infil == 0.0) *runoff = inflow;
  else if ((i_0 + inflow) > max_infil) *runoff = inflow - top_max_moist + top_moist;
  else {
    basis = 1.0 - (i_0 + inflow) / max_infil;
    *runoff = (inflow - top_max_moist + top_moist + top_max_moist * pow(basis, 1.0 * (1.0 + sc->b_infilt)));
  }
  if (*runoff < 0.) *runoff = 0.;
}

/* runoff.c:7-771 with Ndist 1, one frost area, no EXCESS_ICE / LOW_RES_MOIST */
int orc_runoff(const orc_model *m, orc_hru *h, const orc_soil *sc, double ppt) {
  const int dt = m->opt.dt;
  double resid[3], liq[3], ice[3], maxm[3], Ksat[3], Q12[2], evap[3], tmpm[3], moist[3];
  double A, inflow, runoff, tmp_dt_runoff, baseflow = 0, dt_inflow, dt_runoff, Dsmax, tmp_inflow, tmp_moist, tmp_liq;
  double dt_baseflow, rel_moist, frac, tmp_runoff;
  int l, ts, tmplayer, lindex;
  for (l = 0; l < 3; l++) resid[l] = sc->resid_moist[l] * sc->depth[l] * 1000.;
  h->runoff = 0; h->baseflow = 0; h->asat = 0;
  for (l = 0; l < 3; l++) evap[l] = h->layer[l].evap / (double)dt;                 /* :292-293 */
  inflow = ppt;
  for (l = 0; l < 3; l++) {
    Ksat[l] = sc->Ksat[l] / 24.;
    liq[l] = h->layer[l].moist - h->layer[l].ice;
    ice[l] = h->layer[l].ice;
    maxm[l] = sc->max_moist[l];
  }
  for (l = 0; l < 3; l++) tmpm[l] = (liq[l] + ice[l]);
  orc_compute_runoff_and_asat(sc, tmpm, inflow, &A, &runoff);                       /* :436-439 */
  tmp_dt_runoff = runoff / (double)dt;
  dt_inflow = inflow / (double)dt;
  lindex = 2;
  for (ts = 0; ts < dt; ts++) {                                                     /* :451-700 */
    inflow = dt_inflow;
    for (l = 0; l < 2; l++) {                                                       /* Brooks-Corey drainage :475-503 */
      if ((tmp_liq = liq[l] - evap[l]) < resid[l]) tmp_liq = resid[l];
      if (liq[l] > resid[l])
        Q12[l] = Ksat[l] * pow(((tmp_liq - resid[l]) / (sc->max_moist[l] - resid[l])), sc->expt[l]);
      else Q12[l] = 0.;
    }
    for (l = 0; l < 2; l++) {                                                       /* :513-613 */
      if (l == 0) dt_runoff = tmp_dt_runoff; else dt_runoff = 0;
      tmp_inflow = 0.;
      liq[l] = liq[l] + (inflow - dt_runoff) - (Q12[l] + evap[l]);
      if ((liq[l] + ice[l]) > maxm[l]) {
        tmp_inflow = (liq[l] + ice[l]) - maxm[l];
        liq[l] = maxm[l] - ice[l];
        if (l == 0) { Q12[l] += tmp_inflow; tmp_inflow = 0; }
        else {
          tmplayer = l;
          while (tmp_inflow > 0) {
            tmplayer--;
            if (tmplayer < 0) { runoff += tmp_inflow; tmp_inflow = 0; }
            else {
              liq[tmplayer] += tmp_inflow;
              if ((liq[tmplayer] + ice[tmplayer]) > maxm[tmplayer]) {
                tmp_inflow = ((liq[tmplayer] + ice[tmplayer]) - maxm[tmplayer]);
                liq[tmplayer] = maxm[tmplayer] - ice[tmplayer];
              } else tmp_inflow = 0;
            }
          }
        }
      }
      if ((liq[l] + ice[l]) < resid[l]) {
        Q12[l] += (liq[l] + ice[l]) - resid[l];
        liq[l] = resid[l] - ice[l];
      }
      inflow = (Q12[l] + tmp_inflow);
      Q12[l] += tmp_inflow;
    }
    /* ARNO baseflow from the bottom layer :622-698 */
    lindex = 2;
    Dsmax = sc->Dsmax / 24.;
    rel_moist = (liq[lindex] - resid[lindex]) / (sc->max_moist[lindex] - resid[lindex]);
    frac = Dsmax * sc->Ds / sc->Ws;
    dt_baseflow = frac * rel_moist;
    if (rel_moist > sc->Ws) {
      frac = (rel_moist - sc->Ws) / (1 - sc->Ws);
      dt_baseflow += Dsmax * (1 - sc->Ds / sc->Ws) * pow(frac, sc->c);
    }
    if (dt_baseflow < 0) dt_baseflow = 0;
    liq[lindex] += Q12[lindex - 1] - (evap[lindex] + dt_baseflow);
    tmp_moist = 0;
    if ((liq[lindex] + ice[lindex]) < resid[lindex]) {
      dt_baseflow += (liq[lindex] + ice[lindex]) - resid[lindex];
      liq[lindex] = resid[lindex] - ice[lindex];
    }
    if ((liq[lindex] + ice[lindex]) > maxm[lindex]) {
      tmp_moist = ((liq[lindex] + ice[lindex]) - maxm[lindex]);
      liq[lindex] = maxm[lindex] - ice[lindex];
      tmplayer = lindex;
      while (tmp_moist > 0) {
        tmplayer--;
        if (tmplayer < 0) { runoff += tmp_moist; tmp_moist = 0; }
        else {
          liq[tmplayer] += tmp_moist;
          if ((liq[tmplayer] + ice[tmplayer]) > maxm[tmplayer]) {
            tmp_moist = ((liq[tmplayer] + ice[tmplayer]) - maxm[tmplayer]);
            liq[tmplayer] = maxm[tmplayer] - ice[tmplayer];
          } else tmp_moist = 0;
        }
      }
    }
    baseflow += dt_baseflow;
  }
  if (baseflow < 0) {                                                               /* :707-710, uses the leftover lindex */
    h->layer[lindex].evap += baseflow;
    baseflow = 0;
  }
  for (l = 0; l < 3; l++) tmpm[l] = (liq[l] + ice[l]);
  orc_compute_runoff_and_asat(sc, tmpm, 0, &A, &tmp_runoff);
  for (l = 0; l < 3; l++) h->layer[l].moist = liq[l] + ice[l];
  h->asat += A;
  h->runoff += runoff;
  h->baseflow += baseflow;
  orc_wrap_compute_zwt(sc, h);                                                      /* :746 */
  if (m->opt.FULL_ENERGY || m->opt.FROZEN_SOIL) {                                   /* :751-768 */
    for (l = 0; l < 3; l++) moist[l] = h->layer[l].moist;
    if (orc_distribute_node_moisture_properties(m, &h->energy, sc, moist) != 0) return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------ evapotranspiration */

/* canopy_evap.c:218-442 */
static void orc_transpiration(const orc_model *m, const orc_layer *layer, int veg_idx, int month, double rad, double vpd,
                              double net_short, double air_temp, double ra, double f, double delta_t, double Wdew,
                              double elevation, const orc_soil *sc, double *layerevap, const double *root) {
  const double *vl = orc_veg(m, veg_idx);
  const double *Wcr = sc->Wcr, *Wpwp = sc->Wpwp;
  double avail[3], ice[3], moist1 = 0.0, moist2, Wcr1 = 0.0, gsm_inv, rc, evap, root_sum, spare_evap;
  const double rmin = vl[VL_RMIN], rarc = vl[VL_RARC], lai = vl[VL_LAI + month - 1], wdmax = vl[VL_WDMAX + month - 1];
  const float RGL = (float)vl[VL_RGL];
  int i;
  for (i = 0; i < 3; i++) ice[i] = layer[i].ice;
  for (i = 0; i < 2; i++) {
    if (root[i] > 0.) {
      avail[i] = layer[i].moist - layer[i].ice;
      moist1 += avail[i];
      Wcr1 += Wcr[i];
    } else avail[i] = 0.;
  }
  moist2 = layer[2].moist - layer[2].ice;
  avail[2] = moist2;
  if ((moist1 >= Wcr1 && moist2 >= Wcr[2] && Wcr1 > 0.) || (moist1 >= Wcr1 && (1 - root[2]) >= 0.5)
      || (moist2 >= Wcr[2] && root[2] >= 0.5)) {
    gsm_inv = 1.0;
    rc = orc_calc_rc(rmin, net_short, RGL, air_temp, vpd, lai, gsm_inv, 0);
    evap = orc_penman(air_temp, elevation, rad, vpd, ra, rc, rarc) * delta_t / ORC_SEC_PER_DAY
           * (1.0 - f * pow((Wdew / wdmax), (2.0 / 3.0)));
    root_sum = 1.0;
    spare_evap = 0.0;
    for (i = 0; i < 3; i++) {
      if (avail[i] >= Wcr[i]) layerevap[i] = evap * (double)root[i];
      else {
        if (avail[i] >= Wpwp[i]) gsm_inv = (avail[i] - Wpwp[i]) / (Wcr[i] - Wpwp[i]);
        else gsm_inv = 0.0;
        layerevap[i] = evap * gsm_inv * (double)root[i];
        root_sum -= root[i];
        spare_evap = evap * (double)root[i] * (1.0 - gsm_inv);
      }
    }
    if (spare_evap > 0.0)
      for (i = 0; i < 3; i++)
        if (avail[i] >= Wcr[i]) layerevap[i] += (double)root[i] * spare_evap / root_sum;
  } else {
    for (i = 0; i < 3; i++) {
      if (avail[i] >= Wcr[i]) gsm_inv = 1.0;
      else if (avail[i] >= Wpwp[i]) gsm_inv = (avail[i] - Wpwp[i]) / (Wcr[i] - Wpwp[i]);
      else gsm_inv = 0.0;
      if (gsm_inv > 0.0) {
        rc = orc_calc_rc(rmin, net_short, RGL, air_temp, vpd, lai, gsm_inv, 0);
        layerevap[i] = orc_penman(air_temp, elevation, rad, vpd, ra, rc, rarc) * delta_t / ORC_SEC_PER_DAY * (double)root[i]
                       * (1.0 - f * pow((Wdew / wdmax), (2.0 / 3.0)));
      } else layerevap[i] = 0.0;
    }
  }
  for (i = 0; i < 3; i++) {
    if (ice[i] > 0) {
      if (ice[i] >= Wpwp[i]) { if (layerevap[i] > avail[i]) layerevap[i] = avail[i]; }
      else { if (layerevap[i] > layer[i].moist - Wpwp[i]) layerevap[i] = layer[i].moist - Wpwp[i]; }
    } else {
      if (layerevap[i] > layer[i].moist - Wpwp[i]) layerevap[i] = layer[i].moist - Wpwp[i];
    }
    if (layerevap[i] < 0.0) layerevap[i] = 0.0;
  }
}

/* canopy_evap.c:46-212 (Ndist 1, mu 1).  root[] values are the float-rounded fractions of veg_con. */
double orc_canopy_evap(const orc_model *m, orc_layer *layer, orc_vegvar *vv, int calc_evap, int veg_idx, int month,
                       double *Wdew, double delta_t, double rad, double vpd, double net_short, double air_temp, double ra,
                       double elevation, double ppt, const orc_soil *sc, const double *root) {
  const double *vl = orc_veg(m, veg_idx);
  const double wdmax = vl[VL_WDMAX + month - 1];
  double layerevap[3] = {0, 0, 0}, canopyevap, throughfall = 0, tmp_Wdew = *Wdew, f, rc, tmp_Evap;
  int i;
  vv->Wdew = tmp_Wdew;
  if (tmp_Wdew > wdmax) { throughfall = tmp_Wdew - wdmax; tmp_Wdew = wdmax; }
  rc = orc_calc_rc(0.0, net_short, (float)vl[VL_RGL], air_temp, vpd, vl[VL_LAI + month - 1], 1.0, 0);
  canopyevap = pow((tmp_Wdew / wdmax), (2.0 / 3.0)) * orc_penman(air_temp, elevation, rad, vpd, ra, rc, vl[VL_RARC])
               * delta_t / ORC_SEC_PER_DAY;
  if (canopyevap > 0.0 && delta_t == ORC_SEC_PER_DAY) f = fmin(1.0, ((tmp_Wdew + ppt) / canopyevap));
  else if (canopyevap > 0.0) f = fmin(1.0, ((tmp_Wdew) / canopyevap));
  else f = 1.0;
  canopyevap *= f;
  tmp_Wdew += ppt - canopyevap;
  if (tmp_Wdew < 0.0) tmp_Wdew = 0.0;
  if (tmp_Wdew <= wdmax) throughfall += 0.0;
  else { throughfall += tmp_Wdew - wdmax; tmp_Wdew = wdmax; }
  if (calc_evap)
    orc_transpiration(m, layer, veg_idx, month, rad, vpd, net_short, air_temp, ra, f, delta_t, vv->Wdew, elevation, sc,
                      layerevap, root);
  vv->canopyevap = canopyevap;
  vv->throughfall = throughfall;
  vv->Wdew = tmp_Wdew;
  *Wdew = *Wdew;   /* the caller's Wdew[] array is only read (canopy_evap.c:136) */
  tmp_Evap = canopyevap;
  for (i = 0; i < 3; i++) { layer[i].evap = layerevap[i]; tmp_Evap += layerevap[i]; }
  return tmp_Evap * 1.0 / (1000. * delta_t);
}

/* arno_evap.c:61-228 (Ndist 1, mu 1) */
double orc_arno_evap(orc_layer *layer, double rad, double air_temp, double vpd, double depth1, double max_moist,
                     double elevation, double b_infilt, double ra, double delta_t, double moist_resid) {
  double moist, Epot, max_infil, tmp, ratio, evap, as, dummy, beta_asp, tmpsum;
  int num_term, i;
  moist = layer[0].moist - layer[0].ice;
  if (moist > max_moist) moist = max_moist;
  Epot = orc_penman(air_temp, elevation, rad, vpd, ra, 0.0, 0.0) * delta_t / ORC_SEC_PER_DAY;
  max_infil = (1.0 + b_infilt) * max_moist;
  if (b_infilt == -1.0) tmp = max_infil;
  else {
    ratio = 1.0 - (moist) / (max_moist);
    if (ratio > 1.0) return ORC_ERROR;
    else if (ratio < 0.0) return ORC_ERROR;
    else ratio = pow(ratio, (1.0 / (b_infilt + 1.0)));
    tmp = max_infil * (1.0 - ratio);
  }
  if (tmp >= max_infil) evap = Epot;
  else {
    ratio = tmp / max_infil;
    ratio = 1.0 - ratio;
    if (ratio > 1.0) return ORC_ERROR;
    else if (ratio < 0.0) return ORC_ERROR;
    else if (ratio != 0.0) ratio = pow(ratio, b_infilt);
    as = 1 - ratio;
    ratio = pow(ratio, (1.0 / b_infilt));
    dummy = 1.0;
    for (num_term = 1; num_term <= 30; num_term++) {
      tmpsum = ratio;
      for (i = 1; i < num_term; i++) tmpsum *= ratio;
      dummy += b_infilt * tmpsum / (b_infilt + num_term);
    }
    beta_asp = as + (1.0 - as) * (1.0 - ratio) * dummy;
    evap = Epot * beta_asp;
  }
  if (evap > 0.0) {
    if (moist > moist_resid * depth1 * 1000.) {
      if (evap > moist - moist_resid * depth1 * 1000.) evap = moist - moist_resid * depth1 * 1000.;
    } else evap = 0.0;
  }
  layer[0].evap = evap;
  return evap / 1000. / delta_t * 1.0;
}

/* compute_pot_evap.c:8-78, including the stale net_short of Appendix C #4 */
void orc_compute_pot_evap(const orc_model *m, int veg_idx, int month, int dt, double shortwave, double net_longwave,
                          double tair, double vpd, double elevation, const double *ra_surface, const double *ra_overstory,
                          double *pot_evap) {
  static const int ref_crop[ORC_NPET] = {0, 0, 1, 1, 0, 0};
  const int nv = m->opt.nveg_types;
  double net_short = 0.0;   /* uninitialised in the reference; never consumed before being assigned (rs == 0 for types 0,1) */
  int i;
  for (i = 0; i < ORC_NPET; i++) {
    const double *vl = (i < ORC_NPET_NON_NAT) ? orc_veg(m, nv + i) : orc_veg(m, veg_idx);
    double rs = vl[VL_RMIN], rarc = vl[VL_RARC], lai = vl[VL_LAI + month - 1], albedo = vl[VL_ALBEDO + month - 1];
    float RGL = (float)vl[VL_RGL];
    double rc, ra, net_rad;
    if (i >= ORC_NPET_NON_NAT && i == ORC_PET_VEGNOCR) rs = 0;
    rc = orc_calc_rc(rs, net_short, RGL, tair, vpd, lai, 1.0, ref_crop[i]);
    if (i < ORC_NPET_NON_NAT || !(orc_veg(m, veg_idx)[VL_OVERSTORY] != 0)) ra = ra_surface[i];
    else ra = ra_overstory[i];
    net_short = (1.0 - albedo) * shortwave;
    net_rad = net_short + net_longwave;
    pot_evap[i] = orc_penman(tair, elevation, net_rad, vpd, ra, rc, rarc) * dt / 24.0;
  }
}
