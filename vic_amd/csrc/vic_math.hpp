// vic_math.hpp — leaf physics and the SIMT root finder (device only, gfx950).
// Reference citations are pacificclimate/VIC file:line of the algorithm each function computes.
#pragma once
#include "vic_types.hpp"

namespace vic {

// svp.c:7-24 — saturation vapour pressure (Pa)
VIC_DEV double svp(double T) {
  double s = 0.61078 * exp((17.269 * T) / (237.3 + T));
  if (T < 0) s *= 1.0 + .00972 * T + .000042 * T * T;
  return s * 1000.;
}
// svp.c:26-34
VIC_DEV double svp_slope(double T) { return (17.269 * 237.3) / ((237.3 + T) * (237.3 + T)) * svp(T); }

// penman.c:44-95
VIC_DEV double calc_rc(double rs, double net_short, float RGL, double tair, double vpd, double lai, double gsm_inv, bool ref_crop) {
  const double CLOSURE = 4000, RSMAX = 5000, VPDMINFACTOR = 0.1;
  double rc;
  if (rs == 0) rc = 0;
  else if (lai == 0) rc = HUGE_RESIST;
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

// penman.c:96-145 — Penman-Monteith (mm/day).  The parts that depend only on (tair, elevation) are split out so that
// callers evaluating several resistances at one temperature (transpiration, pot-evap) pay for the exponentials once.
struct PenmanBase { double slope, lv, gamma, r_air; };
VIC_DEV PenmanBase penman_base(double tair, double elevation) {
  PenmanBase b;
  b.slope = svp_slope(tair);
  double h = 287 / 9.81 * ((tair + 273.15) + 0.5 * elevation * -0.006);
  double pz = 101300 * exp(-elevation / h);
  b.lv = 2501000 - 2361 * tair;
  b.gamma = 1628.6 * pz / b.lv;
  b.r_air = 0.003486 * pz / (275 + tair);
  return b;
}
VIC_DEV double penman_eval(const PenmanBase& b, double rad, double vpd, double ra, double rc, double rarc) {
  double evap = (b.slope * rad + b.r_air * 1013 * vpd / ra) / (b.lv * (b.slope + b.gamma * (1 + (rc + rarc) / ra))) * SEC_PER_DAY;
  if (vpd >= 0.0 && evap < 0.0) evap = 0.0;
  return evap;
}
VIC_DEV double penman(double tair, double elevation, double rad, double vpd, double ra, double rc, double rarc) {
  return penman_eval(penman_base(tair, elevation), rad, vpd, ra, rc, rarc);
}

// StabilityCorrection.c:44-81
VIC_DEV double stability_correction(double Z, double d, double TSurf, double Tair, double Wind, double Z0) {
  double corr = 1.0;
  const double RiCr = 0.2;
  if (TSurf != Tair) {
    double Ri = G_GRAV * (Tair - TSurf) * (Z - d) / (((Tair + 273.15) + (TSurf + 273.15)) / 2.0 * Wind * Wind);
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

// ------------------------------------------------------------------------------------------------
// Root finder: Brent (1973) with the reference's bracket-expansion and ERROR-aware search
// (root_brent.c:97-337), restructured for SIMT as a state machine with ONE residual evaluation site.
// The reference calls the residual from eight places; inlining a residual that itself contains a soil
// profile solve eight times is not an option on a GPU, and a single site is also what keeps the lanes
// of a wavefront in the same instruction stream: every lane evaluates its own next abscissa in lock
// step, finished lanes idle under the exec mask until the wave's last lane is done (the while
// condition is the wave-level "any lane still active").  The sequence of abscissae each lane
// evaluates is exactly the reference's.
// ------------------------------------------------------------------------------------------------
struct Brent {
  enum Phase : int {
    EVAL_A0, EVAL_B0, SEARCH0, EXP_A, EXP_B, EXP_ONE, SEARCH1, MAIN, DONE
  };
  double a, b, c, d, e, fa, fb, fc, last_bad, last_good, x, result;
  int phase, which_err, i, j, k;

  static constexpr int MAXTRIES = 5, MAXITER = 1000;
  static constexpr double MACHEPS = 3e-8, TSTEP = 10, TTOL = 1e-7;

  VIC_DEV void start(double lower, double upper) {
    a = lower; b = upper; c = 0; d = 0; e = 0; fa = fb = fc = 0; last_bad = last_good = 0;
    which_err = 0; i = j = k = 0; result = ERROR_VAL;
    phase = EVAL_A0; x = a;
  }
  VIC_DEV void fail() { result = ERROR_VAL; phase = DONE; }

  // body of the main loop up to (not including) the evaluation at the new b (root_brent.c:258-322).  Written with
  // selects instead of branches: every lane of a wave executes the same instructions whichever of bisection, secant
  // or inverse quadratic interpolation it takes (the unused quotients are computed and dropped), and each selected
  // value is produced by the reference's own sequence of operations.
  VIC_DEV void main_prestep() {
    const bool c1 = fb * fc > 0;
    const double ba = b - a;
    c = c1 ? a : c; fc = c1 ? fa : fc; d = c1 ? ba : d; e = c1 ? ba : e;
    const bool c2 = fabs(fc) < fabs(fb);
    {
      const double oa = a, ob = b, oc = c, ofa = fa, ofb = fb, ofc = fc;
      a = c2 ? ob : oa; b = c2 ? oc : ob; c = c2 ? ob : oc;
      fa = c2 ? ofb : ofa; fb = c2 ? ofc : ofb; fc = c2 ? ofb : ofc;
    }
    const double tol = 2 * MACHEPS * fabs(b) + TTOL;
    const double m = 0.5 * (c - b);
    if (fabs(m) <= tol || fb == 0) { result = b; phase = DONE; return; }
    const bool bisect = fabs(e) < tol || fabs(fa) <= fabs(fb);
    const double s = fb / fa, q1 = fa / fc, r = fb / fc;
    const bool secant = (a == c);
    double p = secant ? 2 * m * s : s * (2 * m * q1 * (q1 - r) - (b - a) * (r - 1));
    double q = secant ? 1 - s : (q1 - 1) * (r - 1) * (s - 1);
    const bool ppos = p > 0;
    q = ppos ? -q : q;
    p = ppos ? p : -p;
    const double s2 = e;
    const bool accept = !bisect && (2 * p) < (3 * m * q - fabs(tol * q)) && p < fabs(0.5 * s2 * q);
    const double pq = p / q;
    e = accept ? d : m;
    d = accept ? pq : m;
    a = b; fa = fb;
    b += (fabs(d) > tol) ? d : ((m > 0) ? tol : -tol);
    phase = MAIN; x = b;
  }

  // consume the residual at x and pick the next abscissa.  The bracket test (root_brent.c:183) and the main-loop body
  // exist once: a wave whose lanes sit in different phases executes them together instead of one copy per phase.
  VIC_DEV void advance(double fx) {
    int act = 0;   // 1: bracket test, 2: main-loop body
    switch (phase) {
      case EVAL_A0: fa = fx; phase = EVAL_B0; x = b; break;
      case EVAL_B0:
        fb = fx;
        if (fa == ERROR_VAL && fb == ERROR_VAL) { fail(); break; }                     // :129-132
        if (fa == ERROR_VAL || fb == ERROR_VAL) {                                      // :136-150
          if (fa == ERROR_VAL) { which_err = -1; last_bad = a; last_good = b; }
          else { which_err = 1; last_good = a; last_bad = b; }
          c = 0.5 * (last_bad + last_good);
          k = 0; phase = SEARCH0; x = c;
        } else { j = 0; act = 1; }
        break;
      case SEARCH0:                                                                    // :152-175
        fc = fx;
        if (fc == ERROR_VAL && k < MAXITER) { last_bad = c; c = 0.5 * (last_bad + last_good); k++; x = c; break; }
        if (fc == ERROR_VAL) { fail(); break; }
        if (which_err == -1) { a = c; fa = fc; } else { b = c; fb = fc; }
        j = 0; act = 1;
        break;
      case EXP_A: fa = fx; phase = EXP_B; x = b; break;                                // :186-189
      case EXP_B: fb = fx; j++; act = 1; break;
      case EXP_ONE:                                                                    // :192-215
        if (which_err == -1) { fb = fx; if (fb == ERROR_VAL) { fail(); break; } last_good = a; }
        else { fa = fx; if (fa == ERROR_VAL) { fail(); break; } last_good = b; }
        c = 0.5 * (last_good + last_bad);
        k = 0; phase = SEARCH1; x = c;
        break;
      case SEARCH1:                                                                    // :216-238
        fc = fx;
        if (fc == ERROR_VAL && k < MAXITER) { last_bad = c; c = 0.5 * (last_bad + last_good); k++; x = c; break; }
        if (fc == ERROR_VAL) { fail(); break; }
        if (which_err == -1) { a = c; fa = fc; } else { b = c; fb = fc; }
        j++; act = 1;
        break;
      case MAIN:                                                                       // :323-332
        fb = fx;
        if (fb == ERROR_VAL) { fail(); break; }
        i++;
        if (i >= MAXITER) { fail(); break; }
        act = 2;
        break;
      default: break;
    }
    if (act == 1) {                      // bracket-expansion loop head (root_brent.c:183)
      if ((fa * fb) >= 0 && j < MAXTRIES) {
        if (which_err == 0) { a -= TSTEP; b += TSTEP; phase = EXP_A; x = a; }
        else if (which_err == -1) { b += TSTEP; phase = EXP_ONE; x = b; }
        else { a -= TSTEP; phase = EXP_ONE; x = a; }
      } else if ((fa * fb) >= 0) fail();                                 // :244-248
      else { fc = fb; i = 0; act = 2; }
    }
    if (act == 2) main_prestep();
  }
};

// root_brent for a residual that never returns the -999 error sentinel (the frozen-node residual, soil_thermal_eqn.c,
// has no error return): the bisection searches toward a valid side (root_brent.c:129-177, 192-238) are unreachable, so
// the state is a, b, c, d, e, fa, fb, fc and three counters, and one step is branch-light.  Where the reference would
// misread a residual of exactly -999.0 as an error flag this treats it as the number it is.  Same operations otherwise,
// except the quotients of the interpolation step (below).
struct BrentLean {
  enum Phase : int { EVAL_A0, EVAL_B0, EXP_A, EXP_B, MAIN, DONE, FAILED };      // DONE: the root is b
  double a, b, c, d, e, fa, fb, fc, x;
  int phase, i, j;
  VIC_DEV void start(double lower, double upper) {
    a = lower; b = upper; c = 0; d = 0; e = 0; fa = fb = fc = 0;
    i = j = 0;
    phase = EVAL_A0; x = a;
  }
  VIC_DEV void fail() { phase = FAILED; }
  VIC_DEV bool finished() const { return phase >= DONE; }
  VIC_DEV void advance(double fx) {
    if (phase == EVAL_A0 || phase == EXP_A) {              // residual at the lower end, next: the upper end
      fa = fx; x = b;
      phase = (phase == EVAL_A0) ? (int)EVAL_B0 : (int)EXP_B;
      return;
    }
    fb = fx;
    if (phase == MAIN) {                                   // root_brent.c:323-332
      i++;
      if (i >= Brent::MAXITER) { fail(); return; }
    } else {                                               // both ends evaluated: bracket test, root_brent.c:183-248
      j = (phase == EVAL_B0) ? 0 : j + 1;
      if ((fa * fb) >= 0) {
        if (j < Brent::MAXTRIES) { a -= Brent::TSTEP; b += Brent::TSTEP; phase = EXP_A; x = a; }
        else fail();
        return;
      }
      fc = fb; i = 0;
    }
    // main-loop body, root_brent.c:258-322 (see Brent::main_prestep)
    const bool c1 = fb * fc > 0;
    const double ba = b - a;
    c = c1 ? a : c; fc = c1 ? fa : fc; d = c1 ? ba : d; e = c1 ? ba : e;
    const bool c2 = fabs(fc) < fabs(fb);
    {
      const double oa = a, ob = b, oc = c, ofa = fa, ofb = fb, ofc = fc;
      a = c2 ? ob : oa; b = c2 ? oc : ob; c = c2 ? ob : oc;
      fa = c2 ? ofb : ofa; fb = c2 ? ofc : ofb; fc = c2 ? ofb : ofc;
    }
    const double tol = 2 * Brent::MACHEPS * fabs(b) + Brent::TTOL;
    const double m = 0.5 * (c - b);
    if (fabs(m) <= tol || fb == 0) { phase = DONE; return; }
    const bool bisect = fabs(e) < tol || fabs(fa) <= fabs(fb);
#ifdef VIC_REFERENCE_DIVISIONS
    const double s = fb / fa, q1 = fa / fc, r = fb / fc;
    const bool secant = (a == c);
    double p = secant ? 2 * m * s : s * (2 * m * q1 * (q1 - r) - (b - a) * (r - 1));
    double q = secant ? 1 - s : (q1 - 1) * (r - 1) * (s - 1);
#else
    // The interpolation step p/q with s = fb/fa, q1 = fa/fc, r = fb/fc written over a common denominator: p and q are the
    // reference's multiplied by fa (secant) or fa*fc^2 (inverse quadratic), which leaves the sign normalisation, the two
    // acceptance tests (both homogeneous in p, q) and the quotient unchanged and needs one division instead of four.
    // The step only proposes the next trial point; bracket logic and stopping test are untouched (-4 % step time
    // together with SoilThermalEqn::eval).
    const bool secant = (a == c);
    double p = secant ? 2 * m * fb : fb * (2 * m * fa * (fa - fb) - (b - a) * (fb - fc) * fc);
    double q = secant ? fa - fb : (fa - fc) * (fb - fc) * (fb - fa);
#endif
    const bool ppos = p > 0;
    q = ppos ? -q : q;
    p = ppos ? p : -p;
    const double s2 = e;
    const bool accept = !bisect && (2 * p) < (3 * m * q - fabs(tol * q)) && p < fabs(0.5 * s2 * q);
    const double pq = p / q;
    e = accept ? d : m;
    d = accept ? pq : m;
    a = b; fa = fb;
    b += (fabs(d) > tol) ? d : ((m > 0) ? tol : -tol);
    phase = MAIN; x = b;
  }
};

template <class F>
VIC_DEV double root_brent(double lower, double upper, F& f) {
  Brent st;
  st.start(lower, upper);
  while (st.phase != Brent::DONE) {
    double fx = f(st.x);
    st.advance(fx);
  }
  return st.result;
}

// ------------------------------------------------------------------------------------------------ aerodynamics

// calc_veg_params.c:26-41
VIC_DEV double calc_veg_height(double displacement, double L) {
  double X = COEF_DRAG * L;
  return displacement / (1.1 * log(1 + pow(X, 0.25)));
}

// CalcAerodynamic.c:64-271; returns false on the trunk-space error (:214-217)
VIC_DEV bool calc_aerodynamic(bool overstory, double Height, double Trunk, double Z0_SNOW, double Z0_SOIL, double n,
                              Vc& ra, Vc& U, Vc& disp, Vc& zref, Vc& z0) {
  const double K2 = VON_K * VON_K;
  double tmp_wind = U.v[SNOW_FREE];
  if (!overstory) {
    double Z0_Lower = z0.v[SNOW_FREE], d_Lower = disp.v[SNOW_FREE];
    double l2 = log((2. + Z0_Lower) / Z0_Lower);
    double lr = log((zref.v[SNOW_FREE] - d_Lower) / Z0_Lower);
    U.v[SNOW_FREE] = l2 / lr;
    ra.v[SNOW_FREE] = l2 * lr / K2;
    zref.v[CANOPY] = zref.v[SNOW_FREE]; z0.v[CANOPY] = z0.v[SNOW_FREE]; disp.v[CANOPY] = disp.v[SNOW_FREE];
    U.v[CANOPY] = U.v[SNOW_FREE]; ra.v[CANOPY] = ra.v[SNOW_FREE];
    zref.v[SNOW_COVERED] = zref.v[SNOW_FREE];
    z0.v[SNOW_COVERED] = Z0_SNOW;
    disp.v[SNOW_COVERED] = 0.;
    double ls2 = log((2. + Z0_SNOW) / Z0_SNOW), lsr = log(zref.v[SNOW_COVERED] / Z0_SNOW);
    U.v[SNOW_COVERED] = ls2 / lsr;
    ra.v[SNOW_COVERED] = ls2 * lsr / K2;
    zref.v[SNOW_COVERED] = 2. + Z0_SNOW;
    zref.v[GLACIER_SURF] = zref.v[SNOW_FREE];
    z0.v[GLACIER_SURF] = Z0_Lower;
    disp.v[GLACIER_SURF] = 0.;
    double lgr = log(zref.v[GLACIER_SURF] / Z0_Lower);
    U.v[GLACIER_SURF] = l2 / lgr;
    ra.v[GLACIER_SURF] = l2 * lgr / K2;
    zref.v[GLACIER_SURF] = 2. + Z0_Lower;
  } else {
    double Z0_Upper = z0.v[SNOW_FREE], d_Upper = disp.v[SNOW_FREE];
    double Z0_Lower = Z0_SOIL, d_Lower = 0;
    double Zw = 1.5 * Height - 0.5 * d_Upper;
    double Zt = Trunk * Height;
    double zr = zref.v[SNOW_FREE];
    if (Zt < (Z0_Lower + d_Lower)) return false;
    double lru = log((zr - d_Upper) / Z0_Upper);
    ra.v[CANOPY] = lru / K2 * (Height / (n * (Zw - d_Upper)) * (exp(n * (1 - (d_Upper + Z0_Upper) / Height)) - 1)
                               + (Zw - Height) / (Zw - d_Upper) + log((zr - d_Upper) / (Zw - d_Upper)));
    double Uw = log((Zw - d_Upper) / Z0_Upper) / lru;
    double Uh = Uw - (1 - (Height - d_Upper) / (Zw - d_Upper)) / lru;
    U.v[CANOPY] = Uh * exp(n * ((Z0_Upper + d_Upper) / Height - 1.));
    double Ut = Uh * exp(n * (Zt / Height - 1.));
    double l2 = log((2. + Z0_Lower) / Z0_Lower), ltz = log(Zt / Z0_Lower);
    U.v[SNOW_FREE] = Ut * l2 / ltz;
    ra.v[SNOW_FREE] = l2 * ltz / (K2 * Ut);
    if (Zt > (2. + Z0_SNOW)) {
      double ls2 = log((2. + Z0_SNOW) / Z0_SNOW), lts = log(Zt / Z0_SNOW);
      U.v[SNOW_COVERED] = Ut * ls2 / lts;
      ra.v[SNOW_COVERED] = ls2 * lts / (K2 * Ut);
    } else if (Height > (2. + Z0_SNOW)) {
      double lts = log(Zt / Z0_SNOW);
      U.v[SNOW_COVERED] = Uh * exp(n * ((2. + Z0_SNOW) / Height - 1.));
      ra.v[SNOW_COVERED] = lts * lts / (K2 * Ut)
          + Height * lru / (n * K2 * (Zw - d_Upper)) * (exp(n * (1 - Zt / Height)) - exp(n * (1 - (Z0_SNOW + 2.) / Height)));
    } else {
      double lts = log(Zt / Z0_SNOW);
      U.v[SNOW_COVERED] = Uh;
      ra.v[SNOW_COVERED] = lts * lts / (K2 * Ut) + Height * lru / (n * K2 * (Zw - d_Upper)) * (exp(n * (1 - Zt / Height)) - 1);
    }
    zref.v[CANOPY] = zref.v[SNOW_FREE]; z0.v[CANOPY] = z0.v[SNOW_FREE]; disp.v[CANOPY] = disp.v[SNOW_FREE];
    zref.v[SNOW_FREE] = 2. + Z0_Lower; z0.v[SNOW_FREE] = Z0_Lower; disp.v[SNOW_FREE] = d_Lower;
    zref.v[SNOW_COVERED] = 2. + Z0_SNOW; z0.v[SNOW_COVERED] = Z0_SNOW; disp.v[SNOW_COVERED] = 0.;
    zref.v[GLACIER_SURF] = 2. + Z0_Lower; z0.v[GLACIER_SURF] = Z0_Lower; disp.v[GLACIER_SURF] = 0.;
  }
  if (tmp_wind > 0.) {
    U.v[SNOW_FREE] *= tmp_wind;
    ra.v[SNOW_FREE] /= tmp_wind;
#pragma unroll
    for (int k = 1; k < NCASE; k++)
      if (!isnan(U.v[k])) { U.v[k] *= tmp_wind; ra.v[k] /= tmp_wind; }
  } else {
    U.v[SNOW_FREE] *= tmp_wind;
    ra.v[SNOW_FREE] = HUGE_RESIST;
#pragma unroll
    for (int k = 1; k < NCASE; k++) {
      if (!isnan(U.v[k])) U.v[k] *= tmp_wind;
      ra.v[k] = HUGE_RESIST;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ soil thermal properties

// soil_conduction.c:7-105 (Johansen)
VIC_DEV double soil_conductivity(double moist, double Wu, double soil_dens_min, double bulk_dens_min, double quartz,
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

// The same with the layer's constants taken from the derived rows of the cell table: Kdry, Ks^(1-porosity),
// Kw^porosity and the porosity are computed once per domain by soil_conductivity_layer_constants with the expressions
// and the pow of soil_conductivity itself, so the result is identical; three of the five pow calls and the quartz branch
// leave the per-node, per-step path.
struct SoilKLayer { double Kdry, KsP, KwP, porosity; };
VIC_DEV SoilKLayer soil_conductivity_layer_constants(double soil_dens_min, double bulk_dens_min, double quartz, double soil_density,
                                                     double bulk_density, double organic) {
  const double Kw = 0.57, Kdry_org = 0.05, Ks_org = 0.25;
  SoilKLayer c;
  double Kdry_min = (0.135 * bulk_dens_min + 64.7) / (soil_dens_min - 0.947 * bulk_dens_min);
  c.Kdry = (1 - organic) * Kdry_min + organic * Kdry_org;
  c.porosity = 1.0 - bulk_density / soil_density;
  double Ks_min;
  if (quartz < .2) Ks_min = pow(7.7, quartz) * pow(3.0, 1.0 - quartz);
  else Ks_min = pow(7.7, quartz) * pow(2.2, 1.0 - quartz);
  const double Ks = (1 - organic) * Ks_min + organic * Ks_org;
  c.KsP = pow(Ks, 1.0 - c.porosity);
  c.KwP = pow(Kw, c.porosity);
  return c;
}
VIC_DEV double soil_conductivity_pre(double moist, double Wu, const SoilKLayer& c) {
  const double Ki = 2.2, Kw = 0.57;
  double K;
  if (moist > 0.) {
    double Sr = moist / c.porosity;
    double Ksat, Ke;
    if (Wu == moist) {
      Ksat = c.KsP * c.KwP;
      Ke = 0.7 * log10(Sr) + 1.0;
    } else {
      Ksat = c.KsP * pow(Ki, c.porosity - Wu) * pow(Kw, Wu);
      Ke = Sr;
    }
    K = (Ksat - c.Kdry) * Ke + c.Kdry;
    if (K < c.Kdry) K = c.Kdry;
  } else K = c.Kdry;
  return K;
}

// soil_conduction.c:108-139
VIC_DEV double volumetric_heat_capacity(double soil_fract, double water_fract, double ice_fract, double organic_fract) {
  double Cs = 2.0e6 * soil_fract * (1 - organic_fract);
  Cs += 2.7e6 * soil_fract * organic_fract;
  Cs += 4.2e6 * water_fract;
  Cs += 1.9e6 * ice_fract;
  Cs += 1.3e3 * (1. - (soil_fract + water_fract + ice_fract));
  return Cs;
}

// soil_conduction.c:830-863
// x^y for finite x > 0 as exp(y ln x): the freezing-point-depression curve is evaluated once per residual evaluation of
// every frozen-node Brent solve, i.e. several hundred times per HRU and step, and the library pow spends ~250
// instructions on corner cases that cannot occur here.  ln x follows the classic argument reduction
// x = 2^k (1 + f), s = f / (2 + f), ln(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-14 minimax R of
// fdlibm's e_log.c (error < 1 ulp); the result agrees with pow to a few ulp (|y ln x| < 40 here).
VIC_DEV double ln_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int k;
  double m = frexp(x, &k);                                  // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;                              // m in [sqrt(1/2), sqrt(2))
  k = lo ? k - 1 : k;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

VIC_DEV double pow_pos(double x, double y) {
  if (!(x > 0.0 && x < 1.0e300)) return pow(x, y);   // 0, negative, inf, NaN: the library's corner cases
  return exp(y * ln_pos(x));
}
// For callers that guarantee a finite x > 0 (SoilThermalEqn::eval: x = -Lf*T*const with T < 0): no library fallback in the
// Brent loop's code.
VIC_DEV double pow_pos_finite(double x, double y) { return exp(y * ln_pos(x)); }

// Approximations for the predictor phase of the frozen-node Newton iteration (vic_profile.hpp): the hardware's single-
// precision log2 / exp2 (v_log_f32, v_exp_f32; a handful of instructions against ~90 for exp(y ln x) in double) and a
// reciprocal with one Newton-Raphson refinement instead of the IEEE division sequence.  Nothing computed with them
// reaches a result: they only choose the point at which the double-precision iteration starts and scale its steps.
VIC_DEV double pow_pos_approx(double x, float y) {
#ifdef VIC_HOSTEMU
  return (double)exp2f(y * log2f((float)x));
#else
  return (double)__builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf((float)x));
#endif
}
VIC_DEV double rcp_refined(double d) {
#ifdef VIC_HOSTEMU
  return 1.0 / d;
#else
  const double r = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r, 1.0), r, r);
#endif
}

VIC_DEV double maximum_unfrozen_water(double T, double max_moist, double bubble, double expt) {
  double u;
  if (T <= 0) {
    u = max_moist * pow_pos((-LF * T) / 273.16 / (9.81 * bubble / 100.), -(2.0 / (expt - 3.0)));
    if (u > max_moist) u = max_moist;
    if (u < 0) u = 0;
  } else u = max_moist;
  return u;
}

VIC_DEV double linear_interp(double x, double lx, double ux, double ly, double uy) { return (x - lx) / (ux - lx) * (uy - ly) + ly; }

}  // namespace vic
