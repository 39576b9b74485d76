// vic_blowing.hpp — sublimation from blowing snow, options.BLOWING (device only, gfx950).
//
// CalcBlowingSnow.c:101-799 with the reference's compile-time switches as shipped (:49-53: Liston & Sturm mass flux,
// spatially variable wind, variable threshold, fetch dependence, computed probability of occurrence).  Called once per
// snow sub-step for HRUs without overstory (surface_fluxes.c:439-453) and for glacier HRUs (surface_fluxes_glac.c:260-274),
// never inside a root find.  One call walks ten wind-probability intervals; each may run a Newton/bisection solve for the
// shear velocity and two Romberg integrations over the suspension layer, a few thousand transcendentals per call in a
// data-dependent pattern -- so the function is kept OUT OF LINE: the kernels that call it stay as they are when the option
// is off (it is off in every BASELINE configuration), and when it is on its lanes diverge inside their own frame.
// Every expression keeps the reference's operation order.
#pragma once
#include "vic_math.hpp"

namespace vic {

namespace bs {
constexpr double PI_ = 3.1415927;             // vicNl_def.h:276 (mtclim_constants_vic.h:52 only defines PI when nobody has)
constexpr double MW = 18.0148e-3, RGAS = 8.3143, G_STD = 9.80665;        // mtclim_constants_vic.h:44-46
constexpr double KA = .0245187, CSALT = 0.68, KIN_VIS = 1.3e-5, MACHEPS = 1.0e-6, SETTLING = 0.3;   // CalcBlowingSnow.c:38-46
constexpr int MAX_ITER = 100, KPTS = 5, NUMINCS = 10;

struct Ctx { double es, Wind, ZO, EactAir, F, hsalt, phi_r, ushear; };

// concentration of turbulent suspended snow relative to the saltation layer (Kind 1992)
VIC_DEV double phi_t(double z, const Ctx& c) {
  const double temp = (0.5 * c.ushear * c.ushear) / (c.Wind * SETTLING);
  return c.phi_r * ((temp + 1.) * pow((z / c.hsalt), (-1. * SETTLING) / (VON_K * c.ushear)) - temp);
}
// CalcBlowingSnow.c:505-568 (TRANSPORT = false) and :775-799 (TRANSPORT = true): the two integrands of the suspension layer
template <bool TRANSPORT>
VIC_DEV double integrand(double z, const Ctx& c) {
  if (TRANSPORT) {
    const double u_z = c.ushear * log(z / c.ZO) / VON_K;
    return u_z * phi_t(z, c);
  }
  const double Rrz = 4.6e-5 * pow(z, -.258);
  const double ALPHAz = 4.08 + 12.6 * z;
  const double Mz = (4. / 3.) * PI_ * ICE_DENSITY * Rrz * Rrz * Rrz * (1. + (3. / ALPHAz) + (2. / (ALPHAz * ALPHAz)));
  const double Rmean = pow((3. * Mz) / (4. * PI_ * ICE_DENSITY), 1. / 3.);
  const double terminal_v = 1.1e7 * pow(Rmean, 1.8);
  const double fluctuat_v = 0.005 * pow(c.Wind, 1.36);
  const double Vtz = terminal_v + 3. * fluctuat_v * cos(PI_ / 4.);
  const double Re = 2. * Rmean * Vtz / KIN_VIS;
  const double Nu = 1.79 + 0.606 * pow(Re, 0.5);
  const double sigz = ((c.EactAir / c.es) - 1.) * (1.019 + .027 * log(z));
  const double dMdt = 2 * PI_ * Rmean * sigz * Nu / c.F;
  const double psi_t = dMdt / Mz;
  return psi_t * phi_t(z, c);
}

// Romberg integration (CalcBlowingSnow.c:312-422).  The reference keeps every level's estimate and extrapolates over the last
// KPTS of them with Numerical Recipes' polint; only that window is kept here.  The abscissae of the extrapolation are
// h = 1, 1/4, 1/16, ... and the target is 0, so polint's "nearest table entry" is always the last one and its correction
// path runs down the d column: written out for that case (xa strictly decreasing and positive, x = 0).
template <bool TRANSPORT>
VIC_DEV double qromb(const Ctx& c, double a, double b, bool& err) {
  double s[KPTS], h[KPTS];
#pragma unroll
  for (int i = 0; i < KPTS; i++) { s[i] = 0; h[i] = 0; }
  double hj = 1.0, last = 0.0;
  for (int j = 1; j <= MAX_ITER; j++) {
    double sj;
    if (j == 1) sj = 0.5 * (b - a) * (integrand<TRANSPORT>(a, c) + integrand<TRANSPORT>(b, c));
    else {
      int it = 1;
      for (int k = 1; k < j - 1; k++) it <<= 1;
      const double tnm = it;
      const double del = (b - a) / tnm;
      double x = a + 0.5 * del, sum = 0.0;
      for (int k = 1; k <= it; k++, x += del) sum += integrand<TRANSPORT>(x, c);
      sj = 0.5 * (last + (b - a) * sum / tnm);
    }
    last = sj;
#pragma unroll
    for (int i = 0; i < KPTS - 1; i++) { s[i] = s[i + 1]; h[i] = h[i + 1]; }
    s[KPTS - 1] = sj; h[KPTS - 1] = hj;
    if (j >= KPTS) {
      // polint(&h[j-K], &s[j-K], K, 0.0, &ss, &dss) with ns = K
      double cc[KPTS], dd[KPTS];
#pragma unroll
      for (int i = 0; i < KPTS; i++) { cc[i] = s[i]; dd[i] = s[i]; }
      double ss = s[KPTS - 1], dss = 0;
      bool bad = false;
#pragma unroll
      for (int m = 1; m < KPTS; m++) {
#pragma unroll
        for (int i = 0; i < KPTS - m; i++) {
          const double ho = h[i] - 0.0, hp = h[i + m] - 0.0, w = cc[i + 1] - dd[i];
          double den = ho - hp;
          if (den == 0.0) bad = true;
          den = w / den;
          dd[i] = hp * den;
          cc[i] = ho * den;
        }
        dss = dd[KPTS - 1 - m];
        ss += dss;
      }
      if (bad) { err = true; return 0.0; }
      if (fabs(dss) <= MACHEPS * fabs(ss)) return ss;
    }
    hj = 0.25 * hj;
  }
  err = true;                                          // "Too many steps in routine qromb"
  return 0.0;
}

// CalcBlowingSnow.c:483-487
VIC_DEV void get_shear(double x, double& f, double& df, double Ur, double Zr) {
  f = log(2. * G_STD * Zr / .12) + log(1 / (x * x)) - VON_K * Ur / x;
  df = VON_K * Ur / (x * x) - 2. / x;
}

// CalcBlowingSnow.c:424-481
VIC_DEV double rtnewt(double x1, double x2, double acc, double Ur, double Zr, bool& err) {
  double df, dx, dxold, f, fh, fl, xh, xl, rts;
  get_shear(x1, fl, df, Ur, Zr);
  get_shear(x2, fh, df, Ur, Zr);
  if ((fl > 0.0 && fh > 0.0) || (fl < 0.0 && fh < 0.0)) { err = true; return 0.0; }    // the reference stops the program here
  if (fl == 0.0) return x1;
  if (fh == 0.0) return x2;
  if (fl < 0.0) { xl = x1; xh = x2; } else { xh = x1; xl = x2; }
  rts = 0.5 * (x1 + x2);
  dxold = fabs(x2 - x1);
  dx = dxold;
  get_shear(rts, f, df, Ur, Zr);
  for (int j = 1; j <= MAX_ITER; j++) {
    if ((((rts - xh) * df - f) * ((rts - x1) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      rts = xl + dx;
      if (xl == rts) return rts;
    } else {
      dxold = dx;
      dx = f / df;
      const double temp = rts;
      rts -= dx;
      if (temp == rts) return rts;
    }
    if (fabs(dx) < acc) return rts;
    get_shear(rts, f, df, Ur, Zr);
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return 0.0;                                          // "Maximum number of iterations exceeded in rtnewt"
}

// CalcBlowingSnow.c:575-605
VIC_DEV double get_prob(double Tair, double Age, double SurfaceLiquidWater, double U10) {
  double mean_u, sigma;
  if (SurfaceLiquidWater < 0.001) {
    mean_u = 11.2 + 0.365 * Tair + 0.00706 * Tair * Tair + 0.9 * log(Age);
    sigma = 4.3 + 0.145 * Tair + 0.00196 * Tair * Tair;
  } else { mean_u = 21.; sigma = 7.; }
  double prob = 1. / (1. + exp(sqrt(PI_) * (mean_u - U10) / sigma));
  if (prob < 0.0) prob = 0.0;
  if (prob > 1.0) prob = 1.0;
  return prob;
}

// CalcBlowingSnow.c:636-667 after the threshold of :607-633
VIC_DEV bool shear_stress(double U10, double ZO, double& ushear, double& Zo_salt, double utshear) {
  const double umin = utshear, umax = VON_K * U10, xacc = 0.10 * umin;
  double fl, fh, df;
  bool err = false;
  get_shear(umin, fl, df, U10, 10.);
  get_shear(umax, fh, df, U10, 10.);
  if (fl < 0.0 && fh < 0.0) return false;              // "Solution in rtnewt surpasses upper boundary": the reference exits
  if (fl > 0.0 && fh > 0.0) {
    Zo_salt = ZO;
    ushear = VON_K * U10 / log(10. / ZO);
  } else {
    ushear = rtnewt(umin, umax, xacc, U10, 10., err);
    Zo_salt = 0.12 * ushear * ushear / (2. * G_STD);
  }
  return !err;
}

// CalcBlowingSnow.c:669-753
VIC_DEV double calc_sub_flux(double EactAir, double es, double AirDens, double utshear, double ushear, float fe, double U10, double Zo_salt,
                             double F, bool& err) {
  double SubFlux = 0.0;
  const double particle = utshear * 2.8;
  double Qsalt = (CSALT * AirDens / G_STD) * (utshear / ushear) * (ushear * ushear - utshear * utshear);
  Qsalt *= (1. + (500. / (3. * fe)) * (exp(-3. * fe / 500.) - 1.));
  const double hsalt = 0.08436 * pow(ushear, 1.27);
  const double phi_s = Qsalt / (hsalt * particle);
  const double T = 0.5 * (ushear * ushear) / (U10 * SETTLING);
  const double ztop = hsalt * pow(T / (T + 1.), (VON_K * ushear) / (-1. * SETTLING));
  Ctx c;
  c.es = es; c.Wind = U10; c.ZO = Zo_salt; c.EactAir = EactAir; c.F = F; c.hsalt = hsalt; c.phi_r = phi_s; c.ushear = ushear;
  if (!(EactAir >= es)) {
    const double psi_s = integrand<false>(hsalt / 2., c);
    SubFlux = phi_s * psi_s * hsalt;
    SubFlux += qromb<false>(c, hsalt, ztop, err);
  }
  // the transport integral feeds snow.transport only (not an output of the path), but its failure is a failure of the call
  (void)qromb<true>(c, hsalt, ztop, err);
  return SubFlux;
}
}  // namespace bs

// CalcBlowingSnow.c:101-310: kg m-2 s-1 (negative: loss), ERROR_VAL where the reference returns ERROR or stops
__device__ __attribute__((noinline)) double calc_blowing_snow(double Dt, double Tair, int LastSnow, double SurfaceLiquidWater, double Wind,
                                                               double Ls, double AirDens, double EactAir, double ZO, double snowdepth,
                                                               float lag_one, float sigma_slope, int isArtificialBareSoil, float fe,
                                                               double displacement, double roughness) {
  using namespace bs;
  const double Age = LastSnow * (Dt);
  const double es = svp(Tair);
  const double Tk = Tair + KELVIN;
  const double Ros = 0.622 * es / (287 * Tk);
  const double Diffusivity = (2.06e-5) * pow(Tk / 273., 1.75);
  double F = (Ls / (KA * Tk)) * (Ls * MW / (RGAS * Tk) - 1.);
  F += 1. / (Diffusivity * Ros);
  const double wind10 = Wind * log(10. / ZO) / log((2 + ZO) / ZO);
  if (isArtificialBareSoil) { fe = 1500; sigma_slope = .0002; }
  const double ratio = (2.44 - (0.43) * lag_one) * sigma_slope;
  const double sigma_w = wind10 * ratio;
  const double Uo = wind10;
  const double hv = (3. / 2.) * displacement;
  const double Nd = (4. / 3.) * (roughness / displacement);
  double Total = 0.0;
  const double area = 1. / NUMINCS;
  if (snowdepth > 0.0) {
    const bool spatial = sigma_w != 0.;
    const int nint = spatial ? NUMINCS : 1;
    for (int p = 0; p < nint; p++) {
      double U10 = Uo;
      if (spatial) {
        double lower = 0.0, upper = 0.0;
        if (p == 0) { lower = -9999; upper = Uo + sigma_w * log(2. * (p + 1) * area); }
        else if (p > 0 && p < NUMINCS / 2) { lower = Uo + sigma_w * log(2. * (p) * area); upper = Uo + sigma_w * log(2. * (p + 1) * area); }
        else if (p < (NUMINCS - 1) && p >= NUMINCS / 2) {
          lower = Uo - sigma_w * log(2. - 2. * (p * area));
          upper = Uo - sigma_w * log(2. - 2. * ((p + 1.) * area));
        } else if (p == NUMINCS - 1) { lower = Uo - sigma_w * log(2. - 2. * (p * area)); upper = 9999; }
        if (lower > upper) lower = upper;
        if (lower >= Uo)
          U10 = -0.5 * ((upper + sigma_w) * exp((-1. / sigma_w) * (upper - Uo)) - (lower + sigma_w) * exp((-1. / sigma_w) * (lower - Uo))) / area;
        else if (upper <= Uo)
          U10 = 0.5 * ((upper - sigma_w) * exp((1. / sigma_w) * (upper - Uo)) - (lower - sigma_w) * exp((1. / sigma_w) * (lower - Uo))) / area;
        else return ERROR_VAL;
        if (U10 < 0.4) U10 = .4;
        if (U10 > 25.) U10 = 25.;
      }
      const double Uveg = (snowdepth < hv) ? U10 / sqrt(1. + 170 * Nd * (hv - snowdepth)) : U10;
      const double prob = get_prob(Tair, Age, SurfaceLiquidWater, Uveg);
      const double ut10 = (SurfaceLiquidWater < 0.001) ? 9.43 + .18 * Tair + .0033 * Tair * Tair : 9.9;
      const double utshear = VON_K * ut10 / log(10. / ZO);
      double ushear, Zo_salt;
      if (!shear_stress(U10, ZO, ushear, Zo_salt, utshear)) return ERROR_VAL;
      double SubFlux = 0.0;
      if (ushear > utshear) {
        bool err = false;
        SubFlux = calc_sub_flux(EactAir, es, AirDens, utshear, ushear, fe, U10, Zo_salt, F, err);
        if (err) return ERROR_VAL;
      }
      if (spatial) Total += (1. / NUMINCS) * SubFlux * prob;
      else Total = SubFlux * prob;
    }
  }
  if (Total < -.00005) Total = -.00005;
  return Total;
}

}  // namespace vic
