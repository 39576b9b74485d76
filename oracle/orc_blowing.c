/* orc_blowing.c -- TEST INFRASTRUCTURE (CPU oracle): sublimation flux from blowing snow (options.BLOWING).
 *
 * Restates CalcBlowingSnow.c:101-799 of the reference with its compile-time switches as shipped (CalcBlowingSnow.c:49-53:
 * SIMPLE 0, SPATIAL_WIND 1, VAR_THRESHOLD 1, FETCH 1, CALC_PROB 1): the 10-m wind of the cell is spread over ten equal-
 * probability intervals of a Laplace distribution (:199-246); for each interval the probability of blowing snow (Li &
 * Pomeroy 1997, :575-605), the threshold and the actual shear velocity during saltation (:607-667, Newton/bisection of
 * :424-481), and -- when the latter exceeds the former -- the sublimation of the saltation layer plus the Romberg integral
 * of the suspension layer (:669-753, :312-422).  The transport terms the reference also integrates (:729-741) feed
 * snow.transport only, which nothing on the path or in put_data reads; they are evaluated here all the same because
 * a failure inside them (qromb's iteration limit) is a failure of the call.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/. */
#include <math.h>
#include "orc.h"

#define BS_PI 3.1415927                 /* vicNl_def.h:276 (mtclim_constants_vic.h:52 only defines PI when it is not defined yet) */
#define BS_MW 18.0148e-3                /* mtclim_constants_vic.h:44-46 */
#define BS_R 8.3143
#define BS_G_STD 9.80665
#define BS_KA .0245187                  /* CalcBlowingSnow.c:38-45 */
#define BS_CSALT 0.68
#define BS_KIN_VIS 1.3e-5
#define BS_MAX_ITER 100
#define BS_K 5
#define BS_MACHEPS 1.0e-6
#define BS_SETTLING 0.3
#define BS_NUMINCS 10

typedef struct { double es, Wind, AirDens, ZO, EactAir, F, hsalt, phi_r, ushear, Zrh; } bs_ctx;

/* concentration of turbulent suspended snow relative to the saltation layer (Kind 1992), shared by the two integrands */
static double bs_phi_t(double z, const bs_ctx *c) {
  double temp = (0.5 * c->ushear * c->ushear) / (c->Wind * BS_SETTLING);
  return c->phi_r * ((temp + 1.) * pow((z / c->hsalt), (-1. * BS_SETTLING) / (ORC_VON_K * c->ushear)) - temp);
}

/* CalcBlowingSnow.c:505-568: sublimation rate at height z, kg m-3 s-1 */
static double bs_sub_with_height(double z, const bs_ctx *c) {
  double Rrz = 4.6e-5 * pow(z, -.258);
  double ALPHAz = 4.08 + 12.6 * z;
  double Mz = (4. / 3.) * BS_PI * ORC_ICE_DENSITY * Rrz * Rrz * Rrz * (1. + (3. / ALPHAz) + (2. / (ALPHAz * ALPHAz)));
  double Rmean = pow((3. * Mz) / (4. * BS_PI * ORC_ICE_DENSITY), 1. / 3.);
  double terminal_v = 1.1e7 * pow(Rmean, 1.8);
  double fluctuat_v = 0.005 * pow(c->Wind, 1.36);
  double Vtz = terminal_v + 3. * fluctuat_v * cos(BS_PI / 4.);
  double Re = 2. * Rmean * Vtz / BS_KIN_VIS;
  double Nu = 1.79 + 0.606 * pow(Re, 0.5);
  double sigz = ((c->EactAir / c->es) - 1.) * (1.019 + .027 * log(z));
  double dMdt = 2 * BS_PI * Rmean * sigz * Nu / c->F;
  double psi_t = dMdt / Mz;
  return psi_t * bs_phi_t(z, c);
}

/* CalcBlowingSnow.c:775-799: transport rate at height z, kg m-2 s-1 */
static double bs_transport_with_height(double z, const bs_ctx *c) {
  double u_z = c->ushear * log(z / c->ZO) / ORC_VON_K;
  return u_z * bs_phi_t(z, c);
}

typedef double (*bs_fn)(double, const bs_ctx *);

/* polint of Numerical Recipes as the reference has it (CalcBlowingSnow.c:345-385), 1-based arrays of n points */
static int bs_polint(const double *xa, const double *ya, int n, double x, double *y, double *dy) {
  double c[BS_K + 2], d[BS_K + 2];
  int i, m, ns = 1;
  double dif = fabs(x - xa[1]);
  for (i = 1; i <= n; i++) {
    double dift = fabs(x - xa[i]);
    if (dift < dif) { ns = i; dif = dift; }
    c[i] = ya[i];
    d[i] = ya[i];
  }
  *y = ya[ns--];
  for (m = 1; m < n; m++) {
    for (i = 1; i <= n - m; i++) {
      double ho = xa[i] - x, hp = xa[i + m] - x, w = c[i + 1] - d[i], den = ho - hp;
      if (den == 0.0) return -1;                       /* nrerror("Error in routine polint") */
      den = w / den;
      d[i] = hp * den;
      c[i] = ho * den;
    }
    *dy = (2 * ns < (n - m)) ? c[ns + 1] : d[ns--];
    *y += *dy;
  }
  return 0;
}

/* Romberg integration of f over [a, b] (CalcBlowingSnow.c:312-343 with the trapezoid refinement of :387-422); *err is set
 * where the reference calls nrerror (iteration limit, degenerate extrapolation) */
static double bs_qromb(bs_fn f, const bs_ctx *c, double a, double b, int *err) {
  double s[BS_MAX_ITER + 2], h[BS_MAX_ITER + 3], last = 0.0;
  int j;
  h[1] = 1.0;
  for (j = 1; j <= BS_MAX_ITER; j++) {
    if (j == 1) s[j] = 0.5 * (b - a) * (f(a, c) + f(b, c));
    else {
      int it = 1, k;
      double tnm, del, x, sum = 0.0;
      for (k = 1; k < j - 1; k++) it <<= 1;
      tnm = it;
      del = (b - a) / tnm;
      x = a + 0.5 * del;
      for (k = 1; k <= it; k++, x += del) sum += f(x, c);
      s[j] = 0.5 * (last + (b - a) * sum / tnm);
    }
    last = s[j];
    if (j >= BS_K) {
      double ss, dss;
      if (bs_polint(&h[j - BS_K], &s[j - BS_K], BS_K, 0.0, &ss, &dss) != 0) { *err = 1; return 0.0; }
      if (fabs(dss) <= BS_MACHEPS * fabs(ss)) return ss;
    }
    h[j + 1] = 0.25 * h[j];
  }
  *err = 1;                                            /* nrerror("Too many steps in routine qromb") */
  return 0.0;
}

/* CalcBlowingSnow.c:483-487 */
static void bs_get_shear(double x, double *f, double *df, double Ur, double Zr) {
  *f = log(2. * BS_G_STD * Zr / .12) + log(1 / (x * x)) - ORC_VON_K * Ur / x;
  *df = ORC_VON_K * Ur / (x * x) - 2. / x;
}

/* CalcBlowingSnow.c:424-481: Newton with bisection safeguard; *err where the reference exits ("Root must be bracketed") */
static double bs_rtnewt(double x1, double x2, double acc, double Ur, double Zr, int *err) {
  double df, dx, dxold, f, fh, fl, temp, xh, xl, rts;
  int j;
  bs_get_shear(x1, &fl, &df, Ur, Zr);
  bs_get_shear(x2, &fh, &df, Ur, Zr);
  if ((fl > 0.0 && fh > 0.0) || (fl < 0.0 && fh < 0.0)) { *err = 1; return 0.0; }
  if (fl == 0.0) return x1;
  if (fh == 0.0) return x2;
  if (fl < 0.0) { xl = x1; xh = x2; } else { xh = x1; xl = x2; }
  rts = 0.5 * (x1 + x2);
  dxold = fabs(x2 - x1);
  dx = dxold;
  bs_get_shear(rts, &f, &df, Ur, Zr);
  for (j = 1; j <= BS_MAX_ITER; j++) {
    if ((((rts - xh) * df - f) * ((rts - x1) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      rts = xl + dx;
      if (xl == rts) return rts;
    } else {
      dxold = dx;
      dx = f / df;
      temp = rts;
      rts -= dx;
      if (temp == rts) return rts;
    }
    if (fabs(dx) < acc) return rts;
    bs_get_shear(rts, &f, &df, Ur, Zr);
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return 0.0;                                          /* "Maximum number of iterations exceeded in rtnewt" */
}

/* CalcBlowingSnow.c:575-605 */
static double bs_get_prob(double Tair, double Age, double SurfaceLiquidWater, double U10) {
  double mean_u, sigma, prob;
  if (SurfaceLiquidWater < 0.001) {
    mean_u = 11.2 + 0.365 * Tair + 0.00706 * Tair * Tair + 0.9 * log(Age);
    sigma = 4.3 + 0.145 * Tair + 0.00196 * Tair * Tair;
    prob = 1. / (1. + exp(sqrt(BS_PI) * (mean_u - U10) / sigma));
  } else {
    mean_u = 21.;
    sigma = 7.;
    prob = 1. / (1. + exp(sqrt(BS_PI) * (mean_u - U10) / sigma));
  }
  if (prob < 0.0) prob = 0.0;
  if (prob > 1.0) prob = 1.0;
  return prob;
}

/* CalcBlowingSnow.c:607-633 (variable threshold, Li and Pomeroy 1997) */
static double bs_get_thresh(double Tair, double SurfaceLiquidWater, double Zo_salt) {
  double ut10 = (SurfaceLiquidWater < 0.001) ? 9.43 + .18 * Tair + .0033 * Tair * Tair : 9.9;
  return ORC_VON_K * ut10 / log(10. / Zo_salt);
}

/* CalcBlowingSnow.c:636-667 */
static int bs_shear_stress(double U10, double ZO, double *ushear, double *Zo_salt, double utshear) {
  double umin = utshear, umax = ORC_VON_K * U10, xacc = 0.10 * umin, fl, fh, df;
  int err = 0;
  bs_get_shear(umin, &fl, &df, U10, 10.);
  bs_get_shear(umax, &fh, &df, U10, 10.);
  if (fl < 0.0 && fh < 0.0) return -1;                 /* "Solution in rtnewt surpasses upper boundary": the reference exits */
  if (fl > 0.0 && fh > 0.0) {
    *Zo_salt = ZO;
    *ushear = ORC_VON_K * U10 / log(10. / ZO);
  } else {
    *ushear = bs_rtnewt(umin, umax, xacc, U10, 10., &err);
    *Zo_salt = 0.12 * (*ushear) * (*ushear) / (2. * BS_G_STD);
  }
  return err ? -1 : 0;
}

/* CalcBlowingSnow.c:669-753 (Liston & Sturm mass flux, fetch dependence on) */
static double bs_calc_sub_flux(double EactAir, double es, double Zrh, double AirDens, double utshear, double ushear, float fe,
                               double U10, double Zo_salt, double F, double *Transport, int *err) {
  double SubFlux = 0.0, particle = utshear * 2.8, Qsalt, hsalt, phi_s, T, ztop, saltation_transport, suspension_transport;
  bs_ctx c;
  Qsalt = (BS_CSALT * AirDens / BS_G_STD) * (utshear / ushear) * (ushear * ushear - utshear * utshear);
  Qsalt *= (1. + (500. / (3. * fe)) * (exp(-3. * fe / 500.) - 1.));
  hsalt = 0.08436 * pow(ushear, 1.27);
  phi_s = Qsalt / (hsalt * particle);
  T = 0.5 * (ushear * ushear) / (U10 * BS_SETTLING);
  ztop = hsalt * pow(T / (T + 1.), (ORC_VON_K * ushear) / (-1. * BS_SETTLING));
  c.es = es; c.Wind = U10; c.AirDens = AirDens; c.ZO = Zo_salt; c.EactAir = EactAir; c.F = F; c.hsalt = hsalt; c.phi_r = phi_s;
  c.ushear = ushear; c.Zrh = Zrh;
  if (EactAir >= es) SubFlux = 0.0;
  else {
    double psi_s = bs_sub_with_height(hsalt / 2., &c);
    SubFlux = phi_s * psi_s * hsalt;
    SubFlux += bs_qromb(bs_sub_with_height, &c, hsalt, ztop, err);
  }
  saltation_transport = Qsalt * (1 - exp(-3. * fe / 500.));
  suspension_transport = bs_qromb(bs_transport_with_height, &c, hsalt, ztop, err);
  *Transport = (suspension_transport + saltation_transport);
  *Transport /= fe;
  return SubFlux;
}

/* CalcBlowingSnow.c:101-310.  Returns the sublimation flux in kg m-2 s-1 (negative: loss), or ORC_ERROR where the reference
 * returns ERROR or stops the program. */
double orc_calc_blowing_snow(double Dt, double Tair, int LastSnow, double SurfaceLiquidWater, double Wind, double Ls, double AirDens,
                             double EactAir, double ZO, double Zrh, double snowdepth, float lag_one, float sigma_slope,
                             double Tsnow, int isArtificialBareSoil, float fe, double displacement, double roughness,
                             double *TotalTransport) {
  double Age, U10, Uo, prob, es, Ros, F, SubFlux, Diffusivity, ushear, Tk, utshear, upper, lower, Total, area, sigma_w, Zo_salt,
         ratio, wind10, Uveg, hv, Nd, Transport = 0.0;
  int p, err = 0;
  (void)Tsnow;
  Age = LastSnow * (Dt);
  es = orc_svp(Tair);
  Tk = Tair + ORC_KELVIN;
  Ros = 0.622 * es / (287 * Tk);
  Diffusivity = (2.06e-5) * pow(Tk / 273., 1.75);
  F = (Ls / (BS_KA * Tk)) * (Ls * BS_MW / (BS_R * Tk) - 1.);
  F += 1. / (Diffusivity * Ros);
  wind10 = Wind * log(10. / ZO) / log((2 + ZO) / ZO);
  if (isArtificialBareSoil) { fe = 1500; sigma_slope = .0002; }
  ratio = (2.44 - (0.43) * lag_one) * sigma_slope;
  sigma_w = wind10 * ratio;
  Uo = wind10;
  hv = (3. / 2.) * displacement;
  Nd = (4. / 3.) * (roughness / displacement);
  Total = 0.0;
  *TotalTransport = 0.0;
  area = 1. / BS_NUMINCS;
  if (snowdepth > 0.0) {
    if (sigma_w != 0.) {
      for (p = 0; p < BS_NUMINCS; p++) {
        SubFlux = lower = upper = 0.0;
        if (p == 0) { lower = -9999; upper = Uo + sigma_w * log(2. * (p + 1) * area); }
        else if (p > 0 && p < BS_NUMINCS / 2) { lower = Uo + sigma_w * log(2. * (p) * area); upper = Uo + sigma_w * log(2. * (p + 1) * area); }
        else if (p < (BS_NUMINCS - 1) && p >= BS_NUMINCS / 2) {
          lower = Uo - sigma_w * log(2. - 2. * (p * area));
          upper = Uo - sigma_w * log(2. - 2. * ((p + 1.) * area));
        } else if (p == BS_NUMINCS - 1) { lower = Uo - sigma_w * log(2. - 2. * (p * area)); upper = 9999; }
        if (lower > upper) lower = upper;
        U10 = Uo;
        if (lower >= Uo)
          U10 = -0.5 * ((upper + sigma_w) * exp((-1. / sigma_w) * (upper - Uo)) - (lower + sigma_w) * exp((-1. / sigma_w) * (lower - Uo))) / area;
        else if (upper <= Uo)
          U10 = 0.5 * ((upper - sigma_w) * exp((1. / sigma_w) * (upper - Uo)) - (lower - sigma_w) * exp((1. / sigma_w) * (lower - Uo))) / area;
        else return ORC_ERROR;
        if (U10 < 0.4) U10 = .4;
        if (U10 > 25.) U10 = 25.;
        if (snowdepth < hv) Uveg = U10 / sqrt(1. + 170 * Nd * (hv - snowdepth));
        else Uveg = U10;
        prob = bs_get_prob(Tair, Age, SurfaceLiquidWater, Uveg);
        utshear = bs_get_thresh(Tair, SurfaceLiquidWater, ZO);
        if (bs_shear_stress(U10, ZO, &ushear, &Zo_salt, utshear) != 0) return ORC_ERROR;
        if (ushear > utshear) {
          SubFlux = bs_calc_sub_flux(EactAir, es, Zrh, AirDens, utshear, ushear, fe, U10, Zo_salt, F, &Transport, &err);
          if (err) return ORC_ERROR;
        } else { SubFlux = 0.0; Transport = 0.0; }
        Total += (1. / BS_NUMINCS) * SubFlux * prob;
        *TotalTransport += (1. / BS_NUMINCS) * Transport * prob;
      }
    } else {
      U10 = Uo;
      if (snowdepth < hv) Uveg = U10 / sqrt(1. + 170 * Nd * (hv - snowdepth));
      else Uveg = U10;
      prob = bs_get_prob(Tair, Age, SurfaceLiquidWater, Uveg);
      utshear = bs_get_thresh(Tair, SurfaceLiquidWater, ZO);
      if (bs_shear_stress(Uo, ZO, &ushear, &Zo_salt, utshear) != 0) return ORC_ERROR;
      if (ushear > utshear) {
        SubFlux = bs_calc_sub_flux(EactAir, es, Zrh, AirDens, utshear, ushear, fe, Uo, Zo_salt, F, &Transport, &err);
        if (err) return ORC_ERROR;
      } else { SubFlux = 0.0; Transport = 0.0; }
      Total = SubFlux * prob;
      *TotalTransport = Transport * prob;
    }
  }
  if (Total < -.00005) Total = -.00005;
  return Total;
}
