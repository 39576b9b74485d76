/*
 * orc_surface.c — TEST INFRASTRUCTURE (CPU oracle): ground-surface energy balance,
 * soil temperature profile (Liang 1999 quick flux and Cherkauer 1999 explicit finite difference).
 */
#include "orc.h"

void orc_latent_heat_from_snow(double AirDens, double EactAir, double Lv, double Press, double Ra, double TMean, double Vpd,
                               double *LatentHeat, double *LatentHeatSub, double *VaporMassFlux, double *BlowingMassFlux,
                               double *SurfaceMassFlux);

/* estimate_T1.c:8-47 */
static double orc_estimate_T1(double Ts, double T1_old, double T2, double D1, double D2, double kappa1, double kappa2,
                              double Cs1, double Cs2, double dp, double delta_t) {
  double C1 = Cs2 * dp / D2 * (1. - exp(-D2 / dp));
  double C2 = -(1. - exp(D1 / dp)) * exp(-D2 / dp);
  double C3 = kappa1 / D1 - kappa2 / D1 + kappa2 / D1 * exp(-D1 / dp);
  (void)Cs1;
  return (kappa1 / 2. / D1 / D2 * (Ts) + C1 / delta_t * T1_old + (2. * C2 - 1. + exp(-D1 / dp)) * kappa2 / 2. / D1 / D2 * T2)
         / (C1 / delta_t + kappa2 / D1 / D2 * C2 + C3 / 2. / D2);
}

/* ---- soil_thermal_eqn.c:8-131 ---- */
typedef struct { double TL, TU, T0, moist, max_moist, bubble, expt, ice0, A, B, C, D, E; int EXP_TRANS, node; } orc_ste_ctx;

static double orc_soil_thermal_eqn(double T, void *vctx) {
  orc_ste_ctx *c = (orc_ste_ctx *)vctx;
  double value, ice, flux_term1, flux_term2;
  if (T < 0.) {
    ice = c->moist - orc_maximum_unfrozen_water(T, c->max_moist, c->bubble, c->expt);
    if (ice < 0.) ice = 0.;
    if (ice > c->max_moist) ice = c->max_moist;
  } else ice = 0.;
  if (!c->EXP_TRANS) {
    value = -c->A * (T - c->T0) + c->B * (c->TL - c->TU) + c->C * (c->TL - T) - c->D * (T - c->TU) + c->E * (ice - c->ice0);
    flux_term1 = c->B * (c->TL - c->TU);
    flux_term2 = c->C * (c->TL - T) - c->D * (T - c->TU);
    if (c->node == 1)
      if (fabs(c->TL - c->TU) > 5. && (T < c->TL && T < c->TU))
        if ((flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2))
          value = -c->A * (T - c->T0) + c->C * (c->TL - T) - c->D * (T - c->TU) + c->E * (ice - c->ice0);
  } else {
    value = -c->A * (T - c->T0) + c->B * (c->TL - c->TU) + c->C * (c->TL - 2. * T + c->TU) - c->D * (c->TL - c->TU)
            + c->E * (ice - c->ice0);
    flux_term1 = c->B * (c->TL - c->TU);
    flux_term2 = c->C * (c->TL - 2. * T + c->TU) - c->D * (c->TL - c->TU);
    if (c->node == 1)
      if (fabs(c->TL - c->TU) > 5. && (T < c->TL && T < c->TU))
        if ((flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2))
          value = -c->A * (T - c->T0) + c->C * (c->TL - 2. * T + c->TU) - c->D * (c->TL - c->TU) + c->E * (ice - c->ice0);
  }
  return value;
}

/* The per-node freezing parameters calc_soil_thermal_fluxes indexes (frozen_soil.c:397-399).
 * compat: the reference passes the 3-element LAYER arrays max_moist(mm)/bubble/expt, which sit directly in front of
 * the node arrays in soil_con_struct (vicNl_def.h:919-920,932-933,939-940), so index j reads layer[j] for j<3 and
 * node[j-3] otherwise (SURVEY.md Finding 1.2).  fixed: the node arrays. */
static void orc_node_freeze_params(const orc_model *m, const orc_soil *sc, int j, double *mm, double *bub, double *ex) {
  if (m->opt.frozen_compat) {
    if (j < 3) { *mm = sc->max_moist[j]; *bub = sc->bubble[j]; *ex = sc->expt[j]; }
    else { *mm = sc->max_moist_node[j - 3]; *bub = sc->bubble_node[j - 3]; *ex = sc->expt_node[j - 3]; }
  } else { *mm = sc->max_moist_node[j]; *bub = sc->bubble_node[j]; *ex = sc->expt_node[j]; }
}

/* ------------------------------------------------------------------ IMPLICIT soil heat solution
 * solve_T_profile_implicit (frozen_soil.c:229-301) with NewtonRaphsonMethod::{compute, fdjac3, fda_heat_eqn}
 * (newt_raph_func_fast.c:17-170, frozen_soil.c:540-803) and tridiag (newt_raph_func_fast.c:173-219).
 * Semantics of oracle patch P3: ice_new / Cs_new / kappa_new of fda_heat_eqn keep their values
 * from call to call, as the `static` arrays of upstream VIC did.  Every Newton trial starts with a full evaluation that
 * assigns kappa_new[0..n], ice_new[1..n], Cs_new[1..n], so arrays that live as long as one compute() are equivalent; the
 * element kappa_new[n+1] the bottom node's Dkappa reads (!NOFLUX) is never assigned by anyone and keeps the 0 a static
 * array starts with.  Node freezing parameters: the node arrays (P2; with frozen_compat the reference passes the 3-element
 * layer arrays, frozen_soil.c:283-284 -- not reproduced for IMPLICIT). */
typedef struct {
  int n, NOFLUX, EXP_TRANS;
  double deltat, Bexp, Ts, Tb;
  const double *T0, *moist, *ice, *kappa, *Cs;
  const orc_soil *sc;
  double ice_new[VIC_MAX_NODES + 2], Cs_new[VIC_MAX_NODES + 2], kappa_new[VIC_MAX_NODES + 2];
} orc_nr;

static void orc_fda_heat_eqn(orc_nr *q, const double *T_2, double *res, int focus) {
  const orc_soil *sc = q->sc;
  const int n = q->n;
  double DT[VIC_MAX_NODES], DT_down[VIC_MAX_NODES], DT_up[VIC_MAX_NODES], T_up[VIC_MAX_NODES], Dkappa[VIC_MAX_NODES];
  double storage_term, flux_term, phase_term, flux_term1, flux_term2, Lsum;
  int i, lidx, left, right, PAST_BOTTOM;
  if (focus == -1) { left = 0; right = n - 1; }
  else { left = (focus == 0) ? 0 : focus - 1; right = (focus == n - 1) ? n - 1 : focus + 1; }
  if (focus == -1) {
    lidx = 0; Lsum = 0.; PAST_BOTTOM = 0;
    for (i = 0; i < n + 1; i++) {
      q->kappa_new[i] = q->kappa[i];
      if (i >= 1) {
        if (T_2[i - 1] < 0) {
          q->ice_new[i] = q->moist[i] - orc_maximum_unfrozen_water(T_2[i - 1], sc->max_moist_node[i], sc->bubble_node[i], sc->expt_node[i]);
          if (q->ice_new[i] < 0) q->ice_new[i] = 0;
        } else q->ice_new[i] = 0;
        q->Cs_new[i] = q->Cs[i];
        if (q->ice_new[i] != q->ice[i]) {
          q->kappa_new[i] = orc_soil_conductivity(q->moist[i], q->moist[i] - q->ice_new[i], sc->soil_dens_min[lidx], sc->bulk_dens_min[lidx],
                                                  sc->quartz[lidx], sc->soil_density[lidx], sc->bulk_density[lidx], sc->organic[lidx]);
          q->Cs_new[i] = orc_volumetric_heat_capacity(sc->bulk_density[lidx] / sc->soil_density[lidx], q->moist[i] - q->ice_new[i],
                                                      q->ice_new[i], sc->organic[lidx]);
        }
      }
      if (sc->Zsum_node[i] > Lsum + sc->depth[lidx] && !PAST_BOTTOM) {
        Lsum += sc->depth[lidx]; lidx++;
        if (lidx == VIC_NLAYER) { PAST_BOTTOM = 1; lidx = VIC_NLAYER - 1; }
      }
    }
  } else {
    for (i = left; i <= right; i++) {
      if (T_2[i] < 0) {
        q->ice_new[i + 1] = q->moist[i + 1] - orc_maximum_unfrozen_water(T_2[i], sc->max_moist_node[i + 1], sc->bubble_node[i + 1], sc->expt_node[i + 1]);
        if (q->ice_new[i + 1] < 0) q->ice_new[i + 1] = 0;
      } else q->ice_new[i + 1] = 0;
    }
    lidx = 0; Lsum = 0.; PAST_BOTTOM = 0;
    for (i = 0; i <= right + 1; i++) {
      if (i >= left + 1 && q->ice_new[i] != q->ice[i]) {
        q->kappa_new[i] = orc_soil_conductivity(q->moist[i], q->moist[i] - q->ice_new[i], sc->soil_dens_min[lidx], sc->bulk_dens_min[lidx],
                                                sc->quartz[lidx], sc->soil_density[lidx], sc->bulk_density[lidx], sc->organic[lidx]);
        q->Cs_new[i] = orc_volumetric_heat_capacity(sc->bulk_density[lidx] / sc->soil_density[lidx], q->moist[i] - q->ice_new[i],
                                                    q->ice_new[i], sc->organic[lidx]);
      }
      if (sc->Zsum_node[i] > Lsum + sc->depth[lidx] && !PAST_BOTTOM) {
        Lsum += sc->depth[lidx]; lidx++;
        if (lidx == VIC_NLAYER) { PAST_BOTTOM = 1; lidx = VIC_NLAYER - 1; }
      }
    }
  }
  for (i = left; i <= right; i++) {
    if (i == 0) { DT[i] = T_2[i + 1] - q->Ts; DT_up[i] = T_2[i] - q->Ts; DT_down[i] = T_2[i + 1] - T_2[i]; T_up[i] = q->Ts; }
    else if (i == n - 1) { DT[i] = q->Tb - T_2[i - 1]; DT_up[i] = T_2[i] - T_2[i - 1]; DT_down[i] = q->Tb - T_2[i]; T_up[i] = T_2[i - 1]; }
    else { DT[i] = T_2[i + 1] - T_2[i - 1]; DT_up[i] = T_2[i] - T_2[i - 1]; DT_down[i] = T_2[i + 1] - T_2[i]; T_up[i] = T_2[i - 1]; }
    if (i < n - 1) Dkappa[i] = q->kappa_new[i + 2] - q->kappa_new[i];
    else if (!q->NOFLUX) Dkappa[i] = q->kappa_new[i + 2] - q->kappa_new[i];
    else Dkappa[i] = q->kappa_new[i + 1] - q->kappa_new[i];
  }
  for (i = left; i <= right; i++) {
    storage_term = q->Cs_new[i + 1] * (T_2[i] - q->T0[i + 1]) / q->deltat + T_2[i] * (q->Cs_new[i + 1] - q->Cs[i + 1]) / q->deltat;
    if (!q->EXP_TRANS) {
      flux_term1 = Dkappa[i] / sc->alpha[i] * DT[i] / sc->alpha[i];
      flux_term2 = q->kappa_new[i + 1] * (DT_down[i] / sc->gamma[i] - DT_up[i] / sc->beta[i]) / (0.5 * sc->alpha[i]);
    } else {
      const double z = sc->Zsum_node[i + 1] + 1.;
      flux_term1 = Dkappa[i] / 2. * DT[i] / 2. / (q->Bexp * z) / (q->Bexp * z);
      flux_term2 = q->kappa_new[i + 1] * ((DT_down[i] - DT_up[i]) / (q->Bexp * z) / (q->Bexp * z) - DT[i] / 2. / (q->Bexp * z * z));
    }
    /* "cold nose": every node in the full evaluation (the restriction is commented out there, frozen_soil.c:675),
     * the two near-surface nodes in the focus evaluation (:783) */
    if (focus == -1 || i == 0 || i == 1) {
      if (fabs(DT[i]) > 5. && (T_2[i] < T_2[i + 1] && T_2[i] < T_up[i])) {
        if ((flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2)) flux_term1 = 0;
      }
    }
    flux_term = flux_term1 + flux_term2;
    phase_term = ORC_ICE_DENSITY * ORC_LF * (q->ice_new[i + 1] - q->ice[i + 1]) / q->deltat;
    res[i] = flux_term + phase_term - storage_term;
  }
}

static void orc_tridiag(double *a, double *b, double *c, double *r, int n) {      /* newt_raph_func_fast.c:173-219 */
  int j;
  double factor;
  factor = b[0]; b[0] = 1.0; c[0] = c[0] / factor; r[0] = r[0] / factor;
  for (j = 1; j < n; j++) {
    factor = a[j]; a[j] = a[j] - b[j - 1] * factor; b[j] = b[j] - c[j - 1] * factor; r[j] = r[j] - r[j - 1] * factor;
    factor = b[j]; b[j] = 1.0; c[j] = c[j] / factor; r[j] = r[j] / factor;
  }
  for (j = n - 2; j >= 0; j--) {
    factor = c[j]; c[j] = c[j] - b[j + 1] * factor; r[j] = r[j] - r[j + 1] * factor;
    factor = b[j]; r[j] = r[j] / factor;
  }
}

/* returns 0, or 1 when the Newton iteration did not converge in 150 trials (the caller then solves explicitly) */
static int orc_solve_T_profile_implicit(double *T, const double *T0, const double *kappa, const double *Cs, const double *moist,
                                        double deltat, const double *ice, double Dp, int Nnodes, int NOFLUX, int EXP_TRANS,
                                        const orc_soil *sc) {
  enum { MAXTRIAL = 150 };
  const double TOLX = 1e-4, TOLF = 1e-1, R_MAX = 2.0, R_MIN = -5.0, RELAX1 = 0.9, RELAX2 = 0.7, RELAX3 = 0.2, EPS2 = 1e-4;
  const int n = NOFLUX ? Nnodes - 1 : Nnodes - 2;
  double *x = &T[1];
  double fvec[VIC_MAX_NODES], f[VIC_MAX_NODES], p[VIC_MAX_NODES], a[VIC_MAX_NODES], b[VIC_MAX_NODES], c[VIC_MAX_NODES];
  orc_nr q;
  int i, j, k;
  memset(&q, 0, sizeof(q));
  q.n = n; q.NOFLUX = NOFLUX; q.EXP_TRANS = EXP_TRANS; q.deltat = deltat; q.T0 = T0; q.moist = moist; q.ice = ice; q.kappa = kappa;
  q.Cs = Cs; q.sc = sc;
  if (EXP_TRANS) q.Bexp = NOFLUX ? log(Dp + 1.) / (double)n : log(Dp + 1.) / (double)(n + 1);
  q.Ts = T0[0];
  q.Tb = NOFLUX ? T0[n] : T0[n + 1];
  for (i = 0; i < n; i++) x[i] = T0[i + 1];
  for (k = 0; k < MAXTRIAL; k++) {
    double errf = 0.0, errx = 0.0;
    orc_fda_heat_eqn(&q, x, fvec, -1);
    for (i = 0; i < n; i++) errf += fabs(fvec[i]);
    if (errf <= TOLF) goto converged;
    for (j = 0; j < n; j++) {                                                       /* fdjac3, newt_raph_func_fast.c:136-167 */
      const double temp = x[j];
      double h = EPS2 * fabs(temp);
      if (h == 0) h = EPS2;
      x[j] = temp + h;
      h = x[j] - temp;
      orc_fda_heat_eqn(&q, x, f, j);
      x[j] = temp;
      b[j] = (f[j] - fvec[j]) / h;
      if (j != 0) c[j - 1] = (f[j - 1] - fvec[j - 1]) / h;
      if (j != n - 1) a[j + 1] = (f[j + 1] - fvec[j + 1]) / h;
    }
    for (i = 0; i < n; i++) p[i] = -fvec[i];
    orc_tridiag(a, b, c, p, n);
    for (i = 0; i < n; i++) {
      errx += fabs(p[i]);
      if (k > 10 && k <= 20 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX1;
      else if (k > 20 && k <= 60 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX2;
      else if (k > 60 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX3;
      else x[i] += p[i];
    }
    if (errx <= TOLX) goto converged;
  }
  return 1;
converged:
  T[0] = T0[0];
  if (!NOFLUX) T[Nnodes - 1] = T0[Nnodes - 1];
  return 0;
}

/* solve_T_profile (frozen_soil.c:105-225) + calc_soil_thermal_fluxes (:305-505); the A-E coefficients are recomputed
 * on every call, which is what the upstream `static` arrays give (SURVEY.md Finding 1.1). */
static int orc_solve_T_profile(const orc_model *m, double *T, const double *T0, int *Tfbflag, int *Tfbcount,
                               const double *kappa, const double *Cs, const double *moist, double deltat,
                               const double *ice, double Dp, int Nnodes, int NOFLUX, int EXP_TRANS, const orc_soil *sc) {
  const int MAXIT = 1000;
  const double threshold = 1.e-2;
  double A[VIC_MAX_NODES], B[VIC_MAX_NODES], C[VIC_MAX_NODES], D[VIC_MAX_NODES], E[VIC_MAX_NODES], Tlast[VIC_MAX_NODES];
  double Bexp = 0, maxdiff, diff, oldT;
  int j, Done = 0, ItCount = 0;
  const int frozen_on = (sc->FS_ACTIVE && m->opt.FROZEN_SOIL);
  if (EXP_TRANS) Bexp = log(Dp + 1.) / (double)(Nnodes - 1);
  if (!EXP_TRANS) {
    for (j = 1; j < Nnodes - 1; j++) {
      A[j] = Cs[j] * sc->alpha[j - 1] * sc->alpha[j - 1];
      B[j] = (kappa[j + 1] - kappa[j - 1]) * deltat;
      C[j] = 2 * deltat * kappa[j] * sc->alpha[j - 1] / sc->gamma[j - 1];
      D[j] = 2 * deltat * kappa[j] * sc->alpha[j - 1] / sc->beta[j - 1];
      E[j] = ORC_ICE_DENSITY * ORC_LF * sc->alpha[j - 1] * sc->alpha[j - 1];
    }
    if (NOFLUX) {
      j = Nnodes - 1;
      A[j] = Cs[j] * sc->alpha[j - 1] * sc->alpha[j - 1];
      B[j] = (kappa[j] - kappa[j - 1]) * deltat;
      C[j] = 2 * deltat * kappa[j] * sc->alpha[j - 1] / sc->gamma[j - 1];
      D[j] = 2 * deltat * kappa[j] * sc->alpha[j - 1] / sc->beta[j - 1];
      E[j] = ORC_ICE_DENSITY * ORC_LF * sc->alpha[j - 1] * sc->alpha[j - 1];
    }
  } else {
    for (j = 1; j < Nnodes - (NOFLUX ? 0 : 1); j++) {
      A[j] = 4 * Bexp * Bexp * Cs[j] * (sc->Zsum_node[j] + 1) * (sc->Zsum_node[j] + 1);
      B[j] = ((j < Nnodes - 1 ? kappa[j + 1] : kappa[j]) - kappa[j - 1]) * deltat;
      C[j] = 4 * deltat * kappa[j];
      D[j] = 2 * deltat * kappa[j] * Bexp;
      E[j] = 4 * Bexp * Bexp * ORC_ICE_DENSITY * ORC_LF * (sc->Zsum_node[j] + 1) * (sc->Zsum_node[j] + 1);
    }
  }
  for (j = 0; j < Nnodes; j++) T[j] = T0[j];
  for (j = 0; j < Nnodes; j++) Tlast[j] = T[j];
  for (j = 0; j < Nnodes; j++) { Tfbflag[j] = 0; Tfbcount[j] = 0; }

  while (!Done && ItCount < MAXIT) {
    ItCount++;
    maxdiff = threshold;
    for (j = 1; j < Nnodes - 1; j++) {
      oldT = T[j];
      if (T[j] >= 0 || !frozen_on) {
        if (!EXP_TRANS)
          T[j] = (A[j] * T0[j] + B[j] * (T[j + 1] - T[j - 1]) + C[j] * T[j + 1] + D[j] * T[j - 1] + E[j] * (0. - ice[j]))
                 / (A[j] + C[j] + D[j]);
        else
          T[j] = (A[j] * T0[j] + B[j] * (T[j + 1] - T[j - 1]) + C[j] * (T[j + 1] + T[j - 1]) - D[j] * (T[j + 1] - T[j - 1])
                  + E[j] * (0. - ice[j])) / (A[j] + 2. * C[j]);
      } else {
        orc_ste_ctx c;
        c.TL = T[j + 1]; c.TU = T[j - 1]; c.T0 = T0[j]; c.moist = moist[j];
        orc_node_freeze_params(m, sc, j, &c.max_moist, &c.bubble, &c.expt);
        c.ice0 = ice[j]; c.A = A[j]; c.B = B[j]; c.C = C[j]; c.D = D[j]; c.E = E[j]; c.EXP_TRANS = EXP_TRANS; c.node = j;
        /* The cold-nose variant of the residual (node 1, |TL - TU| > 5; soil_thermal_eqn.c:57-72, 84-93) drops the flux term
         * ft1 = B (TL - TU) on an interval Tb < T < Thi below both neighbours when ft1 < 0: the residual then is discontinuous
         * and may change sign more than once, in which case the root depends on the iteration's path and the test knob
         * (orc.h) does not apply.  It does apply when the sign change is unique: the tightened iteration on this same
         * residual then converges to it, and a sign change at or above Thi can only be the root of the smooth branch, above
         * which the residual is the smooth branch and below which it is positive throughout. */
        if (j == 1 && fabs(c.TL - c.TU) > 5.) {
          const double ref_T = orc_root_brent(T0[j] - (ORC_SOIL_DT), T0[j] + (ORC_SOIL_DT), orc_soil_thermal_eqn, &c);
          T[j] = ref_T;
          if (m->node_ttol < 1e-7) {
            const double Tt = orc_root_brent_tol(T0[j] - (ORC_SOIL_DT), T0[j] + (ORC_SOIL_DT), orc_soil_thermal_eqn, &c, m->node_macheps, m->node_ttol);
            const double ft1 = c.B * (c.TL - c.TU);
            int several = 0;
            if (ft1 < 0) {
              double Thi = (c.TL < c.TU) ? c.TL : c.TU, Tb;
              if (!EXP_TRANS) Tb = (c.C * c.TL + c.D * c.TU + ft1) / (c.C + c.D);
              else {
                const double num = c.C * (c.TL + c.TU) - c.D * (c.TL - c.TU), Ta = num / (2. * c.C);
                if (Ta < Thi) Thi = Ta;
                Tb = (num + ft1) / (2. * c.C);
              }
              several = (Tb < Thi) && (orc_is_error(Tt) || Tt < Thi + 1.e-6);
            }
            if (!several && !orc_is_error(Tt)) T[j] = Tt;
          }
        }
        else T[j] = orc_root_brent_tol(T0[j] - (ORC_SOIL_DT), T0[j] + (ORC_SOIL_DT), orc_soil_thermal_eqn, &c, m->node_macheps, m->node_ttol);
        if (orc_is_error(T[j])) {
          if (m->opt.TFALLBACK) { T[j] = T0[j]; Tfbflag[j] = 1; Tfbcount[j]++; }
          else return -1;
        }
      }
      diff = fabs(oldT - T[j]);
      if (diff > maxdiff) maxdiff = diff;
    }
    if (NOFLUX) {
      j = Nnodes - 1;
      oldT = T[j];
      if (T[j] >= 0 || !frozen_on) {
        if (!EXP_TRANS)
          T[j] = (A[j] * T0[j] + B[j] * (T[j] - T[j - 1]) + C[j] * T[j] + D[j] * T[j - 1] + E[j] * (0. - ice[j]))
                 / (A[j] + C[j] + D[j]);
        else
          T[j] = (A[j] * T0[j] + B[j] * (T[j] - T[j - 1]) + C[j] * (T[j] + T[j - 1]) - D[j] * (T[j] - T[j - 1])
                  + E[j] * (0. - ice[j])) / (A[j] + 2. * C[j]);
      } else {
        orc_ste_ctx c;
        c.TL = T[j]; c.TU = T[j - 1]; c.T0 = T0[j]; c.moist = moist[j];
        orc_node_freeze_params(m, sc, j, &c.max_moist, &c.bubble, &c.expt);
        c.ice0 = ice[j]; c.A = A[j]; c.B = B[j]; c.C = C[j]; c.D = D[j]; c.E = E[j]; c.EXP_TRANS = EXP_TRANS; c.node = j;
        T[j] = orc_root_brent_tol(T0[j] - ORC_SOIL_DT, T0[j] + ORC_SOIL_DT, orc_soil_thermal_eqn, &c, m->node_macheps, m->node_ttol);
        if (orc_is_error(T[j])) {
          if (m->opt.TFALLBACK) { T[j] = T0[j]; Tfbflag[j] = 1; Tfbcount[j]++; }
          else return -1;
        }
      }
      diff = fabs(oldT - T[Nnodes - 1]);
      if (diff > maxdiff) maxdiff = diff;
    }
    if (maxdiff <= threshold) Done = 1;
  }
  if (m->opt.TFALLBACK) {                                                           /* cold-nose hack :470-484 (sic: Tlast[j+1]-T[j]) */
    for (j = 1; j < Nnodes - 1; j++) {
      if (Tlast[j - 1] - Tlast[j] > 0 && Tlast[j + 1] - T[j] > 0 && (T[j - 1] - T[j]) - (Tlast[j - 1] - Tlast[j]) > 0
          && (T[j + 1] - T[j]) - (Tlast[j + 1] - Tlast[j]) > 0) {
        T[j] = 0.5 * (T[j - 1] + T[j + 1]);
        Tfbflag[j] = 1;
        Tfbcount[j]++;
      }
    }
  }
  if (!Done) {
    if (m->opt.TFALLBACK) {
      for (j = 0; j < Nnodes; j++) { T[j] = T0[j]; Tfbflag[j] = 1; Tfbcount[j]++; }
    } else return -1;
  }
  return 0;
}

/* ---- surf_energy_bal.h + func_surf_energy_bal.c:9-403 ---- */
typedef struct {
  const orc_model *m;
  const orc_soil *sc;
  int month, VEG, veg_idx;
  double delta_t, Cs1, Cs2, D1, D2, T1_old, T2, Ts_old, bubble, dp, expt, ice0, kappa1, kappa2, max_moist, moist;
  const double *root;
  int UnderStory, overstory;
  double NetShortBare, NetShortGrnd, NetShortSnow, Tair, atmos_density, atmos_pressure, emissivity, LongBareIn, LongSnowIn,
         surf_atten, vp, vpd;
  double Wdew;
  const orc_vc *displacement, *aero_resist, *ref_height, *roughness, *wind_speed;
  double *ra_used;
  double rainfall;
  double Le, Advection, OldTSurf, Tsnow_surf, kappa_snow, melt_energy, snow_coverage, snow_density, snow_swq, snow_water;
  double *deltaCC, *refreeze_energy, *vapor_flux, *blowing_flux, *surface_flux;
  int Nnodes;
  double *Cs_node, *T_node, *Tnew_node;
  int *Tnew_fbflag, *Tnew_fbcount;
  double *ice_node, *kappa_node, *moist_node;
  orc_layer *layer;
  orc_vegvar *vv;
  int INCLUDE_SNOW, NOFLUX, EXP_TRANS, SNOWING;
  double *NetLongBare, *NetLongSnow, *T1, *deltaH, *fusion, *grnd_flux, *latent_heat, *latent_heat_sub, *sensible_heat,
         *snow_flux, *store_error;
} orc_seb_ctx;

static double orc_surf_energy_bal(double Ts, void *vctx) {
  orc_seb_ctx *c = (orc_seb_ctx *)vctx;
  const orc_model *m = c->m;
  const orc_soil *sc = c->sc;
  const int U = c->UnderStory;
  double Evap, LongBareOut, NetBareRad, TMean, Tmp, error, ice;
  TMean = Ts;
  Tmp = TMean + ORC_KELVIN;
  if (c->snow_coverage > 0 && !c->INCLUDE_SNOW) *c->snow_flux = (c->kappa_snow * (c->Tsnow_surf - TMean));
  else if (c->INCLUDE_SNOW) { *c->snow_flux = 0; c->Tsnow_surf = TMean; }
  else *c->snow_flux = 0;

  if (m->opt.QUICK_FLUX) {
    *c->T1 = orc_estimate_T1(TMean, c->T1_old, c->T2, c->D1, c->D2, c->kappa1, c->kappa2, c->Cs1, c->Cs2, c->dp, c->delta_t);
    if (m->opt.GRND_FLUX_TYPE == VIC_GF_406)
      *c->grnd_flux = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten) * (c->kappa1 / c->D1 * ((*c->T1) - TMean));
    else
      *c->grnd_flux = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten)
                      * (c->kappa1 / c->D1 * ((*c->T1) - TMean)
                         + (c->kappa2 / c->D2 * (1. - exp(-c->D1 / c->dp)) * (c->T2 - (*c->T1)))) / 2.;
  } else {
    int err = 1;
    c->T_node[0] = TMean;
    if (m->opt.IMPLICIT)                                                            /* func_surf_energy_bal.c:192-210 */
    {
      err = orc_solve_T_profile_implicit(c->Tnew_node, c->T_node, c->kappa_node, c->Cs_node, c->moist_node, c->delta_t, c->ice_node,
                                         c->dp, c->Nnodes, c->NOFLUX, c->EXP_TRANS, sc);
      if (err == 0) ((orc_model *)m)->implicit_ok++; else ((orc_model *)m)->implicit_failed++;
    }
    if (!m->opt.IMPLICIT || err == 1) {                                             /* explicit, or the implicit solution failed (:212-222) */
      err = orc_solve_T_profile(m, c->Tnew_node, c->T_node, c->Tnew_fbflag, c->Tnew_fbcount, c->kappa_node, c->Cs_node,
                                c->moist_node, c->delta_t, c->ice_node, c->dp, c->Nnodes, c->NOFLUX, c->EXP_TRANS, sc);
      if (err) return ORC_ERROR;
    }
    *c->T1 = c->Tnew_node[1];
    if (m->opt.GRND_FLUX_TYPE == VIC_GF_406)
      *c->grnd_flux = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten) * (c->kappa1 / c->D1 * ((*c->T1) - TMean));
    else
      *c->grnd_flux = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten)
                      * (c->kappa1 / c->D1 * ((*c->T1) - TMean) + (c->kappa2 / c->D2 * (c->Tnew_node[2] - (*c->T1)))) / 2.;
  }
  if (m->opt.GRND_FLUX_TYPE == VIC_GF_FULL)
    *c->deltaH = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten)
                 * (c->Cs1 * ((c->Ts_old + c->T1_old) - (TMean + *c->T1)) * c->D1 / c->delta_t / 2.);
  else
    *c->deltaH = (c->Cs1 * ((c->Ts_old + c->T1_old) - (TMean + *c->T1)) * c->D1 / c->delta_t / 2.);
  if (sc->FS_ACTIVE && m->opt.FROZEN_SOIL) {
    if ((TMean + *c->T1) / 2. < 0.) {
      ice = c->moist - orc_maximum_unfrozen_water((TMean + *c->T1) / 2., c->max_moist, c->bubble, c->expt);
      if (ice < 0.) ice = 0.;
    } else ice = 0.;
    if (m->opt.GRND_FLUX_TYPE == VIC_GF_FULL)
      *c->fusion = (c->snow_coverage + (1. - c->snow_coverage) * c->surf_atten)
                   * (-ORC_ICE_DENSITY * ORC_LF * (c->ice0 - ice) * c->D1 / c->delta_t);
    else
      *c->fusion = (-ORC_ICE_DENSITY * ORC_LF * (c->ice0 - ice) * c->D1 / c->delta_t);
  }
  if (c->INCLUDE_SNOW) {
    if (TMean > 0) *c->deltaCC = ORC_CH_ICE * (c->snow_swq - c->snow_water) * (0 - c->OldTSurf) / c->delta_t;
    else *c->deltaCC = ORC_CH_ICE * (c->snow_swq - c->snow_water) * (TMean - c->OldTSurf) / c->delta_t;
    *c->refreeze_energy = (c->snow_water * ORC_LF * c->snow_density) / c->delta_t;
    *c->deltaCC *= c->snow_coverage;
    *c->refreeze_energy *= c->snow_coverage;
  }
  LongBareOut = ORC_STEFAN_B * Tmp * Tmp * Tmp * Tmp;
  if (c->INCLUDE_SNOW) *c->NetLongSnow = (c->LongSnowIn - c->snow_coverage * LongBareOut);
  *c->NetLongBare = (c->LongBareIn - (1. - c->snow_coverage) * LongBareOut);
  NetBareRad = (c->NetShortBare + (*c->NetLongBare) + *c->grnd_flux + *c->deltaH + *c->fusion);

  if (c->wind_speed->v[U] > 0.0 && c->overstory && c->SNOWING)
    c->ra_used[0] = c->aero_resist->v[U]
                    / orc_stability_correction(c->ref_height->v[U], 0.f, TMean, c->Tair, c->wind_speed->v[U], c->roughness->v[U]);
  else if (c->wind_speed->v[U] > 0.0)
    c->ra_used[0] = c->aero_resist->v[U]
                    / orc_stability_correction(c->ref_height->v[U], c->displacement->v[U], TMean, c->Tair,
                                               c->wind_speed->v[U], c->roughness->v[U]);
  else c->ra_used[0] = ORC_HUGE_RESIST;

  if (c->VEG && !c->SNOWING && orc_veg(m, c->veg_idx)[VL_LAI + c->month - 1] > 0) {
    double wdew = c->Wdew;
    Evap = orc_canopy_evap(m, c->layer, c->vv, 1, c->veg_idx, c->month, &wdew, c->delta_t, NetBareRad, c->vpd,
                           c->NetShortBare, c->Tair, c->ra_used[1], (double)sc->elevation, c->rainfall, sc, c->root);
  } else if (!c->SNOWING) {
    Evap = orc_arno_evap(c->layer, NetBareRad, c->Tair, c->vpd, sc->depth[0], c->max_moist * sc->depth[0] * 1000.,
                         (double)sc->elevation, sc->b_infilt, c->ra_used[0], c->delta_t, sc->resid_moist[0]);
  } else Evap = 0.;

  *c->latent_heat = -ORC_RHO_W * c->Le * Evap;
  *c->latent_heat_sub = 0.;
  if (c->INCLUDE_SNOW) {
    double VaporMassFlux = *c->vapor_flux * ORC_ICE_DENSITY / c->delta_t;
    double BlowingMassFlux = *c->blowing_flux * ORC_ICE_DENSITY / c->delta_t;
    double SurfaceMassFlux = *c->surface_flux * ORC_ICE_DENSITY / c->delta_t;
    double tl, tls;
    orc_latent_heat_from_snow(c->atmos_density, c->vp, c->Le, c->atmos_pressure, c->ra_used[0], TMean, c->vpd, &tl, &tls,
                              &VaporMassFlux, &BlowingMassFlux, &SurfaceMassFlux);
    *c->latent_heat += tl * c->snow_coverage;
    *c->latent_heat_sub = tls * c->snow_coverage;
    *c->vapor_flux = VaporMassFlux * c->delta_t / ORC_ICE_DENSITY;
    *c->blowing_flux = BlowingMassFlux * c->delta_t / ORC_ICE_DENSITY;
    *c->surface_flux = SurfaceMassFlux * c->delta_t / ORC_ICE_DENSITY;
  } else *c->latent_heat *= (1. - c->snow_coverage);

  if (c->snow_coverage < 1 || c->INCLUDE_SNOW) {
    *c->sensible_heat = c->atmos_density * ORC_CP * (c->Tair - (TMean)) / c->ra_used[0];
    if (!c->INCLUDE_SNOW) (*c->sensible_heat) *= (1. - c->snow_coverage);
  } else *c->sensible_heat = 0.;

  error = (NetBareRad + c->NetShortGrnd + c->NetShortSnow + c->emissivity * (*c->NetLongSnow)) + *c->sensible_heat
          + (*c->latent_heat + *c->latent_heat_sub) + *c->snow_flux * c->snow_coverage + c->melt_energy + c->Advection
          - *c->deltaCC;
  if (c->INCLUDE_SNOW) {
    if (c->Tsnow_surf == 0.0 && error > -(*c->refreeze_energy)) {
      *c->refreeze_energy = -error;
      error = 0.0;
    } else error += *c->refreeze_energy;
  }
  *c->store_error = error;
  return error;
}

/* calc_surf_energy_bal.c:7-692.  Returns Tsurf or ORC_ERROR. */
double orc_calc_surf_energy_bal(const orc_model *m, double Le, double LongUnderIn, double NetLongSnow, double NetShortGrnd,
                                double NetShortSnow, double OldTSurf, double ShortUnderIn, double SnowAlbedo,
                                double SnowLatent, double SnowLatentSub, double SnowSensible, double Tair, double VPDcanopy,
                                double VPcanopy, double delta_coverage, double dp, double ice0, double melt_energy,
                                double moist, double snow_coverage, double snow_depth, double BareAlbedo, double surf_atten,
                                orc_vc *aero_resist, double *ra_used, orc_vc *displacement, double *melt, double *ppt,
                                double *rainfall, orc_vc *ref_height, orc_vc *roughness, orc_vc *wind_speed,
                                const double *root, int INCLUDE_SNOW, int UnderStory, int Nnodes, int dt, int hidx,
                                int overstory, int veg_idx, int is_artificial_bare, const orc_atmos *atmos,
                                const orc_dmy *dmy, orc_energy *energy, orc_layer *layer, orc_snow *snow,
                                const orc_soil *sc, orc_vegvar *vv) {
  const double *vl = orc_veg(m, veg_idx);
  int VEG, nidx, Tsurf_fbflag = 0, Tsurf_fbcount = 0;
  double Tnew_node[VIC_MAX_NODES];
  int Tnew_fbflag[VIC_MAX_NODES], Tnew_fbcount[VIC_MAX_NODES];
  double NetLongBare, NetShortBare, LongBareIn, T1 = 0, Tsurf, error, kappa_snow, TmpNetLongSnow, TmpNetShortSnow, LongSnowIn;
  double T_lower, T_upper, Ts_old, delta_t, refrozen_water;
  orc_seb_ctx c;

  for (nidx = 0; nidx < Nnodes; nidx++) { Tnew_fbflag[nidx] = 0; Tnew_fbcount[nidx] = 0; Tnew_node[nidx] = 0; }
  if (!is_artificial_bare) VEG = (vl[VL_LAI + dmy->month - 1] > 0.0) ? 1 : 0;
  else VEG = 0;
  Ts_old = energy->T[0];
  delta_t = (double)dt * 3600.;
  if (snow->depth > 0.) kappa_snow = ORC_K_SNOW * (snow->density) * (snow->density) / snow_depth;
  else kappa_snow = 0;
  NetShortBare = (ShortUnderIn * (1. - (snow_coverage + delta_coverage)) * (1. - BareAlbedo)
                  + ShortUnderIn * (delta_coverage) * (1. - SnowAlbedo));
  LongBareIn = (1. - snow_coverage) * LongUnderIn;
  if (INCLUDE_SNOW || snow->swq == 0) {
    TmpNetLongSnow = NetLongSnow;
    TmpNetShortSnow = NetShortSnow;
    LongSnowIn = snow_coverage * LongUnderIn;
  } else {
    TmpNetShortSnow = 0.;
    TmpNetLongSnow = 0.;
    LongSnowIn = 0.;
  }

  memset(&c, 0, sizeof(c));
  c.m = m; c.sc = sc; c.month = dmy->month; c.VEG = VEG; c.veg_idx = veg_idx; c.delta_t = delta_t;
  c.Cs1 = energy->Cs[0]; c.Cs2 = energy->Cs[1];
  c.D1 = sc->Zsum_node[1] - sc->Zsum_node[0]; c.D2 = sc->Zsum_node[2] - sc->Zsum_node[1];
  c.T1_old = energy->T[1]; c.T2 = energy->T[Nnodes - 1]; c.Ts_old = Ts_old;
  c.bubble = sc->bubble[0]; c.dp = dp; c.expt = sc->expt[0]; c.ice0 = ice0;
  c.kappa1 = energy->kappa[0]; c.kappa2 = energy->kappa[1];
  c.max_moist = sc->max_moist[0] / (sc->depth[0] * 1000.); c.moist = moist; c.root = root;
  c.UnderStory = UnderStory; c.overstory = overstory; c.NetShortBare = NetShortBare; c.NetShortGrnd = NetShortGrnd;
  c.NetShortSnow = TmpNetShortSnow; c.Tair = Tair; c.atmos_density = atmos->density[hidx];
  c.atmos_pressure = atmos->pressure[hidx]; c.emissivity = 1.; c.LongBareIn = LongBareIn; c.LongSnowIn = LongSnowIn;
  c.surf_atten = surf_atten; c.vp = VPcanopy; c.vpd = VPDcanopy; c.Wdew = vv->Wdew;
  c.displacement = displacement; c.aero_resist = aero_resist; c.ra_used = ra_used; c.rainfall = *rainfall;
  c.ref_height = ref_height; c.roughness = roughness; c.wind_speed = wind_speed; c.Le = Le;
  c.Advection = energy->advection; c.OldTSurf = OldTSurf; c.Tsnow_surf = snow->surf_temp; c.kappa_snow = kappa_snow;
  c.melt_energy = melt_energy; c.snow_coverage = snow_coverage; c.snow_density = snow->density; c.snow_swq = snow->swq;
  c.snow_water = snow->surf_water; c.deltaCC = &energy->deltaCC; c.refreeze_energy = &energy->refreeze_energy;
  c.vapor_flux = &snow->vapor_flux; c.blowing_flux = &snow->blowing_flux; c.surface_flux = &snow->surface_flux;
  c.Nnodes = Nnodes; c.Cs_node = energy->Cs_node; c.T_node = energy->T; c.Tnew_node = Tnew_node;
  c.Tnew_fbflag = Tnew_fbflag; c.Tnew_fbcount = Tnew_fbcount; c.ice_node = energy->ice; c.kappa_node = energy->kappa_node;
  c.moist_node = energy->moist; c.layer = layer; c.vv = vv; c.INCLUDE_SNOW = INCLUDE_SNOW; c.NOFLUX = m->opt.NOFLUX;
  c.EXP_TRANS = m->opt.EXP_TRANS; c.SNOWING = snow->snow; c.NetLongBare = &NetLongBare; c.NetLongSnow = &TmpNetLongSnow;
  c.T1 = &T1; c.deltaH = &energy->deltaH; c.fusion = &energy->fusion; c.grnd_flux = &energy->grnd_flux;
  c.latent_heat = &energy->latent; c.latent_heat_sub = &energy->latent_sub; c.sensible_heat = &energy->sensible;
  c.snow_flux = &energy->snow_flux; c.store_error = &energy->error;

  if (m->opt.FULL_ENERGY) {
    if (INCLUDE_SNOW) { T_lower = energy->T[0] - ORC_SURF_DT; T_upper = 0.; }
    else { T_lower = 0.5 * (energy->T[0] + Tair) - ORC_SURF_DT; T_upper = 0.5 * (energy->T[0] + Tair) + ORC_SURF_DT; }
    if (m->opt.QUICK_SOLVE && !m->opt.QUICK_FLUX) {
      /* calc_surf_energy_bal.c:289-309: iterate on the nodes down to the thaw depth + 4 only, with NOFLUX and EXP_TRANS forced
       * FALSE -- the local copies the final evaluation sees too: NOFLUX comes back only with a second iteration (:403),
       * EXP_TRANS never does on this branch (the linear-spacing coefficients then run on whatever node geometry the run has) */
      int tmpNnodes = 0;
      c.NOFLUX = 0;
      c.EXP_TRANS = 0;
      for (nidx = Nnodes - 5; nidx >= 0; nidx--)
        if (energy->T[nidx] >= 0 && energy->T[nidx + 1] < 0) tmpNnodes = nidx + 1;
      if (tmpNnodes == 0) {
        if (energy->T[0] <= 0 && energy->T[1] >= 0) tmpNnodes = Nnodes;
        else tmpNnodes = 3;
      } else tmpNnodes += 4;
      c.Nnodes = tmpNnodes;
    }
    Tsurf = orc_root_brent(T_lower, T_upper, orc_surf_energy_bal, &c);
    if (orc_is_error(Tsurf)) {
      if (m->opt.TFALLBACK) { Tsurf = Ts_old; Tsurf_fbflag = 1; Tsurf_fbcount++; }
      else return ORC_ERROR;
    }
    if (Ts_old * Tsurf < 0 && m->opt.QUICK_SOLVE) {                                   /* :400-480: again on the whole column */
      c.Nnodes = Nnodes;
      c.NOFLUX = m->opt.NOFLUX;                                                       /* :403 */
      c.Tsnow_surf = snow->surf_temp;              /* a fresh SurfEnergyBal object: by-value members restart */
      Tsurf = orc_root_brent(T_lower, T_upper, orc_surf_energy_bal, &c);
      if (orc_is_error(Tsurf)) {
        if (m->opt.TFALLBACK) { Tsurf = Ts_old; Tsurf_fbflag = 1; Tsurf_fbcount++; }
        else return ORC_ERROR;
      }
    }
    c.Nnodes = Nnodes;
  } else Tsurf = Tair;

  /* the final evaluation uses a fresh SurfEnergyBal object (calc_surf_energy_bal.c:489-506): by-value members such as
     Tsnow_surf restart from the caller's values */
  c.Tsnow_surf = snow->surf_temp;
  error = orc_surf_energy_bal(Tsurf, &c);
  if (error == ORC_ERROR) return ORC_ERROR;
  energy->error = error;

  if (m->opt.QUICK_FLUX || !(m->opt.FULL_ENERGY || (m->opt.FROZEN_SOIL && sc->FS_ACTIVE))) {
    Tnew_node[0] = Tsurf;
    Tnew_node[1] = T1;
    Tnew_node[2] = c.T2;
  }
  /* calc_layer_average_thermal_props, frozen_soil.c:12-103 */
  if (m->opt.FROZEN_SOIL && sc->FS_ACTIVE) orc_find_0_degree_fronts(energy, sc->Zsum_node, Tnew_node, Nnodes);
  else energy->Nfrost = 0;
  for (nidx = 0; nidx < Nnodes; nidx++) energy->T[nidx] = Tnew_node[nidx];
  energy->frozen = (energy->Nfrost > 0) ? 1 : 0;
  if (m->opt.QUICK_FLUX) orc_estimate_layer_ice_content_quick_flux(m, layer, energy->T[0], energy->T[1], sc);
  else if (orc_estimate_layer_ice_content(m, layer, energy->T, sc) != 0) return ORC_ERROR;

  if (!snow->snow && !INCLUDE_SNOW) {                                               /* :527-546 */
    if (!is_artificial_bare) {
      if (vl[VL_LAI + dmy->month - 1] <= 0.0) {
        vv->throughfall = *rainfall;
        *ppt = vv->throughfall;
      } else *ppt = vv->throughfall;
    } else *ppt = *rainfall;
  }
  energy->NetShortGrnd = NetShortGrnd;
  if (INCLUDE_SNOW) {
    energy->NetLongUnder = NetLongBare + TmpNetLongSnow;
    energy->NetShortUnder = NetShortBare + TmpNetShortSnow + NetShortGrnd;
  } else {
    energy->NetLongUnder = NetLongBare + NetLongSnow;
    energy->NetShortUnder = NetShortBare + NetShortSnow + NetShortGrnd;
    energy->latent = (SnowLatent + energy->latent);
    energy->latent_sub = (SnowLatentSub + energy->latent_sub);
    energy->sensible = (SnowSensible + energy->sensible);
  }
  energy->LongUnderOut = LongUnderIn - energy->NetLongUnder;
  energy->AlbedoUnder = ((1. - (snow_coverage + delta_coverage)) * BareAlbedo + (snow_coverage + delta_coverage) * SnowAlbedo);
  energy->melt_energy = melt_energy;
  energy->Tsurf = (snow->coverage * snow->surf_temp + (1. - snow->coverage) * Tsurf);

  if (INCLUDE_SNOW) {                                                               /* :589-679 */
    if (-(snow->vapor_flux) > snow->swq) {
      snow->blowing_flux *= -(snow->swq / snow->vapor_flux);
      snow->vapor_flux = -(snow->swq);
      snow->surface_flux = snow->vapor_flux - snow->blowing_flux;
    }
    snow->swq += snow->vapor_flux;
    snow->surf_water += snow->vapor_flux;
    snow->surf_water = (snow->surf_water < 0) ? 0. : snow->surf_water;
    if (energy->refreeze_energy >= 0.0) {
      refrozen_water = energy->refreeze_energy / (ORC_LF * ORC_RHO_W) * delta_t;
      if (refrozen_water > snow->surf_water) {
        refrozen_water = snow->surf_water;
        energy->refreeze_energy = refrozen_water * ORC_LF * ORC_RHO_W / delta_t;
      }
      snow->surf_water -= refrozen_water;
      if (snow->surf_water < 0.0) snow->surf_water = 0.0;
      *melt = 0.0;
    } else {
      *melt = fabs(energy->refreeze_energy) / (ORC_LF * ORC_RHO_W) * delta_t;
      snow->swq -= *melt;
      if (snow->swq < 0) { *melt += snow->swq; snow->swq = 0; }
    }
    if (snow->swq > 0) {
      snow->surf_temp = (Tsurf > 0) ? 0 : Tsurf;
      snow->coldcontent = ORC_CH_ICE * snow->surf_temp * snow->swq;
      snow->depth = 1000. * snow->swq / snow->density;
      if (snow->swq > 0) snow->coverage = 1.; else snow->coverage = 0.;
      if (isnan(snow->surf_temp) || snow->surf_temp > 0)
        energy->snow_flux = (energy->grnd_flux + energy->deltaH + energy->fusion);
    } else {
      snow->density = 0.;
      snow->depth = 0.;
      snow->surf_water = 0;
      snow->pack_water = 0;
      snow->surf_temp = 0;
      snow->pack_temp = 0;
      snow->coverage = 0;
    }
    snow->vapor_flux *= -1;
  }
  energy->Tsurf_fbflag = Tsurf_fbflag;
  energy->Tsurf_fbcount += Tsurf_fbcount;
  for (nidx = 0; nidx < Nnodes; nidx++) {
    energy->T_fbflag[nidx] = Tnew_fbflag[nidx];
    energy->T_fbcount[nidx] += Tnew_fbcount[nidx];
  }
  return Tsurf;
}

/* exported wrapper for the pure-function hook (vicorc_pure) */
double orc_estimate_T1_x(double Ts, double T1_old, double T2, double D1, double D2, double kappa1, double kappa2, double Cs1, double Cs2,
                         double dp, double delta_t) {
  return orc_estimate_T1(Ts, T1_old, T2, D1, D2, kappa1, kappa2, Cs1, Cs2, dp, delta_t);
}
