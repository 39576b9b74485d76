// vic_implicit.hpp — the IMPLICIT soil heat solution on the device (device only, gfx950):
// solve_T_profile_implicit (frozen_soil.c:229-301), NewtonRaphsonMethod::{compute, fdjac3, fda_heat_eqn}
// (newt_raph_func_fast.c:17-170, frozen_soil.c:540-803), tridiag (newt_raph_func_fast.c:173-219).
//
// One lane = one profile solve at a time, taken from the round's work list.  The Newton iteration is short where it converges
// (TOLF is 0.1 W/m3: two to five trials) and long where it does not: 150 trials, after which the HRU goes on the round's
// fall-back list and the explicit kernel (vic_profile.hpp) solves it right after, exactly like func_surf_energy_bal.c:212-222.
// A wave of 64 solves that ran to its slowest lane did 150 trials almost every time (one failing solve in a hundred is enough:
// 2.3 s per step on the cfg3 domain).  So the waves are persistent and the unit of lock-step is ONE TRIAL: a lane whose solve
// has ended -- either way -- writes it out and takes the next one from the list while its neighbours carry on.
// Semantics of upstream VIC's `static` work arrays (oracle patch P3): every trial's full
// evaluation assigns kappa_new[0..n], ice_new[1..n], Cs_new[1..n]; the focus evaluations of the finite-difference
// Jacobian then update them in place, column after column; kappa_new[n+1], which the bottom node's Dkappa reads, is never
// assigned and stays 0.  Node freezing parameters are the node arrays (frozen_compat = 0 only).
// Fallback flags: the explicit solver zeroes Tnew_fbflag / Tnew_fbcount when it runs, the implicit one does not touch them
// (calc_surf_energy_bal.c:222, 685), so a converged implicit solve carries the flags of the root find's most recent
// explicit solve: lastexp[hru] is the record slot that holds them (-1: none yet).
#pragma once
#include "vic_profile.hpp"

namespace vic {

enum { PI_MOIST = 0, PI_ICE, PI_KAPPA, PI_CS, PIMP };     // rows of the implicit solver's item block [nhru][Nn][PIMP]

struct IArgs {
  int ncell, nhru, Nband;
  const double* pimp;
  const int* hpi;
  const double* cell_params;
  const int* hkey;
  int* fb_list;          // HRUs whose iteration failed (NBUCKET segments of cap entries) ...
  int* fb_count;         // ... and their counts [NBUCKET], zero at launch
  int* lastexp;          // [nhru]
  int* cursor;           // next unclaimed entry of the work list, zero at launch
};

struct ImplicitSolver {
  static constexpr int M = VIC_MAX_NODES + 2;
  int n, NOFLUX, EXP_TRANS;
  double deltat, Bexp, Ts, Tb;
  double T0[M], moist[M], ice[M], kappa[M], Cs[M];
  double ice_new[M], Cs_new[M], kappa_new[M];
  // cell constants
  double mmn[M], bub[M], ex[M], al[M], be[M], ga[M], zs[M];
  double sdm[3], bdm[3], qz[3], sden[3], bden[3], org[3], depth[3];

  VIC_DEV void props(int i, int lidx) {
    kappa_new[i] = soil_conductivity(moist[i], moist[i] - ice_new[i], sdm[lidx], bdm[lidx], qz[lidx], sden[lidx], bden[lidx], org[lidx]);
    Cs_new[i] = volumetric_heat_capacity(bden[lidx] / sden[lidx], moist[i] - ice_new[i], ice_new[i], org[lidx]);
  }

  VIC_DEV void fda_heat_eqn(const double* T_2, double* res, int focus) {
    double DT[M], DT_down[M], DT_up[M], T_up[M], Dkappa[M];
    int left, right;
    if (focus == -1) { left = 0; right = n - 1; }
    else { left = (focus == 0) ? 0 : focus - 1; right = (focus == n - 1) ? n - 1 : focus + 1; }
    int lidx = 0;
    double Lsum = 0.;
    bool PAST_BOTTOM = false;
    if (focus == -1) {
      for (int i = 0; i < n + 1; i++) {
        kappa_new[i] = kappa[i];
        if (i >= 1) {
          if (T_2[i - 1] < 0) {
            ice_new[i] = moist[i] - maximum_unfrozen_water(T_2[i - 1], mmn[i], bub[i], ex[i]);
            if (ice_new[i] < 0) ice_new[i] = 0;
          } else ice_new[i] = 0;
          Cs_new[i] = Cs[i];
          if (ice_new[i] != ice[i]) props(i, lidx);
        }
        if (zs[i] > Lsum + depth[lidx] && !PAST_BOTTOM) {
          Lsum += depth[lidx]; lidx++;
          if (lidx == VIC_NLAYER) { PAST_BOTTOM = true; lidx = VIC_NLAYER - 1; }
        }
      }
    } else {
      for (int i = left; i <= right; i++) {
        if (T_2[i] < 0) {
          ice_new[i + 1] = moist[i + 1] - maximum_unfrozen_water(T_2[i], mmn[i + 1], bub[i + 1], ex[i + 1]);
          if (ice_new[i + 1] < 0) ice_new[i + 1] = 0;
        } else ice_new[i + 1] = 0;
      }
      for (int i = 0; i <= right + 1; i++) {
        if (i >= left + 1 && ice_new[i] != ice[i]) props(i, lidx);
        if (zs[i] > Lsum + depth[lidx] && !PAST_BOTTOM) {
          Lsum += depth[lidx]; lidx++;
          if (lidx == VIC_NLAYER) { PAST_BOTTOM = true; lidx = VIC_NLAYER - 1; }
        }
      }
    }
    for (int i = left; i <= right; i++) {
      if (i == 0) { DT[i] = T_2[i + 1] - Ts; DT_up[i] = T_2[i] - Ts; DT_down[i] = T_2[i + 1] - T_2[i]; T_up[i] = Ts; }
      else if (i == n - 1) { DT[i] = Tb - T_2[i - 1]; DT_up[i] = T_2[i] - T_2[i - 1]; DT_down[i] = Tb - T_2[i]; T_up[i] = T_2[i - 1]; }
      else { DT[i] = T_2[i + 1] - T_2[i - 1]; DT_up[i] = T_2[i] - T_2[i - 1]; DT_down[i] = T_2[i + 1] - T_2[i]; T_up[i] = T_2[i - 1]; }
      if (i < n - 1) Dkappa[i] = kappa_new[i + 2] - kappa_new[i];
      else if (!NOFLUX) Dkappa[i] = kappa_new[i + 2] - kappa_new[i];
      else Dkappa[i] = kappa_new[i + 1] - kappa_new[i];
    }
    for (int i = left; i <= right; i++) {
      const double storage_term = Cs_new[i + 1] * (T_2[i] - T0[i + 1]) / deltat + T_2[i] * (Cs_new[i + 1] - Cs[i + 1]) / deltat;
      double flux_term1, flux_term2;
      if (!EXP_TRANS) {
        flux_term1 = Dkappa[i] / al[i] * DT[i] / al[i];
        flux_term2 = kappa_new[i + 1] * (DT_down[i] / ga[i] - DT_up[i] / be[i]) / (0.5 * al[i]);
      } else {
        const double z = zs[i + 1] + 1.;
        flux_term1 = Dkappa[i] / 2. * DT[i] / 2. / (Bexp * z) / (Bexp * z);
        flux_term2 = kappa_new[i + 1] * ((DT_down[i] - DT_up[i]) / (Bexp * z) / (Bexp * z) - DT[i] / 2. / (Bexp * z * z));
      }
      // "cold nose": every node in the full evaluation (frozen_soil.c:675 has the restriction commented out), the two
      // near-surface nodes in the focus evaluation (:783)
      if (focus == -1 || i == 0 || i == 1) {
        if (fabs(DT[i]) > 5. && (T_2[i] < T_2[i + 1] && T_2[i] < T_up[i])) {
          if ((flux_term1 < 0 && flux_term2 > 0) && fabs(flux_term1) > fabs(flux_term2)) flux_term1 = 0;
        }
      }
      const double flux_term = flux_term1 + flux_term2;
      const double phase_term = ICE_DENSITY * LF * (ice_new[i + 1] - ice[i + 1]) / deltat;
      res[i] = flux_term + phase_term - storage_term;
    }
  }

  VIC_DEV static void tridiag(double* a, double* b, double* c, double* r, int n) {
    double factor = b[0];
    b[0] = 1.0; c[0] = c[0] / factor; r[0] = r[0] / factor;
    for (int j = 1; j < n; j++) {
      factor = a[j]; a[j] = a[j] - b[j - 1] * factor; b[j] = b[j] - c[j - 1] * factor; r[j] = r[j] - r[j - 1] * factor;
      factor = b[j]; b[j] = 1.0; c[j] = c[j] / factor; r[j] = r[j] / factor;
    }
    for (int j = n - 2; j >= 0; j--) {
      factor = c[j]; c[j] = c[j] - b[j + 1] * factor; r[j] = r[j] - r[j + 1] * factor;
      factor = b[j]; r[j] = r[j] / factor;
    }
  }

  static constexpr int MAXTRIAL = 150;
  double fvec[M], f[M], p[M], a[M], b[M], c[M];

  // x[0..n-1] = T[1..]: the state before the first trial (NewtonRaphsonMethod::compute, newt_raph_func_fast.c:17-48)
  VIC_DEV void begin(double* x) {
    for (int i = 0; i < M; i++) { kappa_new[i] = 0; ice_new[i] = 0; Cs_new[i] = 0; a[i] = 0; b[i] = 0; c[i] = 0; }
    for (int i = 0; i < n; i++) x[i] = T0[i + 1];
  }

  // trial k of the iteration (newt_raph_func_fast.c:49-104); true when it has converged
  VIC_DEV bool trial(double* x, int k) {
    constexpr double TOLX = 1e-4, TOLF = 1e-1, R_MAX = 2.0, R_MIN = -5.0, RELAX1 = 0.9, RELAX2 = 0.7, RELAX3 = 0.2, EPS2 = 1e-4;
    fda_heat_eqn(x, fvec, -1);
    double errf = 0.0;
    for (int i = 0; i < n; i++) errf += fabs(fvec[i]);
    if (errf <= TOLF) return true;
    for (int j = 0; j < n; j++) {                          // fdjac3
      const double temp = x[j];
      double h = EPS2 * fabs(temp);
      if (h == 0) h = EPS2;
      x[j] = temp + h;
      h = x[j] - temp;
      fda_heat_eqn(x, f, j);
      x[j] = temp;
      b[j] = (f[j] - fvec[j]) / h;
      if (j != 0) c[j - 1] = (f[j - 1] - fvec[j - 1]) / h;
      if (j != n - 1) a[j + 1] = (f[j + 1] - fvec[j + 1]) / h;
    }
    for (int i = 0; i < n; i++) p[i] = -fvec[i];
    tridiag(a, b, c, p, n);
    double errx = 0.0;
    for (int i = 0; i < n; i++) {
      errx += fabs(p[i]);
      if (k > 10 && k <= 20 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX1;
      else if (k > 20 && k <= 60 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX2;
      else if (k > 60 && x[i] < R_MAX && x[i] > R_MIN) x[i] += p[i] * RELAX3;
      else x[i] += p[i];
    }
    return errx <= TOLX;
  }
};

}  // namespace vic

// included by vicgpu_api.hip after its list_append
__global__ __launch_bounds__(64) void vic_profile_solve_implicit(const vic::PArgs a, const vic::IArgs x) {
  using namespace vic;
  __shared__ int bcount[NBUCKET];
  const int lane = threadIdx.x;
  for (int b = lane; b < NBUCKET; b += 64) bcount[b] = a.count[b];
  __syncthreads();
  int ntot = 0;
  for (int b = 0; b < NBUCKET; b++) ntot += bcount[b];
  const int Nn = a.Nn;
  ImplicitSolver S;
  double xs[ImplicitSolver::M];
  bool have = false, exhausted = false;
  int hru = 0, k = 0;
#pragma unroll 1
  for (;;) {
    // lanes without a solve take the next entries of the list (one atomic per wave)
    const bool need = !have && !exhausted;
    const unsigned long long mneed = __ballot(need);
    if (mneed != 0) {
      const int lead = __ffsll((long long)mneed) - 1;
      int base = 0;
      if (lane == lead) base = atomicAdd(x.cursor, __popcll(mneed));
      base = __shfl(base, lead);
      const int slot = base + __popcll(mneed & ((1ull << lane) - 1ull));
      if (need) {
        if (slot >= ntot) exhausted = true;
        else {
          hru = profile_pick(a, bcount, slot);
          S.NOFLUX = a.NOFLUX; S.EXP_TRANS = a.EXP_TRANS;
          S.n = a.NOFLUX ? Nn - 1 : Nn - 2;
          const double* __restrict__ blk = a.pin + (size_t)hru * Nn * PREC;
          const double* __restrict__ im = x.pimp + (size_t)hru * Nn * PIMP;
          S.deltat = im[PI_ICE];                                    // node 0 (a boundary value) has no ice term: its slot carries delta_t
          const int cell = x.hpi[(size_t)HPI_CELL * x.nhru + hru];
          CellView cv{x.cell_params, x.ncell, cell, Nn, x.Nband};
          for (int q = 0; q < ImplicitSolver::M; q++) {
            const bool in = q < Nn;
            S.T0[q] = in ? ((q == 0) ? a.ts[hru] : blk[q * PREC + PR_T0]) : 0.0;
            S.moist[q] = in ? im[q * PIMP + PI_MOIST] : 0.0; S.ice[q] = in ? im[q * PIMP + PI_ICE] : 0.0;
            S.kappa[q] = in ? im[q * PIMP + PI_KAPPA] : 0.0; S.Cs[q] = in ? im[q * PIMP + PI_CS] : 0.0;
            S.mmn[q] = in ? cv.node(CPN_MAX_MOIST, q) : 0.0; S.bub[q] = in ? cv.node(CPN_BUBBLE, q) : 0.0; S.ex[q] = in ? cv.node(CPN_EXPT, q) : 0.0;
            S.al[q] = in ? cv.node(CPN_ALPHA, q) : 1.0; S.be[q] = in ? cv.node(CPN_BETA, q) : 1.0; S.ga[q] = in ? cv.node(CPN_GAMMA, q) : 1.0;
            S.zs[q] = in ? cv.node(CPN_ZSUM, q) : 0.0;
          }
          for (int l = 0; l < 3; l++) {
            S.sdm[l] = cv.lay(CPL_SOIL_DENS_MIN, l); S.bdm[l] = cv.lay(CPL_BULK_DENS_MIN, l); S.qz[l] = cv.lay(CPL_QUARTZ, l);
            S.sden[l] = cv.lay(CPL_SOIL_DENSITY, l); S.bden[l] = cv.lay(CPL_BULK_DENSITY, l); S.org[l] = cv.lay(CPL_ORGANIC, l);
            S.depth[l] = cv.lay(CPL_DEPTH, l);
          }
          const double Dp = cv.s(CP_DP);
          S.Bexp = a.EXP_TRANS ? (a.NOFLUX ? log(Dp + 1.) / (double)S.n : log(Dp + 1.) / (double)(S.n + 1)) : 0.0;
          S.Ts = S.T0[0];
          S.Tb = a.NOFLUX ? S.T0[S.n] : S.T0[S.n + 1];
          for (int q = 0; q < ImplicitSolver::M; q++) xs[q] = 0;
          S.begin(xs);
          k = 0;
          have = true;
        }
      }
    }
    if (!__any(have)) break;
    bool failed = false;
    if (have) {
      const bool conv = S.trial(xs, k);
      k++;
      if (conv) {
        const int ps = a.pslot[hru];
        double* __restrict__ rec = a.pout + (size_t)hru * pout_hru_stride(Nn) + ps * pout_stride(Nn);
        // the flags of the root find's most recent explicit solve (or none)
        const int le = x.lastexp[hru];
        unsigned long long meta = 1ull << 32;                  // ok, no fallback flags
        if (le >= 0) {
          const double* __restrict__ src = a.pout + (size_t)hru * pout_hru_stride(Nn) + le * pout_stride(Nn);
          meta = (unsigned long long)__double_as_longlong(src[Nn]) | (1ull << 32);
          if (le != ps)
            for (int q = 0; q < (Nn + 1) / 2; q++) rec[Nn + 1 + q] = src[Nn + 1 + q];     // the packed int counters
        } else {
          int* __restrict__ cnt = reinterpret_cast<int*>(rec + Nn + 1);
          for (int q = 0; q < Nn; q++) cnt[q] = 0;
        }
        rec[0] = S.T0[0];
        for (int q = 0; q < S.n; q++) rec[q + 1] = xs[q];
        if (!a.NOFLUX) rec[Nn - 1] = S.T0[Nn - 1];
        rec[Nn] = __longlong_as_double((long long)meta);
        a.pout[(size_t)hru * pout_hru_stride(Nn) + pout_key(Nn, ps)] = S.T0[0];
        if (le >= 0) x.lastexp[hru] = ps;
        have = false;
      } else if (k >= ImplicitSolver::MAXTRIAL) {
        failed = true;
        x.lastexp[hru] = a.pslot[hru];                         // the explicit kernel writes this slot next
        have = false;
      }
    }
    list_append(x.fb_list, x.fb_count, a.cap, failed, failed ? x.hkey[hru] : 0, hru);
  }
}
