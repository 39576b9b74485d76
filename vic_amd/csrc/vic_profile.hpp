// vic_profile.hpp — the explicit soil temperature profile solve as a kernel of its own (device only, gfx950).
//
// solve_T_profile + calc_soil_thermal_fluxes (frozen_soil.c:105-225, 305-505): Gauss-Seidel sweeps over the thermal
// nodes (<= 1000) until no node moves by more than 0.01 K; a node at or above 0 C (or with frozen soil off) has a
// closed-form update, a frozen node is the root of the node's heat balance with the freezing-point-depression curve in
// it (soil_thermal_eqn.c).  The solve is called once per residual evaluation of the Brent iteration on the surface
// temperature (func_surf_energy_bal.c:190) and carries most of the arithmetic of a FROZEN_SOIL step, with a trip count
// that differs from HRU to HRU.  It runs here on a compacted, keyed work list with persistent waves:
//
//   * lane = one profile solve; the reference's loop nest runs as written -- sweeps { nodes { node solve } } -- with the
//     wave in lock step: the work lists are keyed by the HRU's number of frozen nodes (vic_fd_stage), so the 64 solves of a
//     wave have their frozen nodes in the same places; a lane whose Gauss-Seidel iteration has ended writes its record
//     and takes the next item at the gate before the following sweep (the sweep count is per lane);
//   * what a node visit needs is constant over all sweeps of a solve (and over all solves of one Brent iteration on
//     Tsurf): the item block (vic_surface.hpp) holds it pre-folded, 10 doubles per node.  The 10-node instantiation
//     loads the block ONCE per solve into registers (node loop unrolled, 2 waves per SIMD) -- the sweeps then touch no
//     memory at all; the generic instantiation (any other node count) reads the record at each visit;
//   * the frozen-node root.  The reference brackets the root in T0 +- 0.25 K (expanding by +-10 K up to five times) and
//     runs Brent's method to a tolerance of 1e-7 K (root_brent.c).  The node balance is  f(T) = N - S*T + E*ice(T),
//     strictly decreasing and continuous in T, so the root the reference converges to is THE root of f, whatever the
//     method: here a bracket-safeguarded Newton iteration from the previous sweep's temperature, with the derivative of
//     the freezing curve in closed form (it costs no further transcendental), stopped when the step is below 1e-8 K
//     (the error left is below 1e-13 K) -- 3-4 evaluations instead of Brent's 8-12, and no 17-28-evaluation tail at the
//     0 C kink.  N >= 0 means the root lies at or above 0 C where ice = 0: T = N / S without iterating.  What the
//     reference's failure modes mean for a monotone f is kept: no root within T0 +- 50.25 K -> the node falls back to T0
//     (TFALLBACK) or the solve reports an error.  One case is NOT monotone: the reference's "cold nose" variant of the
//     residual at node 1 when |TL - TU| > 5 K (soil_thermal_eqn.c:103-110) switches a flux term off discontinuously;
//     those visits run the reference's Brent (BrentLean, vic_math.hpp) on the reference's residual.
//
// Per lane the closed-form updates execute the reference's sequence of floating-point operations; the frozen-node roots
// agree with the reference's to the reference's own stopping tolerance (measured: tests/test_gpu_parity.py).
#pragma once
#include "vic_surface.hpp"

namespace vic {

// Work lists are kept in NBUCKET segments by a per-HRU key (the number of frozen nodes at the start of the step, and
// whether a node sits within SOIL_DT of 0 C -- see vic_fd_stage): the profile kernel takes the segments one after the
// other, most expensive first, so that the HRUs a wave works on at any time have the same nodes frozen.
#ifndef PROFILE_NBUCKET
#define PROFILE_NBUCKET (2 * (VIC_MAX_NODES + 2))
#endif
constexpr int NBUCKET = PROFILE_NBUCKET;
static_assert(NBUCKET <= 64, "one lane per work-list segment when the pending total is summed");

struct PArgs {
  const double* __restrict__ pin;    // item blocks [nhru][Nn][PREC]
  const double* __restrict__ ts;     // trial surface temperature [nhru]
  double* __restrict__ pout;         // [nhru][pout_hru_stride(Nn)]: two solution records + the trial temperature of each
  const int* __restrict__ pslot;     // [nhru] record the next solve of this HRU is written to
  const int* __restrict__ list;      // HRUs to solve: NBUCKET segments of `cap` entries
  const int* __restrict__ count;     // entries per segment [NBUCKET]
  int cap;
  int* next;                         // work-list cursor (zero at launch; the evaluation kernel clears it again)
  int* count_zero;                   // segment counters [NBUCKET] of the list the following evaluation kernel appends to (cleared here)
  int* evalonly_zero;                // counter of HRUs whose next evaluation needs no solve (cleared here)
  // the number of evaluations pending after this kernel = entries of the round's work list + evaluation-only HRUs of the round
  // before, summed here (block 0) into ONE word on its own cache line for the waves of the evaluation kernel
  const int* pend_counts;            // [NBUCKET] (the round's list; not `count` when this launch works on IMPLICIT's fall-back list)
  const int* pend_eo;
  int* pend_out;
  int Nn, NOFLUX, EXP_TRANS, TFALLBACK;
  const int* jl;                     // QUICK_SOLVE: [nhru] nodes 1 .. jl - 1 are solved (calc_surf_energy_bal.c:289-299), or null: all
};

// One solution record: T[Nn], {fbmask | ok << 32}, int fallback counts [Nn].  Every HRU keeps the records of its last
// two solves together with the trial temperatures they belong to: the Brent iteration on Tsurf ends with one more
// evaluation AT the root, which is one of the last two trial points unless the solver fell back -- the same inputs give
// the same profile bit for bit, so that solve is looked up instead of repeated (vic_surf_eval).
__host__ __device__ inline int pout_stride(int Nn) { return Nn + 1 + (Nn + 1) / 2; }
__host__ __device__ inline int pout_hru_stride(int Nn) { return 2 * pout_stride(Nn) + 2; }
__host__ __device__ inline int pout_key(int Nn, int slot) { return 2 * pout_stride(Nn) + slot; }

#ifndef LOCKSTEP_GATE_LANES
#define LOCKSTEP_GATE_LANES 16
#endif
constexpr int LOCKSTEP_GATE = LOCKSTEP_GATE_LANES;   // idle lanes that must be waiting before the write-back / fetch section runs

constexpr double NODE_ROOT_RANGE = SOIL_DT + Brent::MAXTRIES * Brent::TSTEP;   // the reference finds roots within T0 +- this
constexpr double NODE_NEWTON_TOL = 1.e-8;                                        // last Newton step, K
#ifndef VIC_NEWTON_PREDICTOR
#define VIC_NEWTON_PREDICTOR 1
#endif
#ifndef VIC_NOSE_CLASSIFY
#define VIC_NOSE_CLASSIFY 1
#endif
#ifndef VIC_PREDICT_TOL2
#define VIC_PREDICT_TOL2 1.e-7
#endif
#ifndef VIC_NEWTON_ACCEPT
#define VIC_NEWTON_ACCEPT 2.e-11
#endif
constexpr double NODE_PREDICT_TOL2 = VIC_PREDICT_TOL2;    // the predictor stops when step^2 <= this * |T| (its error is then < 1e-6 K)
constexpr int NODE_PREDICT_MAXIT = 12;
constexpr double NODE_NEWTON_ACCEPT = VIC_NEWTON_ACCEPT;  // (1 + |Y|) s^2 / |T| below this: the error left after the step is below 1e-11 K
constexpr int NODE_NEWTON_MAXIT = 200;

// The per-node constants of one record (PR_AT0 .. PR_EMM), see vic_surface.hpp
struct NodeK {
  double AT0, B, C, D, EI, S, G, Y, EM, EMM;
  VIC_DEV void load(const double* __restrict__ r) {
    AT0 = r[PR_AT0]; B = r[PR_B]; C = r[PR_C]; D = r[PR_D]; EI = r[PR_EI]; S = r[PR_S]; G = r[PR_G]; Y = r[PR_Y];
    EM = r[PR_EMOIST]; EMM = r[PR_EMM];
  }
  // E * ice(T) for T < 0 (soil_thermal_eqn.c:66-71 with maximum_unfrozen_water), E*u(T) and whether the curve is active
  VIC_DEV double eice(double T, double& Eu, bool& curved) const {
    Eu = G * exp(Y * ln_pos(-T));
    curved = true;
    if (Eu > EMM) { Eu = EMM; curved = false; }        // u > max_moist
    double Ei = EM - Eu;
    if (Ei < 0.) { Ei = 0.; curved = false; }
    if (Ei > EMM) { Ei = EMM; curved = false; }
    return Ei;
  }
};

// One Gauss-Seidel visit of a node for all lanes of the wave (call convergently; `sweeping` = this lane takes part).
// NODE1: the visit is node 1, where the reference's residual has its cold-nose variant.
// Returns the new node temperature; failed: the reference's root finder would have returned ERROR.
template <bool NODE1, bool NEWTON>
VIC_DEV double node_visit(bool sweeping, bool frozen_on, bool EXP_TRANS, const NodeK& K, double oldT, double Tdn, double Tup, double T0j,
                          bool& failed) {
  // numerator of the closed-form update in the reference's operation order (frozen_soil.c:388-393, 429-436)
  double N;
  if (!EXP_TRANS) N = K.AT0 + K.B * (Tdn - Tup) + K.C * Tdn + K.D * Tup + K.EI;
  else N = K.AT0 + K.B * (Tdn - Tup) + K.C * (Tdn + Tup) - K.D * (Tdn - Tup) + K.EI;
  const bool fz = sweeping && frozen_on && oldT < 0;
  // `steep`: visits that run the reference's Brent iteration -- all frozen visits when NEWTON is off, otherwise only the
  // discontinuous cold-nose case at node 1
  bool steep = false;
#ifdef VIC_ABL_NONOSE          // ablation builds (tools/exp): wrong results, they only attribute time
  const bool nose = false;
#else
  const bool nose = NODE1 && fabs(Tdn - Tup) > 5.;
#endif
  if (!NEWTON) steep = fz;
#if !VIC_NOSE_CLASSIFY
  else if (NODE1) steep = fz && nose;
#endif
  double x = N / K.S;                       // unfrozen node, or root in T >= 0 where ice = 0
  failed = false;
  PROF_WAVE(18); PROF_VOTE(19, fz);

  // ---- safeguarded Newton on f(T) = N - S T + E ice(T) in T < 0 (f(0) = N < 0 there: 0 is an upper bound of the root)
  bool act = NEWTON && fz && !steep && N < 0;
#ifdef VIC_ABL_NONEWTON
  act = false;
#endif
  if (NEWTON && __any(act)) {
    PROF_WAVE(22);
    x = act ? oldT : x;
#if VIC_NEWTON_PREDICTOR
    // Predictor: the same Newton iteration with the freezing curve through the hardware's single-precision log2 / exp2
    // (relative error ~3e-7, ~35 instructions per iteration instead of ~130) until the step is small enough for the
    // double-precision iteration below to finish in one evaluation.  It only moves the starting point: its bracket is its
    // own (a sign of f within the approximation's noise must not narrow the bracket of the real iteration), and it gives
    // up after NODE_PREDICT_MAXIT iterations wherever it stands.
    {
      const float Yf = (float)K.Y;
      double plo = -1.e300, phi = 0.0;
      bool pre = act;
      int pit = 0;
      PROF_T0(t_pre);
      while (__any(pre)) {
        PROF_WAVE(23); PROF_VOTE(24, pre);
        if (pre) {
          double Eu = K.G * pow_pos_approx(-x, Yf);
          bool curved = true;
          if (Eu > K.EMM) { Eu = K.EMM; curved = false; }
          double Ei = K.EM - Eu;
          if (Ei < 0.) { Ei = 0.; curved = false; }
          if (Ei > K.EMM) { Ei = K.EMM; curved = false; }
          const double f = N - K.S * x + Ei;
          if (f > 0) plo = x; else phi = x;
          const double den = K.S * x + (curved ? K.Y * Eu : 0.0);
          const double step = f * x * rcp_refined(den);
          double xn = x + step;
          pit++;
          // as below; the iterate stays negative and finite whatever the approximation does (phi <= 0, both ends finite or the
          // doubling step)
          if (!(xn > plo && xn < phi)) xn = (plo > -1.e299) ? 0.5 * (plo + phi) : x + x - 1.0;
          if (step * step <= NODE_PREDICT_TOL2 * fabs(x) || pit >= NODE_PREDICT_MAXIT) pre = false;
          x = xn;
        }
      }
      PROF_ADD(22, t_pre);
    }
#endif
    double lo = -1.e300, hi = 0.0;
    int it = 0;
    PROF_T0(t_nw);
    while (__any(act)) {
      PROF_WAVE(20); PROF_VOTE(21, act);
      if (act) {
        double Eu;
        bool curved;
        const double Ei = K.eice(x, Eu, curved);
        const double f = N - K.S * x + Ei;
        if (f > 0) lo = x; else hi = x;
        const double den = K.S * x + (curved ? K.Y * Eu : 0.0);          // = x f'(x), negative
#if VIC_NEWTON_PREDICTOR
        const double step = f * x * rcp_refined(den);                     // a step's last digits do not matter (see below)
#else
        const double step = f * x / den;
#endif
        double xn = x + step;
        it++;
        if (fabs(step) <= NODE_NEWTON_TOL) act = false;                      // converged: the step is taken as it is
#if VIC_NEWTON_PREDICTOR
        // On the smooth branch of the curve the error left after a Newton step s is |f''/(2 f')| s^2 <= (1 + |Y|) s^2 / (2 |x|)
        // (f' = -S - Y Eu / x, f'' = -Y (Y - 1) Eu / x^2, |x| the smaller end of the step: |s| <= 0.05 |x| keeps it within 5 %,
        // the bound below has that margin).  When that bound is below the tolerance and the step stays
        // clear of the kink where the ice content reaches zero (Eu = E moist; Eu is convex in T, so twice its linear change is
        // an upper bound for small steps), the step is taken without another evaluation to confirm it.
        else if (curved && fabs(step) <= 0.05 * fabs(x) && (1. + fabs(K.Y)) * step * step <= NODE_NEWTON_ACCEPT * fabs(x)
                 && Eu * (1. + 2.2 * fabs(K.Y * step) * rcp_refined(fabs(x))) < K.EM) act = false;
#endif
        else {
          // a step that leaves the bracket (a kink of the curve between x and the root) is replaced by a bisection; a step
          // down can only leave it once a lower bound is known
          if (!(xn > lo && xn < hi)) xn = (lo > -1.e299) ? 0.5 * (lo + hi) : x + x - 1.0;
          if (it >= NODE_NEWTON_MAXIT) { act = false; failed = true; }
        }
        x = xn;
      }
    }
    PROF_ADD(23, t_nw);
  }
#if VIC_NOSE_CLASSIFY
  // ---- node 1, cold nose (soil_thermal_eqn.c:57-72, 84-93).  With |TL - TU| > 5 K the reference drops the flux term
  // ft1 = B (TL - TU) from the residual where ft1 < 0, T < min(TL, TU), ft2(T) > 0 and |ft1| > |ft2(T)|; ft2 decreases in T, so
  // that is an interval Tb < T < Thi below both neighbours (ft2 > 0 holds there by itself; EXP_TRANS: below its own zero as
  // well), on which the residual is g(T) + |ft1| instead of g(T) = N - S T + E ice(T).  g decreases, so wherever the root
  // r of g lies at or above Thi the residual has the sign of g everywhere and r is its only sign change: what the Newton
  // iteration above has found is what the reference's Brent iteration converges to.  Only when r lies below Thi (the node
  // would end up colder than both neighbours) can the two branches offer several sign changes; those visits replay the
  // reference's iteration on the reference's residual.  (A margin of 1e-6 K sends borderline roots to the replay.)
  if (NEWTON && NODE1) {
    const double ft1n = K.B * (Tdn - Tup);
    bool sp = fz && nose && ft1n < 0;
    if (__any(sp)) {
      double Thi = fmin(Tdn, Tup), Tb;
      if (!EXP_TRANS) Tb = (K.C * Tdn + K.D * Tup + ft1n) / (K.C + K.D);
      else {
        const double num = K.C * (Tdn + Tup) - K.D * (Tdn - Tup);
        Thi = fmin(Thi, num / (2. * K.C));
        Tb = (num + ft1n) / (2. * K.C);
      }
      steep = sp && Tb < Thi && !failed && x < Thi + 1.e-6;
    }
  }
#endif
  // ---- the reference's Brent iteration (root_brent.c:97-337) on the reference's residual
  if (NODE1 || !NEWTON) {
    if (__any(steep)) {
      PROF_WAVE(25); PROF_VOTE(26, steep);
      PROF_T0(t_br);
      BrentLean br;
      br.phase = BrentLean::DONE;
      if (steep) br.start(T0j - SOIL_DT, T0j + SOIL_DT);
      const double ft1 = K.B * (Tdn - Tup);
      while (__any(steep && !br.finished())) {
        if (steep && !br.finished()) {
          const double T = br.x;
          double Ei = 0.;
          if (T < 0.) { double Eu; bool cv_; Ei = K.eice(T, Eu, cv_); }
          double v = N - K.S * T + Ei;
          const double ft2 = !EXP_TRANS ? K.C * (Tdn - T) - K.D * (T - Tup) : K.C * (Tdn - 2. * T + Tup) - K.D * (Tdn - Tup);
          if (NODE1 && nose && (T < Tdn && T < Tup) && (ft1 < 0 && ft2 > 0) && fabs(ft1) > fabs(ft2)) v -= ft1;
          br.advance(v);
          PROF_LANE(27);
        }
      }
      if (steep) { x = br.b; if (br.phase == BrentLean::FAILED) failed = true; }
      PROF_ADD(25, t_br);
    }
  }
  // the reference searches T0 +- 0.25 K, widened by 10 K up to five times (root_brent.c:183-248)
  if (fz && !steep && !(fabs(x - T0j) <= NODE_ROOT_RANGE)) failed = true;
#ifdef VIC_DEBUG_NODE
  if (failed) printf("node failed: NODE1 %d steep %d N %g S %g x %.17g T0j %g oldT %g Tdn %g Tup %g G %g Y %g EM %g EMM %g\n", (int)NODE1, (int)steep, N, K.S, x, T0j, oldT, Tdn, Tup, K.G, K.Y, K.EM, K.EMM);
#endif
  return x;
}

// the end of a solve: cold-nose hack, non-convergence, fallback bookkeeping (frozen_soil.c:470-493); T / T0 accessors differ
// between the two kernels, so this is a macro-free helper over plain arrays of the lane's column
template <int NN>
VIC_DEV void profile_finish(int Nn, bool TFALLBACK, bool converged, bool& ok, unsigned& fbmask, double* T, const double* T0, int* cnt_add) {
#pragma unroll
  for (int k = 0; k < NN; k++) cnt_add[k] = 0;
  if (ok && TFALLBACK) {      // cold-nose hack, frozen_soil.c:470-484 (sic: Tlast[j+1] - T[j]); Tlast == T0
#pragma unroll
    for (int k = 1; k < NN - 1; k++) {
      if (k < Nn - 1) {
        const double Tk = T[k], Tm = T[k - 1], Tp = T[k + 1], Lk = T0[k], Lm = T0[k - 1], Lp = T0[k + 1];
        if (Lm - Lk > 0 && Lp - Tk > 0 && (Tm - Tk) - (Lm - Lk) > 0 && (Tp - Tk) - (Lp - Lk) > 0) {
          T[k] = 0.5 * (Tm + Tp);
          fbmask |= (1u << k);
          cnt_add[k] += 1;
        }
      }
    }
  }
  if (ok && !converged) {
    if (TFALLBACK) {
#pragma unroll
      for (int k = 0; k < NN; k++)
        if (k < Nn) { T[k] = T0[k]; cnt_add[k] += 1; }
      fbmask |= (Nn >= 32) ? 0xFFFFFFFFu : ((1u << Nn) - 1u);
    } else ok = false;
  }
}

// work-list slot -> HRU (segments are taken highest key first: most frozen nodes = most work)
VIC_DEV int profile_pick(const PArgs& a, const int* bcount, int slot) {
  int rem = slot, found = 0;
#pragma unroll 1
  for (int b = NBUCKET - 1; b >= 0; b--) {
    const int cb = bcount[b];
    if (rem < cb) { found = b * a.cap + rem; break; }
    rem -= cb;
  }
  return a.list[found];
}

// ------------------------------------------------------------------------------------------------
// 10 nodes (the sample global file's and BASELINE's node count): node constants and temperatures in registers
// ------------------------------------------------------------------------------------------------
#ifndef PROFILE_REG_WAVES
#define PROFILE_REG_WAVES 2
#endif
template <int NN, bool NEWTON>
__global__ __launch_bounds__(64) VIC_WAVES_PER_EU(PROFILE_REG_WAVES, PROFILE_REG_WAVES) void vic_profile_solve_reg(const PArgs a) {
  __shared__ int bcount[NBUCKET];
  __shared__ double T0l[NN * 64];
  __shared__ double Kl[3 * (NN - 2) * 64];             // G, Y, E*moist of the interior nodes: only frozen visits read them
  const int lane = threadIdx.x;
#define T0(j) T0l[(j) * 64 + lane]
#define KL(f, j) Kl[((f) * (NN - 2) + (j) - 1) * 64 + lane]
  for (int b = lane; b < NBUCKET; b += 64) bcount[b] = a.count[b];
  if (blockIdx.x == 0) {
    for (int b = lane; b < NBUCKET; b += 64) a.count_zero[b] = 0;
    if (lane == 0) *a.evalonly_zero = 0;
    int v = (lane < NBUCKET) ? a.pend_counts[lane] : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl(v, lane ^ off);
    if (lane == 0) *a.pend_out = v + *a.pend_eo;
  }
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int b = 0; b < NBUCKET; b++) n += bcount[b];
  if ((int)blockIdx.x * 64 >= n) return;

  constexpr int Nn = NN;
  const int jlast = a.NOFLUX ? Nn : Nn - 1;
  const int MAXIT = 1000;
  const double threshold = 1.e-2;
  const bool EXP_TRANS = a.EXP_TRANS != 0;

  bool have = false, sweeping = false, converged = false, ok = true, frozen_on = false;
  int hru = 0, ps = 0, it = 1, jl_lane = 0;
  unsigned fbmask = 0;
  int evcnt = 0;                                       // 1 once a node solver of this solve has fallen back (see below)
  double T[NN];
  // node constants of the interior nodes 1 .. NN-2: seven in registers, three in LDS (see KL)
  double kAT0[NN - 1], kB[NN - 1], kC[NN - 1], kD[NN - 1], kEI[NN - 1], kS[NN - 1], kEMM[NN - 1];
#pragma unroll
  for (int k = 0; k < NN; k++) T[k] = 0;
#pragma unroll
  for (int k = 0; k < NN - 1; k++) { kAT0[k] = 0; kB[k] = 0; kC[k] = 0; kD[k] = 0; kEI[k] = 0; kS[k] = 1; kEMM[k] = 0; }
  bool more = true;                                    // wave-uniform: the work list has items nobody has taken yet
  while (true) {
    const unsigned long long idle = __ballot(!have);
    if (more && (__popcll(idle) >= LOCKSTEP_GATE || idle == ~0ull)) {
      PROF_T0(t_gate);
      const int nidle = __popcll(idle), leader = __ffsll((long long)idle) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(a.next, nidle);
      base = __builtin_amdgcn_readlane(base, leader);
      if (base + nidle >= n) more = false;
      const int slot = base + __popcll(idle & ((1ull << lane) - 1ull));
      if (!have && slot < n) {
        hru = profile_pick(a, bcount, slot);
        // an item block starts on a 16-byte boundary (10 nodes x 11 doubles = 880 bytes): lets pairs of words load as one access
        const double* __restrict__ blk = static_cast<const double*>(__builtin_assume_aligned(a.pin + (size_t)hru * Nn * PREC, 16));
        ps = a.pslot[hru];
        jl_lane = a.jl ? a.jl[hru] : jlast;
        frozen_on = blk[PR_AT0] != 0.0;
        const double Ts = a.ts[hru];
#pragma unroll
        for (int k = 0; k < NN; k++) { const double t = (k == 0) ? Ts : blk[k * PREC + PR_T0]; T0(k) = t; T[k] = t; }
#pragma unroll
        for (int k = 1; k < NN - 1; k++) {
          const double* __restrict__ r = blk + k * PREC;
          kAT0[k] = r[PR_AT0]; kB[k] = r[PR_B]; kC[k] = r[PR_C]; kD[k] = r[PR_D]; kEI[k] = r[PR_EI]; kS[k] = r[PR_S]; kEMM[k] = r[PR_EMM];
          KL(0, k) = r[PR_G]; KL(1, k) = r[PR_Y]; KL(2, k) = r[PR_EMOIST];
        }
        have = true; fbmask = 0; evcnt = 0; ok = true; it = 1;
        converged = (jlast <= 1);
        sweeping = !converged;
      }
      PROF_ADD(20, t_gate);
    }
    if (!__any(have)) break;
    PROF_T0(t_sweep);
    {
      double maxdiff = threshold;
      PROF_WAVE(16); PROF_VOTE(17, sweeping);
#pragma unroll
      for (int j = 1; j < NN; j++) {
        if (j < jlast) {
          const double oldT = T[j];
          const double Tdn = (j == Nn - 1) ? oldT : T[(j + 1 < NN) ? j + 1 : j], Tup = T[j - 1];
          const double T0j = T0(j);
          NodeK Kj;
          if (j == NN - 1) Kj.load(a.pin + ((size_t)hru * Nn + j) * PREC);      // the bottom node (NOFLUX only) is not cached
          else {
            constexpr int q = 0;
            const int jj = (j < NN - 1) ? j : 1 + q;
            Kj.AT0 = kAT0[jj]; Kj.B = kB[jj]; Kj.C = kC[jj]; Kj.D = kD[jj]; Kj.EI = kEI[jj]; Kj.S = kS[jj]; Kj.EMM = kEMM[jj];
            Kj.G = KL(0, jj); Kj.Y = KL(1, jj); Kj.EM = KL(2, jj);
          }
          bool failed;
          double newT;
          const bool swj = sweeping && j < jl_lane;          // QUICK_SOLVE: the lane's column ends earlier
          if (j == 1) newT = node_visit<true, NEWTON>(swj, frozen_on, EXP_TRANS, Kj, oldT, Tdn, Tup, T0j, failed);
          else newT = node_visit<false, NEWTON>(swj, frozen_on, EXP_TRANS, Kj, oldT, Tdn, Tup, T0j, failed);
          if (swj) {
            if (failed) {
              if (a.TFALLBACK) {
                // node fallback: T0 and a count.  The counters live in the solution record (rare path): the first event of a
                // solve clears them, the end of the solve adds its own events or, without any, writes them whole
                int* cnt = reinterpret_cast<int*>(a.pout + (size_t)hru * pout_hru_stride(Nn) + ps * pout_stride(Nn) + Nn + 1);
                if (evcnt == 0) {
#pragma unroll
                  for (int k = 0; k < NN; k++) cnt[k] = 0;
                }
                evcnt = 1;
                cnt[j] += 1;
                newT = T0j; fbmask |= (1u << j);
              } else { ok = false; sweeping = false; }
            }
            if (sweeping) {
              T[j] = newT;
              const double diff = fabs(oldT - newT);
              if (diff > maxdiff) maxdiff = diff;
            }
          }
        }
      }
      if (sweeping) {                               // end of a Gauss-Seidel sweep (frozen_soil.c:466)
        if (maxdiff <= threshold) { converged = true; sweeping = false; }
        else if (it >= MAXIT) sweeping = false;
        else it++;
      }
    }
    PROF_ADD(21, t_sweep);
    PROF_T0(t_fin);
    if (have && !sweeping) {                        // this lane's item is through: finish it and free the lane
      double T0v[NN];
      int cadd[NN];
#pragma unroll
      for (int k = 0; k < NN; k++) T0v[k] = T0(k);
      profile_finish<NN>(a.jl ? (jl_lane + 1 < Nn ? jl_lane + 1 : Nn) : Nn, a.TFALLBACK != 0, converged, ok, fbmask, T, T0v, cadd);
      double* __restrict__ rec = a.pout + (size_t)hru * pout_hru_stride(Nn) + ps * pout_stride(Nn);
      int* __restrict__ cnt = reinterpret_cast<int*>(rec + Nn + 1);
#pragma unroll
      for (int k = 0; k < NN; k++) rec[k] = T[k];
      rec[Nn] = __longlong_as_double((long long)((unsigned long long)fbmask | ((unsigned long long)(ok ? 1 : 0) << 32)));
      // fallback counters: zero unless something happened in this solve (node-solver fallbacks were written as they happened)
#pragma unroll
      for (int k = 0; k < NN; k++) {
        if (evcnt == 0) cnt[k] = cadd[k];
        else cnt[k] += cadd[k];
      }
      a.pout[(size_t)hru * pout_hru_stride(Nn) + pout_key(Nn, ps)] = T0(0);
      have = false;
    }
    PROF_ADD(24, t_fin);
  }
#undef T0
#undef KL
}

// ------------------------------------------------------------------------------------------------
// any node count: node columns in LDS, the node's record read at each visit
// ------------------------------------------------------------------------------------------------
#ifndef LS_WAVES
#define LS_WAVES 2
#endif
template <int NN, bool NEWTON>
__global__ __launch_bounds__(64) VIC_WAVES_PER_EU(LS_WAVES, LS_WAVES) void vic_profile_solve_lockstep(const PArgs a) {
  __shared__ int bcount[NBUCKET];
  __shared__ double Tl[NN * 64];
  __shared__ double T0l[NN * 64];
  const int lane = threadIdx.x;
#define T(j) Tl[(j) * 64 + lane]
#define T0(j) T0l[(j) * 64 + lane]
  for (int b = lane; b < NBUCKET; b += 64) bcount[b] = a.count[b];
  if (blockIdx.x == 0) {
    for (int b = lane; b < NBUCKET; b += 64) a.count_zero[b] = 0;
    if (lane == 0) *a.evalonly_zero = 0;
    int v = (lane < NBUCKET) ? a.pend_counts[lane] : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl(v, lane ^ off);
    if (lane == 0) *a.pend_out = v + *a.pend_eo;
  }
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int b = 0; b < NBUCKET; b++) n += bcount[b];
  if ((int)blockIdx.x * 64 >= n) return;

  const int Nn = (NN == VIC_MAX_NODES) ? a.Nn : NN;
  const int jlast = a.NOFLUX ? Nn : Nn - 1;
  const int MAXIT = 1000;
  const double threshold = 1.e-2;
  const bool EXP_TRANS = a.EXP_TRANS != 0;

#define LS_REC() (a.pout + (size_t)hru * pout_hru_stride(Nn) + ps * pout_stride(Nn))
#define LS_CNT(j) (reinterpret_cast<int*>(LS_REC() + Nn + 1)[j])
  bool have = false, sweeping = false, converged = false, ok = true, frozen_on = false;
  int hru = 0, ps = 0, it = 1, jl_lane = 0;
  unsigned fbmask = 0;
  const double* __restrict__ blk = a.pin;
  bool more = true;
  while (true) {
    const unsigned long long idle = __ballot(!have);
    if (more && (__popcll(idle) >= LOCKSTEP_GATE || idle == ~0ull)) {
      const int nidle = __popcll(idle), leader = __ffsll((long long)idle) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(a.next, nidle);
      base = __builtin_amdgcn_readlane(base, leader);
      if (base + nidle >= n) more = false;
      const int slot = base + __popcll(idle & ((1ull << lane) - 1ull));
      if (!have && slot < n) {
        hru = profile_pick(a, bcount, slot);
        blk = a.pin + (size_t)hru * Nn * PREC;
        ps = a.pslot[hru];
        jl_lane = a.jl ? a.jl[hru] : jlast;
        frozen_on = blk[PR_AT0] != 0.0;
        const double Ts = a.ts[hru];
#pragma unroll
        for (int k = 0; k < NN; k++)
          if (k < Nn) { const double t = (k == 0) ? Ts : blk[k * PREC + PR_T0]; T0(k) = t; T(k) = t; LS_CNT(k) = 0; }
        have = true; fbmask = 0; ok = true; it = 1;
        converged = (jlast <= 1);
        sweeping = !converged;
      }
    }
    if (!__any(have)) break;
    {
      double maxdiff = threshold;
      PROF_WAVE(16); PROF_VOTE(17, sweeping);
#pragma unroll 1
      for (int j = 1; j < NN; j++) {
        if (j < jlast) {
          double oldT = 0, Tdn = 0, Tup = 0, T0j = 0;
          NodeK K;
          K.AT0 = 0; K.B = 0; K.C = 0; K.D = 0; K.EI = 0; K.S = 1; K.G = 0; K.Y = 0; K.EM = 0; K.EMM = 0;
          if (sweeping) {
            oldT = T(j);
            Tdn = (j == Nn - 1) ? oldT : T((j + 1 < NN) ? j + 1 : j); Tup = T(j - 1);
            T0j = T0(j);
            K.load(blk + j * PREC);
          }
          bool failed;
          double newT;
          const bool swj = sweeping && j < jl_lane;          // QUICK_SOLVE: the lane's column ends earlier
          if (j == 1) newT = node_visit<true, NEWTON>(swj, frozen_on, EXP_TRANS, K, oldT, Tdn, Tup, T0j, failed);
          else newT = node_visit<false, NEWTON>(swj, frozen_on, EXP_TRANS, K, oldT, Tdn, Tup, T0j, failed);
          if (swj) {
            if (failed) {
              if (a.TFALLBACK) { newT = T0j; fbmask |= (1u << j); LS_CNT(j) += 1; }
              else { ok = false; sweeping = false; }
            }
            if (sweeping) {
              T(j) = newT;
              const double diff = fabs(oldT - newT);
              if (diff > maxdiff) maxdiff = diff;
            }
          }
        }
      }
      if (sweeping) {                               // end of a Gauss-Seidel sweep (frozen_soil.c:466)
        if (maxdiff <= threshold) { converged = true; sweeping = false; }
        else if (it >= MAXIT) sweeping = false;
        else it++;
      }
    }
    if (have && !sweeping) {                        // this lane's item is through: finish it and free the lane
      const int nq = a.jl ? (jl_lane + 1 < Nn ? jl_lane + 1 : Nn) : Nn;      // the column this solve covered (QUICK_SOLVE: shorter)
      if (ok && a.TFALLBACK) {      // cold-nose hack, frozen_soil.c:470-484 (sic: Tlast[j+1] - T(j)); Tlast == T0
#pragma unroll 1
        for (int k = 1; k < nq - 1; k++) {
          const double Tk = T(k), Tm = T(k - 1), Tp = T(k + 1), Lk = T0(k), Lm = T0(k - 1), Lp = T0(k + 1);
          if (Lm - Lk > 0 && Lp - Tk > 0 && (Tm - Tk) - (Lm - Lk) > 0 && (Tp - Tk) - (Lp - Lk) > 0) {
            T(k) = 0.5 * (Tm + Tp);
            fbmask |= (1u << k);
            LS_CNT(k) += 1;
          }
        }
      }
      if (ok && !converged) {
        if (a.TFALLBACK) {
#pragma unroll 1
          for (int k = 0; k < nq; k++) { T(k) = T0(k); LS_CNT(k) += 1; }
          fbmask |= (nq >= 32) ? 0xFFFFFFFFu : ((1u << nq) - 1u);
        } else ok = false;
      }
      double* __restrict__ rec = LS_REC();
#pragma unroll 1
      for (int k = 0; k < Nn; k++) rec[k] = T(k);
      rec[Nn] = __longlong_as_double((long long)((unsigned long long)fbmask | ((unsigned long long)(ok ? 1 : 0) << 32)));
      a.pout[(size_t)hru * pout_hru_stride(Nn) + pout_key(Nn, ps)] = T0(0);
      have = false;
    }
  }
}
#undef T
#undef T0
#undef LS_REC
#undef LS_CNT

}  // namespace vic
