// vic_profile.hpp — the explicit soil temperature profile solve as a kernel of its own (device only, gfx950).
//
// solve_T_profile + calc_soil_thermal_fluxes (frozen_soil.c:105-225, 305-505): Gauss-Seidel sweeps over the thermal
// nodes (<= 1000), each frozen node a Brent root find (root_brent.c, <= 1000 residual evaluations with a pow,
// soil_thermal_eqn.c).  It is called once per residual evaluation of the Brent iteration on the surface temperature
// (func_surf_energy_bal.c:190) and carries > 90 % of the arithmetic of a FROZEN_SOIL step, with a trip count that
// differs from HRU to HRU by an order of magnitude.  It needs ~25 live doubles where the rest of the step needs
// several hundred, so it runs here at many waves per SIMD, on a compacted work list, and balances its own load:
//
//   * one lane = one solve at a time; the loop nest is flattened into ONE wave loop in which every lane carries its
//     own (sweep, node, Brent) state and advances by one unit of work per trip -- a lane on an unfrozen node moves on
//     while its neighbours iterate their Brent;
//   * the waves are persistent and pull work: a lane that has finished its solve waits until a quarter of the wave
//     has (or nothing else can run), then the waiting lanes write their results and take the next entries of the
//     work list with one wave-aggregated atomic.  All waves therefore drain together, and the rare, memory-bound
//     write-back / load section runs for 16 lanes at a time instead of for one or two lanes on most trips;
//   * node temperatures (the only per-lane-indexed data that changes) live in LDS as [node][lane]; the constant
//     per-node records are read from the item block (vic_surface.hpp) when a lane enters a node.
//
// Per lane the sequence of floating-point operations is the reference's, up to the three last-bit exceptions listed at
// SoilThermalEqn (vic_surface.hpp) and BrentLean (vic_math.hpp).
#pragma once
#include "vic_surface.hpp"

namespace vic {

// Work lists are kept in NBUCKET segments by a per-HRU key (the number of frozen nodes at the start of the step, and
// whether a node sits within SOIL_DT of 0 C -- see vic_fd_stage): the
// profile kernel takes the segments one after the other, most expensive first, so that the HRUs a wave works on at any time
// have the same nodes in Brent solves and need about the same number of trips -- measured on the bench workload, waves
// of identical HRUs run the whole step 36 % faster than waves of neighbouring cells (tools/exp/homogeneous.py).
#ifndef PROFILE_NBUCKET
#define PROFILE_NBUCKET (2 * (VIC_MAX_NODES + 2))
#endif
constexpr int NBUCKET = PROFILE_NBUCKET;

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
  int Nn, NOFLUX, EXP_TRANS, TFALLBACK;
};

// One solution record: T[Nn], {fbmask | ok << 32}, int fallback counts [Nn].  Every HRU keeps the records of its last
// two solves together with the trial temperatures they belong to: the Brent iteration on Tsurf ends with one more
// evaluation AT the root, which is one of the last two trial points unless the solver fell back -- the same inputs give
// the same profile bit for bit, so that solve is looked up instead of repeated (vic_surf_eval).
__host__ __device__ inline int pout_stride(int Nn) { return Nn + 1 + (Nn + 1) / 2; }
__host__ __device__ inline int pout_hru_stride(int Nn) { return 2 * pout_stride(Nn) + 2; }
__host__ __device__ inline int pout_key(int Nn, int slot) { return 2 * pout_stride(Nn) + slot; }

#ifndef PROFILE_GATE_LANES
#define PROFILE_GATE_LANES 16
#endif
constexpr int PROFILE_GATE = PROFILE_GATE_LANES;
#ifndef LOCKSTEP_GATE_LANES
#define LOCKSTEP_GATE_LANES 16
#endif
constexpr int LOCKSTEP_GATE = LOCKSTEP_GATE_LANES;   // same for the lock-step kernel's refill of lanes whose item is through     // lanes that must be waiting before the write-back / fetch section runs

template <int NN>
__global__ __launch_bounds__(64) void vic_profile_solve(const PArgs a) {
  __shared__ double Tl[NN * 64];      // current iterate  [node][lane]
  __shared__ double T0l[NN * 64];     // previous step    [node][lane]
  __shared__ int bcount[NBUCKET];
  const int lane = threadIdx.x;
  for (int b = lane; b < NBUCKET; b += 64) bcount[b] = a.count[b];
  if (blockIdx.x == 0) {
    for (int b = lane; b < NBUCKET; b += 64) a.count_zero[b] = 0;
    if (lane == 0) *a.evalonly_zero = 0;
  }
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int b = 0; b < NBUCKET; b++) n += bcount[b];
  if ((int)blockIdx.x * 64 >= n) return;           // more waves than work: nothing to pull (the grid is sized from an upper bound)

  const int Nn = (NN == VIC_MAX_NODES) ? a.Nn : NN;
  const int jlast = a.NOFLUX ? Nn : Nn - 1;     // exclusive upper node of a sweep
  const int MAXIT = 1000;
  const double threshold = 1.e-2;
#define TL(j) Tl[(j) * 64 + lane]
#define T0L(j) T0l[(j) * 64 + lane]
// the solution record of the current solve (pointers are recomputed where needed: they would cost six registers)
#define REC() (a.pout + (size_t)hru * pout_hru_stride(Nn) + ps * pout_stride(Nn))
#define CNT(j) (reinterpret_cast<int*>(REC() + Nn + 1)[j])      // fallback counters live in the record (rarely touched)

  enum { NODE = 1, BRENT = 2, FINISH = 3, IDLE = 4 };   // FINISH: solve done (or nothing yet), waiting at the gate
  int mode = FINISH;
  int hru = -1, it = 1, j = 1;
  bool frozen_on = false, ok = true, converged = false;
  unsigned fbmask = 0;
  double maxdiff = threshold, oldT = 0;
  const double* __restrict__ blk = a.pin;
  int ps = 0;
  BrentLean br;
  SoilThermalEqn eq;
  br.phase = BrentLean::DONE;
#ifdef VIC_PROF
  long long tp_gate = 0, tp_node = 0, tp_eq = 0, tp_adv = 0, tp_done = 0, tp_last = (long long)__builtin_readcyclecounter();
#define TP(acc) do { const long long n_ = (long long)__builtin_readcyclecounter(); acc += n_ - tp_last; tp_last = n_; } while (0)
#else
#define TP(acc) do { } while (0)
#endif

  while (true) {
    TP(tp_done);
    // ---- gate: write-back of finished solves + fetch of new items (wave-uniform branch)
    const unsigned long long waiting = __ballot(mode == FINISH);
    const unsigned long long running = __ballot(mode == NODE || mode == BRENT);
    if (waiting == 0 && running == 0) break;             // every lane is IDLE
    if (waiting != 0 && (__popcll(waiting) >= PROFILE_GATE || running == 0)) {
      if (mode == FINISH) {
        if (hru >= 0) {
          if (ok && a.TFALLBACK) {      // cold-nose hack, frozen_soil.c:470-484 (sic: Tlast[j+1] - T[j]); Tlast == T0
#pragma unroll 1
            for (int k = 1; k < Nn - 1; k++) {
              const double Tk = TL(k), Tm = TL(k - 1), Tp = TL(k + 1), Lk = T0L(k), Lm = T0L(k - 1), Lp = T0L(k + 1);
              if (Lm - Lk > 0 && Lp - Tk > 0 && (Tm - Tk) - (Lm - Lk) > 0 && (Tp - Tk) - (Lp - Lk) > 0) {
                TL(k) = 0.5 * (Tm + Tp);
                fbmask |= (1u << k);
                CNT(k) += 1;
              }
            }
          }
          if (ok && !converged) {
            if (a.TFALLBACK) {
#pragma unroll 1
              for (int k = 0; k < Nn; k++) { TL(k) = T0L(k); CNT(k) += 1; }
              fbmask |= (Nn >= 32) ? 0xFFFFFFFFu : ((1u << Nn) - 1u);
            } else ok = false;
          }
          double* __restrict__ out = REC();
#pragma unroll
          for (int k = 0; k < NN; k++)
            if (k < Nn) out[k] = TL(k);
          out[Nn] = __longlong_as_double((long long)((unsigned long long)fbmask | ((unsigned long long)(ok ? 1 : 0) << 32)));
          a.pout[(size_t)hru * pout_hru_stride(Nn) + pout_key(Nn, ps)] = T0L(0);   // the trial surface temperature of this record
        }
        // next item: one atomic for all waiting lanes
        const int leader = __ffsll((long long)waiting) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(a.next, __popcll(waiting));
        base = __builtin_amdgcn_readlane(base, leader);
        const int slot = base + __popcll(waiting & ((1ull << lane) - 1ull));
        if (slot < n) {
          {
            int rem = slot, found = 0;              // segment of this slot, highest key (most work) first
#pragma unroll 1
            for (int b = NBUCKET - 1; b >= 0; b--) {
              const int cb = bcount[b];
              if (rem < cb) { found = b * a.cap + rem; break; }
              rem -= cb;
            }
            hru = a.list[found];
          }
          blk = a.pin + (size_t)hru * Nn * PREC;
          ps = a.pslot[hru];
          frozen_on = blk[PR_A] != 0.0;
          const double Ts = a.ts[hru];
#pragma unroll
          for (int k = 0; k < NN; k++)
            if (k < Nn) { const double t = (k == 0) ? Ts : blk[k * PREC + PR_T0]; T0L(k) = t; TL(k) = t; CNT(k) = 0; }
          fbmask = 0; ok = true; it = 1; j = 1; maxdiff = threshold;
          converged = (jlast <= 1);
          mode = converged ? FINISH : NODE;
        } else { hru = -1; mode = IDLE; }
      }
    }
    TP(tp_gate);
    // ---- one unit of work per lane
    bool node_done = false;
    double newT = 0;
    if (mode == NODE) {
      oldT = TL(j);
      const bool bottom = (j == Nn - 1);        // only reached with NOFLUX (frozen_soil.c:423-464)
      const double Tdn = bottom ? oldT : TL(j + 1), Tup = TL(j - 1);
      const double* __restrict__ r = blk + j * PREC;
      const double A = r[PR_A], B = r[PR_B], C = r[PR_C], D = r[PR_D], T0j = T0L(j);
      if (oldT >= 0 || !frozen_on) {
        const double EI = r[PR_EI];
        if (!a.EXP_TRANS) newT = (A * T0j + B * (Tdn - Tup) + C * Tdn + D * Tup + EI) / (A + C + D);
        else newT = (A * T0j + B * (Tdn - Tup) + C * (Tdn + Tup) - D * (Tdn - Tup) + EI) / (A + 2. * C);
        node_done = true;
      } else {
        eq.TL = Tdn; eq.TU = Tup; eq.T0 = T0j; eq.moist = r[PR_MOIST]; eq.ice0 = r[PR_ICE];
        eq.A = A; eq.C = C; eq.D = D; eq.E = r[PR_E];
        eq.max_moist = r[PR_MAXM];
        eq.prepare(B, r[PR_CURVE_DIV], r[PR_CURVE_EXP], j);
        br.start(T0j - SOIL_DT, T0j + SOIL_DT);
        mode = BRENT;
      }
    }
    TP(tp_node);
    if (mode == BRENT) {
      const double fx = eq.eval(br.x, a.EXP_TRANS != 0);
      TP(tp_eq);
      br.advance(fx);
      if (br.finished()) {
        double rt = br.b;
        if (br.phase == BrentLean::FAILED) {
          if (a.TFALLBACK) { rt = eq.T0; fbmask |= (1u << j); CNT(j) += 1; }
          else { ok = false; mode = FINISH; }
        }
        if (mode == BRENT) { newT = rt; node_done = true; mode = NODE; }
      }
    }
    TP(tp_adv);
    if (node_done) {
      TL(j) = newT;
      const double diff = fabs(oldT - newT);
      if (diff > maxdiff) maxdiff = diff;
      j++;
      if (j >= jlast) {                           // end of a Gauss-Seidel sweep (frozen_soil.c:466)
        if (maxdiff <= threshold) { converged = true; mode = FINISH; }
        else if (it >= MAXIT) mode = FINISH;
        else { it++; j = 1; maxdiff = threshold; }
      }
    }
  }
#ifdef VIC_PROF
  if (lane == 0) {
    atomicAdd(&vic_prof_cyc[20], (unsigned long long)tp_gate); atomicAdd(&vic_prof_cyc[21], (unsigned long long)tp_node);
    atomicAdd(&vic_prof_cyc[22], (unsigned long long)tp_eq); atomicAdd(&vic_prof_cyc[23], (unsigned long long)tp_adv);
    atomicAdd(&vic_prof_cyc[24], (unsigned long long)tp_done);
  }
#endif
#undef TP
#undef TL
#undef T0L
#undef CNT
#undef REC
}

// ------------------------------------------------------------------------------------------------
// Lock-step variant: with the work lists keyed by the frozen-node count, the 64 HRUs a wave holds have their Brent solves
// at the same nodes, so the reference's loop nest can run as written -- sweeps { nodes { Brent } } -- with the wave
// iterating each node's Brent until its slowest lane is through.  Node columns live in LDS, there is no per-lane mode
// inside a sweep; a lane whose Gauss-Seidel iteration has ended writes its record and takes the next item at the gate
// before the following sweep, so the sweep count is per lane (91 % of lanes busy per sweep) while every node visit still
// costs the slowest lane's Brent (12.8 wave iterations for 8.2 per lane on cfg3: the 53 % lane utilisation that is left).
// ------------------------------------------------------------------------------------------------
// tools/hostemu with -DVIC_HOSTEMU_HIST: histogram of Brent evaluations per node solve, split by whether the bracket
// T0 +- SOIL_DT contains 0 C (how the work-list key of vic_fd_stage was chosen)
#ifdef VIC_HOSTEMU_HIST
static long vic_hist[64];
static long vic_hist2[64];
static void vic_hist_print() { for (int i = 0; i < 32; i++) printf("nit %d far %ld near %ld\n", i, vic_hist[i], vic_hist2[i]); }
static void vic_hist_add(int n, bool near0) { static bool reg = (atexit(vic_hist_print), true); (void)reg; (near0 ? vic_hist2 : vic_hist)[n < 63 ? n : 63]++; }
#endif
#ifndef LS_WAVES
#define LS_WAVES 3      // 136 VGPRs; 4 waves/SIMD (128 VGPRs, 44 B scratch, LDS then allows 15 waves per CU) measured 1.5 % slower
#endif
template <int NN>
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
  // Per-lane item state.  A lane keeps its item across sweeps; when the item's Gauss-Seidel iteration ends the lane writes
  // the record back and waits at the gate for the next item, so a wave is not held to its slowest HRU's sweep count.
  bool have = false, sweeping = false, converged = false, ok = true, frozen_on = false;
  int hru = 0, ps = 0, it = 1;
  unsigned fbmask = 0;
  const double* __restrict__ blk = a.pin;
#ifdef VIC_HOSTEMU_HIST
  int hist_nit = 0;
#endif
  bool more = true;                                  // wave-uniform: the work list has items nobody has taken yet
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
        int rem = slot, found = 0;
#pragma unroll 1
        for (int b = NBUCKET - 1; b >= 0; b--) {
          const int cb = bcount[b];
          if (rem < cb) { found = b * a.cap + rem; break; }
          rem -= cb;
        }
        hru = a.list[found];
        blk = a.pin + (size_t)hru * Nn * PREC;
        ps = a.pslot[hru];
        frozen_on = blk[PR_A] != 0.0;
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
          bool fz = false;
          double oldT = 0, newT = 0;
          BrentLean br;
          SoilThermalEqn eq;
          br.phase = BrentLean::DONE;
          if (sweeping) {
            oldT = T(j);
            const double Tdn = (j == Nn - 1) ? oldT : T((j + 1 < NN) ? j + 1 : j), Tup = T(j - 1);
            const double* __restrict__ r = blk + j * PREC;
            const double A = r[PR_A], B = r[PR_B], C = r[PR_C], D = r[PR_D], T0j = T0(j);
            if (oldT >= 0 || !frozen_on) {
              const double EI = r[PR_EI];
              if (!EXP_TRANS) newT = (A * T0j + B * (Tdn - Tup) + C * Tdn + D * Tup + EI) / (A + C + D);
              else newT = (A * T0j + B * (Tdn - Tup) + C * (Tdn + Tup) - D * (Tdn - Tup) + EI) / (A + 2. * C);
            } else {
              eq.TL = Tdn; eq.TU = Tup; eq.T0 = T0j; eq.moist = r[PR_MOIST]; eq.ice0 = r[PR_ICE];
              eq.A = A; eq.C = C; eq.D = D; eq.E = r[PR_E];
              eq.max_moist = r[PR_MAXM];
              eq.prepare(B, r[PR_CURVE_DIV], r[PR_CURVE_EXP], j);
              br.start(T0j - SOIL_DT, T0j + SOIL_DT);
              fz = true;
            }
          }
          PROF_WAVE(18); PROF_VOTE(19, fz); if (__any(fz)) PROF_WAVE(22);
          while (__any(fz && !br.finished())) {
            PROF_WAVE(20); PROF_VOTE(21, fz && !br.finished());
#ifdef VIC_HOSTEMU_HIST
            if (fz && !br.finished()) hist_nit++;
#endif
            if (fz && !br.finished()) {
              const double fx = eq.eval(br.x, EXP_TRANS);
              br.advance(fx);
            }
          }
#ifdef VIC_HOSTEMU_HIST
          if (fz) { vic_hist_add(hist_nit, fabs(eq.T0) < SOIL_DT); hist_nit = 0; }
#endif
          if (sweeping) {
            if (fz) {
              newT = br.b;
              if (br.phase == BrentLean::FAILED) {
                if (a.TFALLBACK) { newT = eq.T0; fbmask |= (1u << j); LS_CNT(j) += 1; }
                else { ok = false; sweeping = false; }
              }
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
      if (ok && a.TFALLBACK) {      // cold-nose hack, frozen_soil.c:470-484 (sic: Tlast[j+1] - T(j)); Tlast == T0
#pragma unroll
        for (int k = 1; k < NN - 1; k++) {
          if (k < Nn - 1) {
            const double Tk = T(k), Tm = T(k - 1), Tp = T(k + 1), Lk = T0(k), Lm = T0(k - 1), Lp = T0(k + 1);
            if (Lm - Lk > 0 && Lp - Tk > 0 && (Tm - Tk) - (Lm - Lk) > 0 && (Tp - Tk) - (Lp - Lk) > 0) {
              T(k) = 0.5 * (Tm + Tp);
              fbmask |= (1u << k);
              LS_CNT(k) += 1;
            }
          }
        }
      }
      if (ok && !converged) {
        if (a.TFALLBACK) {
#pragma unroll
          for (int k = 0; k < NN; k++)
            if (k < Nn) { T(k) = T0(k); LS_CNT(k) += 1; }
          fbmask |= (Nn >= 32) ? 0xFFFFFFFFu : ((1u << Nn) - 1u);
        } else ok = false;
      }
      double* __restrict__ rec = LS_REC();
#pragma unroll
      for (int k = 0; k < NN; k++)
        if (k < Nn) rec[k] = T(k);
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
