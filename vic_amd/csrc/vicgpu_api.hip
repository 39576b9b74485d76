// vicgpu_api.hip — C-ABI of the MI355X VIC hot path (include/vicgpu.h) and its kernels.  gfx950 only.
//
// Kernels
//   vic_hru_step<NN>   one lane per HRU: the per-HRU body of full_energy (full_energy.c:216-456) = aerodynamics,
//                      prepare_full_energy, surface_fluxes (snow, ground energy balance, pot. evap), runoff.
//                      HBM-side it is a streaming read-modify-write of the SoA state table; all physics is fp64 VALU.
//   vic_fd_stage<NN>,  the same step for the finite-difference soil profile (FROZEN_SOIL / QUICK_FLUX off), cut at the
//   vic_profile_solve_*,   ground-surface root finder into a pipeline: stage kernel (everything around the root finder,
//   vic_surf_eval      context parked in HBM) -> rounds of { profile solves on a compacted work list ; residual +
//                      Brent step on Tsurf } -> stage kernel.  See vic_profile.hpp for why.
//   vic_cell_reduce    one lane per cell: atmos->out_prec/out_rain/out_snow (full_energy.c:429-431) summed in hruList
//                      order (deterministic, no atomics) and the Cv-weighted per-cell accumulators.
//   vic_put_sum/_finish/_aggregate   put_data (put_data.c:7-760), the aggregated output variables (vic_putdata.hpp)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "vicgpu.h"
#include <cstddef>
#include <type_traits>
#include "vic_glacier.hpp"
#include "vic_profile.hpp"
#include "vic_putdata.hpp"

using namespace vic;

#define HIPIGN(call) do { hipError_t ign_ = (call); (void)ign_; } while (0)
#define HIPCHK(ctx, call)                                                                              \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                  \
      return VICGPU_ERR_HIP;                                                                           \
    }                                                                                                  \
  } while (0)

// Host <-> device copies and fills of the set-up and read-back calls go through the context's own (non-blocking) stream
// and are waited for there: a copy on the null stream is not ordered against kernels on a non-blocking stream, and a
// pageable host-to-device copy may return before its last bytes have landed in device memory.
static hipError_t copy_on(hipStream_t st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e;
}
static hipError_t fill_on(hipStream_t st, void* dst, int value, size_t bytes) {
  hipError_t e = hipMemsetAsync(dst, value, bytes, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e;
}

// hstate[hru]: bits 0-1 state (0 idle, 1 evaluation pending, 2 root found: the stage kernel's turn), from bit 2 the EBG_* class
// of the root find.  (Stage, profile record and forcing sub-step packed into the same word, so that every load of the
// evaluation kernel can issue behind this one, were measured: 26.9 / 27.2 vs 27.0 / 27.2 ms per step -- nothing; removed.)
constexpr int HS_STATE = 3, HS_CLS_SHIFT = 2;
// XCD-aware launch order.  HRUs are numbered slot-major (hru = slot * ncell + cell), so the 64 HRUs of a block are 64
// consecutive cells of one (tile, band) slot, and the ~300 cell-parameter rows and the forcing rows of those cells are read
// again by the block of every other slot.  Workgroups go round-robin over the 8 XCDs and every XCD has its own L2: in launch
// order "all cells of slot 0, then slot 1 ..." those re-reads are a whole domain apart and come from HBM every time.  With
// map_nslot > 0 a launch covers a REGULAR list (nslot slots x ccount cells, entry = slot * ccount + cell) and block b takes
// cell block (b >> 3) / nslot * 8 + (b & 7) of slot (b >> 3) % nslot: the blocks of one cell block's slots are consecutive on
// ONE XCD, so its table rows are fetched from HBM once and hit in that XCD's L2 for the other slots.
struct LaunchMap {
  int nslot = 0, ccount = 0;         // nslot == 0: identity (irregular lists)
  __host__ __device__ int nblocks(int gcount) const {
    if (nslot == 0) return (gcount + 63) / 64;
    const int ncb = (ccount + 63) / 64;
    return (ncb + 7) / 8 * 8 * nslot;
  }
  // list index of (block, lane), or -1
  VIC_DEV int index(int block, int lane, int gcount) const {
    if (nslot == 0) { const int gi = block * 64 + lane; return gi < gcount ? gi : -1; }
    const int q = block >> 3, cb = q / nslot * 8 + (block & 7), cell = cb * 64 + lane;
    return cell < ccount ? (q % nslot) * ccount + cell : -1;
  }
};

struct KArgs {
  Opt o;
  LaunchMap map;
  int ncell, nhru, nveg_rows, write_fluxes;
  const double* veglib;
  const double* cell_params;
  const int* hpi;
  const double* hpd;
  const double* forcing;            // this step: [VIC_NFORCE][NF+1][ncell]
  const unsigned char* snowflag;    // this step: [NF+1][ncell]
  Dmy dmy;
  double* sd;
  int* si;
  double* flux;
  int* hru_err;                     // [nhru]
  const int* glist;                 // HRUs of this launch (a cell chunk), or null: all HRUs in order
  int gcount;
  // finite-difference pipeline only (null otherwise)
  unsigned long long* ctx;          // parked per-HRU context, [hru / 64][word][hru % 64]
  double* pin;                      // profile item blocks [nhru][Nn][PREC]
  double* ts;                       // trial surface temperature [nhru]
  double* pout;                     // profile solutions [nhru][pout_hru_stride(Nn)] (two records + their keys)
  int* pslot;                       // [nhru] record the next profile solve writes
  int* hstate;                      // [nhru] 0 idle, 1 residual evaluation pending, 2 root found: stage kernel's turn
  int* list;                        // work list the stage kernel appends to (NBUCKET segments of list_cap entries)
  int* count;                       // [NBUCKET]
  int list_cap;
  int* hkey;                        // [nhru] work-list segment of each HRU (number of frozen nodes)
  double* pimp;                     // IMPLICIT: the implicit solver's item blocks [nhru][Nn][PIMP]
  int* lastexp;                     // IMPLICIT: [nhru] record slot holding the flags of the root find's last explicit solve
  int* jl;                          // QUICK_SOLVE: [nhru] end of the column the profile kernel solves
  int phase;                        // 0: start of the step; p >= 1: after the root finder of sub-step p - 1
};

// ------------------------------------------------------------------------------------------------ parked context
// Plain structs are parked word by word.  The table is tiled by wave: one wave's whole context is a single contiguous slab
// (a handful of pages) instead of one row per word spread over the whole table, and inside the slab every lane owns runs of
// G consecutive words: [hru / 64][word / G][hru % 64][G].  G = 1 is the plain [word][lane] tiling (8-byte-per-lane rows run
// the load path at half its rate); G = 2 makes every access 16 bytes; G = 8 gives a lane whole 64-byte sectors, so a wave
// formed from the pending lists (sparse rounds: lane = pending HRU, 64 different slabs) wastes nothing of what it fetches,
// while a dense wave still reads its slab front to back (its 16-byte accesses, 64 bytes apart, fill the same lines over four
// instructions).  Measured, same box: evaluation kernel 6.8 vs 7.7-8.0 ms per step with the sparse rounds starting at 30 %
// pending instead of 4 %; the opening stage, which WRITES the context, 4.7-4.9 vs 4.3-4.5 ms.  So the slab has two regions:
// what the evaluation kernel reads (SurfSolve, SurfEBMut, SurfEBConst: words below CTX_NA) in groups of VIC_CTX_GROUP = 8, what
// only the two stage kernels exchange (everything after) in pairs.  (G = 16 and 32 measure like 8, G = 4 worse than 2.)
#ifndef VIC_CTX_AOS
#define VIC_CTX_AOS 0
#endif
#ifndef VIC_CTX_GROUP
#define VIC_CTX_GROUP 8
#endif
#ifndef VIC_CTX_GROUP_B
#define VIC_CTX_GROUP_B 2
#endif
// Word W of HRU g:  AOS    [hru][word]                        one HRU's context is one contiguous block (measured in round 2:
//                                                             sparse rounds -35 %, dense rounds +23 %)
//                   else   region A [hru / 64][W / G][hru % 64][G], then region B the same with G_B and W - CTX_NA
constexpr size_t CTX_NA = sizeof(SurfSolve) / 8 + sizeof(SurfEBMut) / 8 + offsetof(SurfEBConst, Cs2) / 8;
constexpr size_t CTX_NA_PAD = (CTX_NA + VIC_CTX_GROUP - 1) / VIC_CTX_GROUP * VIC_CTX_GROUP;
constexpr size_t ctx_padded_words(size_t words) {      // slab words per lane
  return CTX_NA_PAD + ((words > CTX_NA ? words - CTX_NA : 0) + VIC_CTX_GROUP_B - 1) / VIC_CTX_GROUP_B * VIC_CTX_GROUP_B;
}
struct CtxRef {
  unsigned long long* p;    // word 0 of this HRU (AOS) / of this HRU's wave slab
  int lane;
  VIC_DEV static CtxRef at(unsigned long long* base, size_t words_per_hru, size_t g) {
#if VIC_CTX_AOS
    return CtxRef{base + g * words_per_hru, 0};
#else
    return CtxRef{base + (g >> 6) * (ctx_padded_words(words_per_hru) * 64), (int)(g & 63)};
#endif
  }
  VIC_DEV unsigned long long* word(size_t W) const {
#if VIC_CTX_AOS
    return p + W;
#else
    if (W < CTX_NA) return p + (W / VIC_CTX_GROUP) * (64 * VIC_CTX_GROUP) + lane * VIC_CTX_GROUP + (W % VIC_CTX_GROUP);
    const size_t V = W - CTX_NA;
    return p + CTX_NA_PAD * 64 + (V / VIC_CTX_GROUP_B) * (64 * VIC_CTX_GROUP_B) + lane * VIC_CTX_GROUP_B + (V % VIC_CTX_GROUP_B);
#endif
  }
};
template <class T>
VIC_DEV void ctx_put(const CtxRef& r, size_t word0, const T& v) {
  static_assert(sizeof(T) % 8 == 0 && std::is_trivially_copyable<T>::value, "context structs are arrays of 8-byte words");
  constexpr int NW = sizeof(T) / 8;
  unsigned long long tmp[NW];
  __builtin_memcpy(tmp, &v, sizeof(T));
#pragma unroll
  for (int i = 0; i < NW; i++) *r.word(word0 + i) = tmp[i];
}
template <class T>
VIC_DEV void ctx_get(const CtxRef& r, size_t word0, T& v) {
  static_assert(sizeof(T) % 8 == 0 && std::is_trivially_copyable<T>::value, "context structs are arrays of 8-byte words");
  constexpr int NW = sizeof(T) / 8;
  unsigned long long tmp[NW];
#pragma unroll
  for (int i = 0; i < NW; i++) tmp[i] = *r.word(word0 + i);
  __builtin_memcpy(&v, tmp, sizeof(T));
}
// SurfEBConst / SurfEBMut are parked group by group (vic_surface.hpp): word ranges of the groups
constexpr int EBC_W_POST = offsetof(SurfEBConst, delta_t) / 8, EBC_W_ALWAYS = offsetof(SurfEBConst, ice0) / 8,
              EBC_W_FROZEN = offsetof(SurfEBConst, kappa_snow) / 8, EBC_W_SNOWCOV = offsetof(SurfEBConst, LongSnowIn) / 8,
              EBC_W_INCL = offsetof(SurfEBConst, lmoist) / 8, EBC_W_EVAP = offsetof(SurfEBConst, Wdew) / 8,
              EBC_W_CANOPY = offsetof(SurfEBConst, Cs2) / 8;
constexpr int EBM_W_FEED = offsetof(SurfEBMut, deltaCC) / 8, EBM_W_IN3 = offsetof(SurfEBMut, Tsnow_surf) / 8,
              EBM_W_TSNOW = offsetof(SurfEBMut, ra_used) / 8, EBM_W_RA1 = EBM_W_TSNOW + 1, EBM_W_VV = offsetof(SurfEBMut, vv) / 8,
              EBM_W_KEEP = offsetof(SurfEBMut, Tnew2) / 8;
static_assert(offsetof(SurfEBMut, fusion) / 8 == EBM_W_IN3 - 1 && offsetof(SurfEBMut, layerevap) / 8 == EBM_W_VV + 3, "SurfEBMut layout");
#ifndef VIC_FINAL_SHORTCUT
#define VIC_FINAL_SHORTCUT 1
#endif
constexpr size_t CW_SV = sizeof(SurfSolve) / 8, CW_EBM = sizeof(SurfEBMut) / 8, CW_EBC = EBC_W_CANOPY,      // Cs2 is never parked
                 CW_P = sizeof(SubStep) / 8, CW_L = sizeof(SubLoop) / 8, CW_C = sizeof(StepConst) / 8;
constexpr size_t CO_SV = 0, CO_EBM = CO_SV + CW_SV, CO_EBC = CO_EBM + CW_EBM, CO_P = CO_EBC + CW_EBC, CO_L = CO_P + CW_P,
                 CO_C = CO_L + CW_L, CO_W = CO_C + CW_C;
constexpr size_t CW_W = sizeof(WCarry) / 8, CO_WM = CO_W + CW_W;
template <int NN> constexpr size_t ctx_words() { return CO_WM + sizeof(WCarryMulti<NN>) / 8; }
static_assert(sizeof(StepConstPost) <= sizeof(StepConst), "StepConstPost is parked in StepConst's words");
static_assert(CTX_NA == CO_P, "region A of the context slab = what the evaluation kernel reads");
// SubLoop in two parts: the head always, the sub-step sums only once a sub-step has been booked (they are zero before)
constexpr size_t CW_L_HEAD = offsetof(SubLoop, st_AlbedoOver) / 8;
// SurfSolve: the Brent state and the abscissa (rewritten by every evaluation), then the rest
constexpr size_t CW_SV_ITER = offsetof(SurfSolve, Tsurf) / 8;

template <class T>
VIC_DEV void ctx_put_words(const CtxRef& r, size_t word0, const T& v, int first, int last) {
  constexpr int NW = sizeof(T) / 8;
#pragma unroll
  for (int i = 0; i < NW; i++)
    if (i >= first && i < last) {
      unsigned long long w;
      __builtin_memcpy(&w, reinterpret_cast<const char*>(&v) + 8 * i, 8);
      *r.word(word0 + i) = w;
    }
}
// word by word into the object (no whole-struct copy: the conditional group loads of the evaluation kernel must not make the
// struct an aggregate the optimiser keeps in memory)
template <class T>
VIC_DEV void ctx_get_words(const CtxRef& r, size_t word0, T& v, int first, int last) {
  constexpr int NW = sizeof(T) / 8;
#pragma unroll
  for (int i = 0; i < NW; i++)
    if (i >= first && i < last) {
      const unsigned long long w = *r.word(word0 + i);
      __builtin_memcpy(reinterpret_cast<char*>(&v) + 8 * i, &w, 8);
    }
}

// The residual's inputs, group by group: `cls` = EBG_* bits of the HRU's root find (which groups its evaluations use)
VIC_DEV int surf_eb_class(const SurfEBConst& c) {
#ifdef VIC_DEBUG_ALLGROUPS
  return VIC_DEBUG_ALLGROUPS;
#endif
  return (c.frozen_on ? EBG_FROZEN : 0) | ((c.snow_coverage > 0 && !c.INCLUDE_SNOW) ? EBG_SNOWCOV : 0) | (c.INCLUDE_SNOW ? EBG_INCL : 0)
         | (!c.SNOWING ? EBG_EVAP : 0) | ((c.VEG && !c.SNOWING) ? EBG_CANOPY : 0);
}
VIC_DEV void ebc_put(const CtxRef& cx, const SurfEBConst& c, int cls) {
  ctx_put_words(cx, CO_EBC, c, 0, EBC_W_ALWAYS);
  if (cls & EBG_FROZEN) ctx_put_words(cx, CO_EBC, c, EBC_W_ALWAYS, EBC_W_FROZEN);
  if (cls & EBG_SNOWCOV) ctx_put_words(cx, CO_EBC, c, EBC_W_FROZEN, EBC_W_SNOWCOV);
  if (cls & EBG_INCL) ctx_put_words(cx, CO_EBC, c, EBC_W_SNOWCOV, EBC_W_INCL);
  if (cls & EBG_EVAP) ctx_put_words(cx, CO_EBC, c, EBC_W_INCL, EBC_W_EVAP);
  if (cls & EBG_CANOPY) ctx_put_words(cx, CO_EBC, c, EBC_W_EVAP, EBC_W_CANOPY);
}
VIC_DEV void ebc_get(const CtxRef& cx, SurfEBConst& c, int cls) {
  ctx_get_words(cx, CO_EBC, c, 0, EBC_W_ALWAYS);
  if (cls & EBG_FROZEN) ctx_get_words(cx, CO_EBC, c, EBC_W_ALWAYS, EBC_W_FROZEN);
  if (cls & EBG_SNOWCOV) ctx_get_words(cx, CO_EBC, c, EBC_W_FROZEN, EBC_W_SNOWCOV);
  if (cls & EBG_INCL) ctx_get_words(cx, CO_EBC, c, EBC_W_SNOWCOV, EBC_W_INCL);
  if (cls & EBG_EVAP) ctx_get_words(cx, CO_EBC, c, EBC_W_INCL, EBC_W_EVAP);
  if (cls & EBG_CANOPY) ctx_get_words(cx, CO_EBC, c, EBC_W_EVAP, EBC_W_CANOPY);
}

// wave-aggregated append of this lane's HRU to segment `key` of a work list (order is irrelevant: HRUs never interact).
// One atomic per distinct key, all of them in flight together: every lane finds the lanes that share its key (one
// ballot per possible key), the first of each group reserves the group's entries.
VIC_DEV void list_append(int* __restrict__ list, int* count, int cap, bool pred, int key, int g) {
  if (__ballot(pred) == 0) return;
  unsigned long long mine = 0;
#pragma unroll 1
  for (int k = 0; k < NBUCKET; k++) {
    const unsigned long long m = __ballot(pred && key == k);
    if (key == k) mine = m;
  }
  const int lane = (int)__lane_id();
  const int rank = __popcll(mine & ((1ull << lane) - 1ull));
  int base = 0;
  if (pred && rank == 0) base = atomicAdd(count + key, __popcll(mine));
  base = __shfl(base, pred ? __ffsll((long long)mine) - 1 : lane);
  if (pred) list[(size_t)key * cap + base + rank] = g;
}

#include "vic_implicit.hpp"

// ------------------------------------------------------------------------------------------------ state table I/O
// node_props = false leaves the node moisture / ice / conductivity / heat-capacity rows for load_node_props
template <int NN>
VIC_DEV void load_state(const KArgs& a, int g, HruWork<NN>& w, bool node_props = true) {
  const int Nn = a.o.Nnode;
  const size_t nh = a.nhru;
  const double* __restrict__ sd = a.sd;
  const int* __restrict__ si = a.si;
#define SD(row) sd[(size_t)(row) * nh + g]
#define SI(row) si[(size_t)(row) * nh + g]
#pragma unroll
  for (int l = 0; l < 3; l++) { w.moist[l] = SD(SD_MOIST0 + l); w.ice[l] = SD(SD_ICE0 + l); w.layer_T[l] = SD(SD_LAYER_T0 + l); w.evap[l] = 0; }
  SoilEnergy& so = w.so; SnowEnergy& se = w.se; Snow& s = w.snow;
  so.snow_flux = SD(SD_SNOW_FLUX); so.grnd_flux = SD(SD_GRND_FLUX); so.deltaH = SD(SD_DELTAH); so.fusion = SD(SD_FUSION);
  so.LongUnderOut = SD(SD_LONGUNDEROUT); se.Tfoliage = SD(SD_TFOLIAGE);
  s.albedo = SD(SD_SNOW_ALBEDO); s.coldcontent = SD(SD_SNOW_COLDCONTENT); s.coverage = SD(SD_SNOW_COVERAGE);
  s.density = SD(SD_SNOW_DENSITY); s.depth = SD(SD_SNOW_DEPTH); s.pack_temp = SD(SD_SNOW_PACK_TEMP);
  s.pack_water = SD(SD_SNOW_PACK_WATER); s.snow_canopy = SD(SD_SNOW_CANOPY); s.surf_temp = SD(SD_SNOW_SURF_TEMP);
  s.surf_water = SD(SD_SNOW_SURF_WATER); s.swq = SD(SD_SNOW_SWQ); s.tmp_int_storage = SD(SD_SNOW_TMP_INT_STORAGE);
  s.store_swq = SD(SD_SNOW_STORE_SWQ); s.store_coverage = SD(SD_SNOW_STORE_COVERAGE); s.swq_slope = SD(SD_SNOW_SWQ_SLOPE);
  s.max_swq = SD(SD_SNOW_MAX_SWQ);
  s.blowing_flux = 0; s.canopy_vapor_flux = 0; s.mass_error = 0; s.melt = 0; s.Qnet = 0; s.surface_flux = 0; s.vapor_flux = 0;
  w.vv.Wdew = SD(SD_WDEW); w.vv.canopyevap = 0; w.vv.throughfall = 0;
  w.Tcanopy = SD(SD_TCANOPY); so.Tsurf = SD(SD_TSURF); se.AlbedoOver = SD(SD_ALBEDO_OVER); so.AlbedoUnder = SD(SD_ALBEDO_UNDER);
  se.canopy_advection = SD(SD_CANOPY_ADVECTION); se.canopy_latent = SD(SD_CANOPY_LATENT);
  se.canopy_latent_sub = SD(SD_CANOPY_LATENT_SUB); se.canopy_sensible = SD(SD_CANOPY_SENSIBLE);
  se.canopy_refreeze = SD(SD_CANOPY_REFREEZE);
  se.advected_sensible = so.advected_sensible = SD(SD_ADVECTED_SENSIBLE);
  se.advection = so.advection = SD(SD_ADVECTION);
  se.deltaCC = so.deltaCC = SD(SD_DELTACC);
  se.refreeze_energy = so.refreeze_energy = SD(SD_REFREEZE_ENERGY);
  so.melt_energy = SD(SD_MELT_ENERGY);
  se.error = so.error = SD(SD_ERROR);
  se.latent = so.latent = SD(SD_LATENT); se.latent_sub = so.latent_sub = SD(SD_LATENT_SUB);
  se.sensible = so.sensible = SD(SD_SENSIBLE);
  se.snow_flux = so.snow_flux;
  se.LongOverIn = SD(SD_LONGOVERIN); se.NetLongOver = SD(SD_NETLONGOVER); se.NetShortOver = SD(SD_NETSHORTOVER);
  se.ShortOverIn = SD(SD_SHORTOVERIN);
  so.NetShortGrnd = 0; so.NetLongUnder = SD(SD_NETLONGUNDER); so.NetShortUnder = 0;
  w.gl.surf_temp = SD(SD_GLAC_SURF_TEMP); w.gl.water_storage = SD(SD_GLAC_WATER_STORAGE);
  w.gl.cum_mass_balance = SD(SD_GLAC_CUM_MASS_BALANCE);
  w.gl.cold_content = NAN; w.gl.Qnet = NAN; w.gl.mass_balance = NAN; w.gl.ice_mass_balance = 0; w.gl.accumulation = NAN;
  w.gl.melt = NAN; w.gl.vapor_flux = NAN; w.gl.outflow = NAN; w.gl.outflow_coef = NAN; w.gl.inflow = NAN;
  w.deltaCC_glac = 0; w.glacier_flux = 0; w.glacier_melt_energy = 0;
  so.kappa[0] = so.kappa[1] = so.Cs[0] = so.Cs[1] = 0;
#pragma unroll
  for (int f = 0; f < 3; f++) { so.fdepth[f] = 0; so.tdepth[f] = 0; }
#pragma unroll
  for (int n = 0; n < NN; n++) {
    if (n < Nn) {
      w.nd.T[n] = SD(VICGPU_SD_NODE(SDN_T, n, Nn));
      if (node_props) {
        w.nd.moist[n] = SD(VICGPU_SD_NODE(SDN_MOIST, n, Nn)); w.nd.ice[n] = SD(VICGPU_SD_NODE(SDN_ICE, n, Nn));
        w.nd.kappa[n] = SD(VICGPU_SD_NODE(SDN_KAPPA, n, Nn)); w.nd.Cs[n] = SD(VICGPU_SD_NODE(SDN_CS, n, Nn));
      } else { w.nd.moist[n] = 0; w.nd.ice[n] = 0; w.nd.kappa[n] = 0; w.nd.Cs[n] = 0; }
      w.nd.fbflag[n] = SI(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)); w.nd.fbcount[n] = SI(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn));
    } else {
      w.nd.T[n] = 0; w.nd.moist[n] = 0; w.nd.ice[n] = 0; w.nd.kappa[n] = 0; w.nd.Cs[n] = 0; w.nd.fbflag[n] = 0; w.nd.fbcount[n] = 0;
    }
  }
  s.last_snow = SI(SI_SNOW_LAST_SNOW); s.MELTING = SI(SI_SNOW_MELTING); s.snow = SI(SI_SNOW_SNOW); s.store_snow = SI(SI_SNOW_STORE_SNOW);
  s.surf_temp_fbcount = SI(SI_SNOW_SURF_TEMP_FBCOUNT); s.surf_temp_fbflag = SI(SI_SNOW_SURF_TEMP_FBFLAG);
  so.Tsurf_fbcount = SI(SI_TSURF_FBCOUNT); so.Tsurf_fbflag = SI(SI_TSURF_FBFLAG);
  se.Tfoliage_fbcount = SI(SI_TFOLIAGE_FBCOUNT); se.Tfoliage_fbflag = SI(SI_TFOLIAGE_FBFLAG);
  so.frozen = SI(SI_FROZEN); so.Nfrost = SI(SI_NFROST); so.Nthaw = SI(SI_NTHAW);
  w.gl.surf_temp_fbcount = SI(SI_GLAC_SURF_TEMP_FBCOUNT); w.gl.surf_temp_fbflag = SI(SI_GLAC_SURF_TEMP_FBFLAG);
#undef SD
#undef SI
}

// the node rows that do not change during a step (distribute_node_moisture_properties rewrites them at its end)
template <int NN>
VIC_DEV void load_node_props(const KArgs& a, int g, Nodes<NN>& nd) {
  const int Nn = a.o.Nnode;
  const size_t nh = a.nhru;
  const double* __restrict__ sd = a.sd;
#pragma unroll
  for (int n = 0; n < NN; n++) {
    if (n < Nn) {
      nd.moist[n] = sd[(size_t)VICGPU_SD_NODE(SDN_MOIST, n, Nn) * nh + g]; nd.ice[n] = sd[(size_t)VICGPU_SD_NODE(SDN_ICE, n, Nn) * nh + g];
      nd.kappa[n] = sd[(size_t)VICGPU_SD_NODE(SDN_KAPPA, n, Nn) * nh + g]; nd.Cs[n] = sd[(size_t)VICGPU_SD_NODE(SDN_CS, n, Nn) * nh + g];
    }
  }
}

// Phase p >= 1 of the stage kernel: the part of the HRU's working set that neither crosses the root finder in the parked
// context nor is assigned by the bookkeeping before it is read -- state the step has not touched yet, from the state table
// (see WCarry, vic_step.hpp); what the bookkeeping assigns starts as zero.
template <int NN>
VIC_DEV void load_untouched_state(const KArgs& a, int g, HruWork<NN>& w) {
  const int Nn = a.o.Nnode;
  const size_t nh = a.nhru;
  const double* __restrict__ sd = a.sd;
  const int* __restrict__ si = a.si;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    w.moist[l] = sd[(size_t)(SD_MOIST0 + l) * nh + g]; w.ice[l] = sd[(size_t)(SD_ICE0 + l) * nh + g];
    w.layer_T[l] = sd[(size_t)(SD_LAYER_T0 + l) * nh + g]; w.evap[l] = 0;
  }
  w.vv.Wdew = sd[(size_t)SD_WDEW * nh + g]; w.vv.canopyevap = 0; w.vv.throughfall = 0;
  SoilEnergy& so = w.so;
  so.deltaCC = 0; so.refreeze_energy = 0; so.deltaH = 0; so.fusion = 0; so.grnd_flux = 0; so.latent = 0; so.latent_sub = 0; so.sensible = 0;
  so.snow_flux = 0; so.error = 0; so.NetShortGrnd = 0; so.NetLongUnder = 0; so.NetShortUnder = 0; so.LongUnderOut = 0; so.AlbedoUnder = 0;
  so.melt_energy = 0; so.Tsurf = 0; so.kappa[0] = so.kappa[1] = so.Cs[0] = so.Cs[1] = 0;
#pragma unroll
  for (int f = 0; f < 3; f++) { so.fdepth[f] = 0; so.tdepth[f] = 0; }
  so.advected_sensible = sd[(size_t)SD_ADVECTED_SENSIBLE * nh + g];
  so.Tsurf_fbflag = 0; so.Tsurf_fbcount = si[(size_t)SI_TSURF_FBCOUNT * nh + g];
  so.frozen = 0; so.Nfrost = 0; so.Nthaw = si[(size_t)SI_NTHAW * nh + g];
  w.Tcanopy = 0;
  w.gl.surf_temp = sd[(size_t)SD_GLAC_SURF_TEMP * nh + g]; w.gl.water_storage = sd[(size_t)SD_GLAC_WATER_STORAGE * nh + g];
  w.gl.cum_mass_balance = sd[(size_t)SD_GLAC_CUM_MASS_BALANCE * nh + g];
  w.gl.cold_content = NAN; w.gl.Qnet = NAN; w.gl.mass_balance = NAN; w.gl.ice_mass_balance = 0; w.gl.accumulation = NAN;
  w.gl.melt = NAN; w.gl.vapor_flux = NAN; w.gl.outflow = NAN; w.gl.outflow_coef = NAN; w.gl.inflow = NAN;
  w.gl.surf_temp_fbcount = si[(size_t)SI_GLAC_SURF_TEMP_FBCOUNT * nh + g]; w.gl.surf_temp_fbflag = si[(size_t)SI_GLAC_SURF_TEMP_FBFLAG * nh + g];
  w.deltaCC_glac = 0; w.glacier_flux = 0; w.glacier_melt_energy = 0;
#pragma unroll
  for (int n = 0; n < NN; n++) {
    w.nd.T[n] = 0; w.nd.moist[n] = 0; w.nd.ice[n] = 0; w.nd.kappa[n] = 0; w.nd.Cs[n] = 0; w.nd.fbflag[n] = 0;
    w.nd.fbcount[n] = (n < Nn) ? si[(size_t)VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn) * nh + g] : 0;
  }
#pragma unroll
  for (int p = 0; p < NPET; p++) w.pot_evap[p] = 0;
}

template <int NN>
VIC_DEV void store_state(const KArgs& a, int g, const HruWork<NN>& w) {
  const int Nn = a.o.Nnode;
  const size_t nh = a.nhru;
  double* __restrict__ sd = a.sd;
  int* __restrict__ si = a.si;
#define SD(row) sd[(size_t)(row) * nh + g]
#define SI(row) si[(size_t)(row) * nh + g]
#pragma unroll
  for (int l = 0; l < 3; l++) { SD(SD_MOIST0 + l) = w.moist[l]; SD(SD_ICE0 + l) = w.ice[l]; SD(SD_LAYER_T0 + l) = w.layer_T[l]; }
  const SoilEnergy& so = w.so; const SnowEnergy& se = w.se; const Snow& s = w.snow;
  SD(SD_SNOW_FLUX) = so.snow_flux; SD(SD_GRND_FLUX) = so.grnd_flux; SD(SD_DELTAH) = so.deltaH; SD(SD_FUSION) = so.fusion;
  SD(SD_LONGUNDEROUT) = so.LongUnderOut; SD(SD_TFOLIAGE) = se.Tfoliage;
  SD(SD_SNOW_ALBEDO) = s.albedo; SD(SD_SNOW_COLDCONTENT) = s.coldcontent; SD(SD_SNOW_COVERAGE) = s.coverage;
  SD(SD_SNOW_DENSITY) = s.density; SD(SD_SNOW_DEPTH) = s.depth; SD(SD_SNOW_PACK_TEMP) = s.pack_temp;
  SD(SD_SNOW_PACK_WATER) = s.pack_water; SD(SD_SNOW_CANOPY) = s.snow_canopy; SD(SD_SNOW_SURF_TEMP) = s.surf_temp;
  SD(SD_SNOW_SURF_WATER) = s.surf_water; SD(SD_SNOW_SWQ) = s.swq; SD(SD_SNOW_TMP_INT_STORAGE) = s.tmp_int_storage;
  SD(SD_SNOW_STORE_SWQ) = s.store_swq; SD(SD_SNOW_STORE_COVERAGE) = s.store_coverage; SD(SD_SNOW_SWQ_SLOPE) = s.swq_slope;
  SD(SD_SNOW_MAX_SWQ) = s.max_swq; SD(SD_WDEW) = w.vv.Wdew;
  SD(SD_TCANOPY) = w.Tcanopy; SD(SD_TSURF) = so.Tsurf; SD(SD_ALBEDO_OVER) = w.AlbedoOver_avg; SD(SD_ALBEDO_UNDER) = so.AlbedoUnder;
  SD(SD_CANOPY_ADVECTION) = se.canopy_advection; SD(SD_CANOPY_LATENT) = se.canopy_latent;
  SD(SD_CANOPY_LATENT_SUB) = se.canopy_latent_sub; SD(SD_CANOPY_SENSIBLE) = se.canopy_sensible;
  SD(SD_CANOPY_REFREEZE) = se.canopy_refreeze; SD(SD_ADVECTED_SENSIBLE) = so.advected_sensible;
  SD(SD_ADVECTION) = so.advection; SD(SD_DELTACC) = so.deltaCC; SD(SD_REFREEZE_ENERGY) = so.refreeze_energy;
  SD(SD_MELT_ENERGY) = so.melt_energy; SD(SD_ERROR) = so.error; SD(SD_LATENT) = so.latent; SD(SD_LATENT_SUB) = so.latent_sub;
  SD(SD_SENSIBLE) = so.sensible; SD(SD_LONGOVERIN) = w.LongOverIn_avg; SD(SD_NETLONGOVER) = w.NetLongOver_avg;
  SD(SD_NETSHORTOVER) = w.NetShortOver_avg; SD(SD_SHORTOVERIN) = w.ShortOverIn_avg; SD(SD_NETLONGUNDER) = so.NetLongUnder;
  SD(SD_GLAC_SURF_TEMP) = w.gl.surf_temp; SD(SD_GLAC_WATER_STORAGE) = w.gl.water_storage;
  SD(SD_GLAC_CUM_MASS_BALANCE) = w.gl.cum_mass_balance;
#pragma unroll
  for (int n = 0; n < NN; n++) {
    if (n < Nn) {
      SD(VICGPU_SD_NODE(SDN_T, n, Nn)) = w.nd.T[n]; SD(VICGPU_SD_NODE(SDN_MOIST, n, Nn)) = w.nd.moist[n];
      SD(VICGPU_SD_NODE(SDN_ICE, n, Nn)) = w.nd.ice[n]; SD(VICGPU_SD_NODE(SDN_KAPPA, n, Nn)) = w.nd.kappa[n];
      SD(VICGPU_SD_NODE(SDN_CS, n, Nn)) = w.nd.Cs[n];
      SI(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)) = w.nd.fbflag[n]; SI(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn)) = w.nd.fbcount[n];
    }
  }
  SI(SI_SNOW_LAST_SNOW) = s.last_snow; SI(SI_SNOW_MELTING) = s.MELTING; SI(SI_SNOW_SNOW) = s.snow; SI(SI_SNOW_STORE_SNOW) = s.store_snow;
  SI(SI_SNOW_SURF_TEMP_FBCOUNT) = s.surf_temp_fbcount; SI(SI_SNOW_SURF_TEMP_FBFLAG) = s.surf_temp_fbflag;
  SI(SI_TSURF_FBCOUNT) = so.Tsurf_fbcount; SI(SI_TSURF_FBFLAG) = so.Tsurf_fbflag;
  SI(SI_TFOLIAGE_FBCOUNT) = se.Tfoliage_fbcount; SI(SI_TFOLIAGE_FBFLAG) = se.Tfoliage_fbflag;
  SI(SI_FROZEN) = so.frozen; SI(SI_NFROST) = so.Nfrost; SI(SI_NTHAW) = so.Nthaw;
  SI(SI_GLAC_SURF_TEMP_FBCOUNT) = w.gl.surf_temp_fbcount; SI(SI_GLAC_SURF_TEMP_FBFLAG) = w.gl.surf_temp_fbflag;
#undef SD
#undef SI
}

template <int NN>
VIC_DEV void store_flux(const KArgs& a, int g, const HruWork<NN>& w, bool glac) {
  const size_t nh = a.nhru;
  double* __restrict__ fx = a.flux;
#define FX(row) fx[(size_t)(row) * nh + g]
  // the three per-HRU precipitation terms are always written: vic_cell_reduce consumes them
  FX(FX_OUT_PREC) = w.out_prec; FX(FX_OUT_RAIN) = w.out_rain; FX(FX_OUT_SNOW) = w.out_snow;
  FX(FX_RUNOFF) = w.runoff; FX(FX_BASEFLOW) = w.baseflow;
  FX(FX_EVAP0) = w.evap[0]; FX(FX_EVAP1) = w.evap[1]; FX(FX_EVAP2) = w.evap[2];
  FX(FX_CANOPYEVAP) = w.vv.canopyevap; FX(FX_SNOW_VAPOR_FLUX) = w.snow.vapor_flux;
  FX(FX_SNOW_CANOPY_VAPOR_FLUX) = w.snow.canopy_vapor_flux; FX(FX_GLAC_MASS_BALANCE) = w.gl.mass_balance;
  if (!a.write_fluxes) return;
  FX(FX_ASAT) = w.asat; FX(FX_INFLOW) = w.inflow; FX(FX_THROUGHFALL) = w.vv.throughfall;
  FX(FX_SNOW_BLOWING_FLUX) = w.snow.blowing_flux; FX(FX_SNOW_SURFACE_FLUX) = w.snow.surface_flux; FX(FX_SNOW_MELT) = w.snow.melt;
  FX(FX_SNOW_MASS_ERROR) = w.snow.mass_error; FX(FX_SNOW_QNET) = w.snow.Qnet;
#pragma unroll
  for (int p = 0; p < NPET; p++) FX(FX_POT_EVAP0 + p) = w.pot_evap[p];
  FX(FX_AERO_RESIST_SURFACE) = w.aero_resist_surface; FX(FX_AERO_RESIST_OVERSTORY) = w.aero_resist_overstory;
  FX(FX_ROOTMOIST) = w.rootmoist; FX(FX_WETNESS) = w.wetness;
  FX(FX_ZWT) = w.zwt.zwt; FX(FX_ZWT2) = w.zwt.zwt2; FX(FX_ZWT3) = w.zwt.zwt3;
#pragma unroll
  for (int l = 0; l < 3; l++) FX(FX_ZWTL0 + l) = w.zwt.lz[l];
  // the frost / thaw fronts exist where find_0_degree_fronts ran (surface_fluxes.c, FROZEN_SOIL); glacier HRUs keep the
  // values initialize_model_state gave them (vicgpu_set_fluxes), as they do in the reference
  if (a.o.FROZEN_SOIL && !glac) {
#pragma unroll
    for (int l = 0; l < 3; l++) { FX(FX_FDEPTH0 + l) = w.so.fdepth[l]; FX(FX_TDEPTH0 + l) = w.so.tdepth[l]; }
  }
  FX(FX_ATMOS_LATENT) = w.AtmosLatent; FX(FX_ATMOS_LATENT_SUB) = w.AtmosLatentSub; FX(FX_ATMOS_SENSIBLE) = w.AtmosSensible;
  FX(FX_LONG_UNDER_IN) = w.LongUnderIn; FX(FX_NET_LONG_ATMOS) = w.NetLongAtmos; FX(FX_NET_LONG_UNDER) = w.so.NetLongUnder;
  FX(FX_NET_SHORT_ATMOS) = w.NetShortAtmos; FX(FX_NET_SHORT_GRND) = w.so.NetShortGrnd; FX(FX_NET_SHORT_UNDER) = w.so.NetShortUnder;
  FX(FX_SHORT_UNDER_IN) = w.ShortUnderIn_avg;
  FX(FX_GLAC_ICE_MASS_BALANCE) = w.gl.ice_mass_balance; FX(FX_GLAC_ACCUMULATION) = w.gl.accumulation;
  FX(FX_GLAC_MELT) = w.gl.melt; FX(FX_GLAC_VAPOR_FLUX) = w.gl.vapor_flux; FX(FX_GLAC_INFLOW) = w.gl.inflow;
  FX(FX_GLAC_OUTFLOW) = w.gl.outflow; FX(FX_GLAC_OUTFLOW_COEF) = w.gl.outflow_coef; FX(FX_GLAC_QNET) = w.gl.Qnet;
  FX(FX_GLAC_COLD_CONTENT) = w.gl.cold_content; FX(FX_GLACIER_FLUX) = w.glacier_flux; FX(FX_DELTACC_GLAC) = w.deltaCC_glac;
  FX(FX_GLACIER_MELT_ENERGY) = w.glacier_melt_energy;
#undef FX
}

// ------------------------------------------------------------------------------------------------ HRU kernels
struct HruId { int c, band, veg_idx; bool is_glacier, is_art_bare, run; };

VIC_DEV HruId hru_id(const KArgs& a, int g) {
  const size_t nh = a.nhru;
  HruId id;
  id.c = a.hpi[(size_t)HPI_CELL * nh + g];
  id.band = a.hpi[(size_t)HPI_BAND * nh + g];
  id.veg_idx = a.hpi[(size_t)HPI_VEG_INDEX * nh + g];
  id.is_glacier = a.hpi[(size_t)HPI_IS_GLACIER * nh + g] != 0;
  id.is_art_bare = a.hpi[(size_t)HPI_IS_ARTIFICIAL_BARE * nh + g] != 0;
  const double Cv = a.hpd[(size_t)HPD_CV * nh + g];
  // full_energy.c:220
  const bool active = (Cv > 0.0) || (id.is_glacier && a.o.GLACIER_DYNAMICS && Cv >= 0.0);
  const double area = a.cell_params[(size_t)VICGPU_CP_BAND(CPB_AREAFRACT, id.band, a.o.Nnode, a.o.Nband) * a.ncell + id.c];
  id.run = active && ((area > 0) || (id.is_glacier && a.o.GLACIER_DYNAMICS && area >= 0.0));
  return id;
}

VIC_DEV void store_zero_record(const KArgs& a, int g) {
  const size_t nh = a.nhru;
  double* fx = a.flux;
  fx[(size_t)FX_OUT_PREC * nh + g] = 0; fx[(size_t)FX_OUT_RAIN * nh + g] = 0; fx[(size_t)FX_OUT_SNOW * nh + g] = 0;
  fx[(size_t)FX_RUNOFF * nh + g] = 0; fx[(size_t)FX_BASEFLOW * nh + g] = 0;
  fx[(size_t)FX_EVAP0 * nh + g] = 0; fx[(size_t)FX_EVAP1 * nh + g] = 0; fx[(size_t)FX_EVAP2 * nh + g] = 0;
  fx[(size_t)FX_CANOPYEVAP * nh + g] = 0; fx[(size_t)FX_SNOW_VAPOR_FLUX * nh + g] = 0;
  fx[(size_t)FX_SNOW_CANOPY_VAPOR_FLUX * nh + g] = 0; fx[(size_t)FX_GLAC_MASS_BALANCE * nh + g] = 0;
  a.hru_err[g] = 0;
}

VIC_DEV Soil3 load_soil3(const CellView& cv) {
  Soil3 s3;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    s3.depth[l] = cv.lay(CPL_DEPTH, l); s3.max_moist[l] = cv.lay(CPL_MAX_MOIST, l); s3.Wcr[l] = cv.lay(CPL_WCR, l);
    s3.Wpwp[l] = cv.lay(CPL_WPWP, l); s3.resid_moist[l] = cv.lay(CPL_RESID_MOIST, l);
  }
  return s3;
}

// full_energy.c:216-354: state in, prepare_full_energy, aerodynamic resistances.  Returns the error bits.
template <int NN, bool GLAC>
VIC_DEV int hru_prologue(const KArgs& a, int g, const HruId& id, const CellView& cv, const VegLib& vl, const Forcing& fc, const Soil3& s3,
                         HruWork<NN>& w, StepConst& C, bool node_props = true) {
  const Opt& o = a.o;
  const size_t nh = a.nhru;
  const int month = a.dmy.month;
  const int veg_idx = id.veg_idx;
  int err = 0;
  C.veg_idx = veg_idx; C.band = id.band; C.is_art_bare = id.is_art_bare ? 1 : 0;
#pragma unroll
  for (int l = 0; l < 3; l++) C.root[l] = (double)(float)a.hpd[(size_t)(HPD_ROOT0 + l) * nh + g];
  if (o.BLOWING) {
    C.sigma_slope = (double)(float)a.hpd[(size_t)HPD_SIGMA_SLOPE * nh + g]; C.lag_one = (double)(float)a.hpd[(size_t)HPD_LAG_ONE * nh + g];
    C.fetch = (double)(float)a.hpd[(size_t)HPD_FETCH * nh + g];
  } else { C.sigma_slope = 0; C.lag_one = 0; C.fetch = 0; }
  load_state<NN>(a, g, w, node_props);
  w.snow.vapor_flux = 0.; w.snow.canopy_vapor_flux = 0.;                  // full_energy.c:261-262

  const double wind_h = vl.f(veg_idx, VL_WIND_H);
  const double lai_cur = vl.f(veg_idx, VL_LAI + month - 1);
  C.surf_atten = exp(-vl.f(veg_idx, VL_RAD_ATTEN) * lai_cur);   // full_energy.c:282

  // prepare_full_energy.c:8-94
  C.moist0 = w.moist[0] / (s3.depth[0] * 1000.); C.ice0 = 0.;
  if (o.FROZEN_SOIL && cv.s(CP_FS_ACTIVE) != 0.0) {
    const double tm = (w.nd.T[0] + w.nd.T[1]) / 2.;
    if (tm < 0.) {
      C.ice0 = C.moist0 - maximum_unfrozen_water(tm, s3.max_moist[0] / (s3.depth[0] * 1000.), cv.lay(CPL_BUBBLE, 0), cv.lay(CPL_EXPT, 0));
      if (C.ice0 < 0.) C.ice0 = 0.;
    }
  }
  top_layer_thermal_properties(cv, s3, w.moist, w.ice, w.so.kappa, w.so.Cs);
  C.bare_albedo = GLAC ? cv.s(CP_GLAC_ALBEDO) : vl.f(veg_idx, VL_ALBEDO + month - 1);

  // aerodynamic resistances for the 6 PET surfaces and the current vegetation (full_energy.c:302-354).  The loop is
  // kept rolled (7 x CalcAerodynamic); what it indexes by p lives in locals, not in C (see vsel()).
  Vc Ra, U, disp, zref, z0, ap[NPET];
#pragma unroll
  for (int k = 0; k < NCASE; k++) { disp.v[k] = NAN; zref.v[k] = NAN; z0.v[k] = NAN; U.v[k] = NAN; Ra.v[k] = NAN; }
#pragma unroll
  for (int q = 0; q < NPET; q++) {
#pragma unroll
    for (int k = 0; k < NCASE; k++) ap[q].v[k] = NAN;
  }
  bool overstory = false;
  const double rough = cv.s(CP_ROUGH), snow_rough = cv.s(CP_SNOW_ROUGH), wind = fc.v(VIC_F_WIND, o.NR);
#pragma unroll 1
  for (int p = 0; p < NPET + 1; p++) {
    const int pet_idx = (p < NPET_NON_NAT) ? o.nveg_types + p : veg_idx;
    if (pet_idx == o.GLACIER_ID) z0.v[SNOW_FREE] = cv.s(CP_GLAC_ROUGH);      // sic: library index compared with a class id
    else z0.v[SNOW_FREE] = vl.f(pet_idx, VL_ROUGHNESS + month - 1);
    disp.v[SNOW_FREE] = vl.f(pet_idx, VL_DISPLACEMENT + month - 1);
    overstory = vl.f(pet_idx, VL_OVERSTORY) != 0.0;
    if (p >= NPET_NON_NAT && z0.v[SNOW_FREE] == 0) z0.v[SNOW_FREE] = rough;
    const double height = calc_veg_height(disp.v[SNOW_FREE], lai_cur);
    if (disp.v[SNOW_FREE] < wind_h) zref.v[SNOW_FREE] = wind_h;
    else zref.v[SNOW_FREE] = disp.v[SNOW_FREE] + wind_h + z0.v[SNOW_FREE];
    const double wind_corr = log((zref.v[SNOW_FREE] - 0.) / rough) / log((o.wind_h - 0.) / rough);
    U.v[SNOW_FREE] = wind * wind_corr;
    U.v[CANOPY] = NAN; U.v[SNOW_COVERED] = NAN; U.v[GLACIER_SURF] = NAN;
#pragma unroll
    for (int k = 0; k < NCASE; k++) Ra.v[k] = NAN;
    if (!calc_aerodynamic(overstory, height, vl.f(pet_idx, VL_TRUNK_RATIO), snow_rough, rough, vl.f(pet_idx, VL_WIND_ATTEN), Ra, U,
                          disp, zref, z0))
      err |= VICGPU_CELLERR_AERO;
    // ap[p] = Ra without a run-time index: hipcc 7.2's alloca-to-vector promotion mis-generated the dynamically indexed
    // store of this 24-double array in several builds of vic_hru_step; a select per slot also keeps ap in
    // registers by construction
#pragma unroll
    for (int q = 0; q < NPET; q++) {
#pragma unroll
      for (int k = 0; k < NCASE; k++) ap[q].v[k] = (p == q) ? Ra.v[k] : ap[q].v[k];
    }
  }
#pragma unroll
  for (int p = 0; p < NPET; p++) C.aero_pet[p] = ap[p];
  C.Ra = Ra; C.U = U; C.disp = disp; C.zref = zref; C.z0 = z0;
  C.overstory = overstory ? 1 : 0;
  w.aero_resist_surface = Ra.v[SNOW_FREE];
  w.aero_resist_overstory = Ra.v[CANOPY];
#pragma unroll
  for (int p = 0; p < NPET; p++) w.pot_evap[p] = 0;
  return err;
}

// full_energy.c:437-455 and state / flux out
template <int NN>
VIC_DEV void hru_epilogue(const KArgs& a, int g, const CellView& cv, const Soil3& s3, const StepConst& C, HruWork<NN>& w, int err, bool glac) {
  // root zone moisture and wetness (full_energy.c:437-455)
  w.rootmoist = 0; w.wetness = 0;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    if (C.root[l] > 0) w.rootmoist += w.moist[l];
    w.wetness += (w.moist[l] - s3.Wpwp[l]) / (cv.lay(CPL_POROSITY, l) * s3.depth[l] * 1000 - s3.Wpwp[l]);
  }
  w.wetness /= 3;

  bool finite = true;
#pragma unroll
  for (int l = 0; l < 3; l++) finite = finite && isfinite(w.moist[l]);
  finite = finite && isfinite(w.nd.T[0]) && isfinite(w.snow.swq);
  if (!finite) err |= VICGPU_CELLERR_NAN;

  PROF_T0(t_store);
  store_state<NN>(a, g, w);
  store_flux<NN>(a, g, w, glac);
  a.hru_err[g] = err;
  PROF_ADD(8, t_store);
}

// The whole HRU step in one lane: glacier HRUs (GLAC) and QUICK_FLUX (no soil-profile solve)
template <int NN, bool GLAC>
__global__ __launch_bounds__(64) void vic_hru_step(const KArgs a) {
  const int gi = a.map.index(blockIdx.x, threadIdx.x, a.gcount);
  if (gi < 0) return;
  const int g = a.glist ? a.glist[gi] : gi;
  const Opt& o = a.o;
  const HruId id = hru_id(a, g);
  // two instantiations share this body: GLAC = false handles ordinary HRUs (and writes the zero record of inactive
  // ones), GLAC = true handles glacier HRUs; the domain numbering keeps either kind wave-uniform
  if (id.is_glacier != GLAC && id.run) return;
  if (!id.run) {
    if (!GLAC) store_zero_record(a, g);
    return;
  }
  PROF_T0(t_kernel);
  CellView cv{a.cell_params, a.ncell, id.c, o.Nnode, o.Nband};
  VegLib vl{a.veglib};
  Forcing fc{a.forcing, a.snowflag, a.ncell, id.c, o.NR + 1};
  const Soil3 s3 = load_soil3(cv);
  HruWork<NN> w;
  StepConst C;
  int err = hru_prologue<NN, GLAC>(a, g, id, cv, vl, fc, s3, w, C);
  PROF_ADD(1, t_kernel);

  if (!(err & VICGPU_CELLERR_AERO)) {
    bool ok;
    if constexpr (GLAC) {
      GlacEnergy ge;
      double nlu, nsu, sui;
      const double blow[4] = {C.sigma_slope, C.lag_one, C.fetch, (double)C.is_art_bare};
      ok = surface_fluxes_glac<NN>(o, cv, vl, s3, fc, a.dmy, C.veg_idx, C.band, C.bare_albedo, C.aero_pet, C.Ra, C.U, C.zref, C.z0, C.disp, blow, w, w.gl,
                                   w.so.NetLongUnder, ge, nlu, nsu, sui);
      // hru.energy = step_energy + step averages (surface_fluxes_glac.c:485-526)
      SoilEnergy& so = w.so; SnowEnergy& se = w.se;
      so.snow_flux = ge.snow_flux; so.grnd_flux = ge.grnd_flux; so.deltaH = 0; so.fusion = 0; so.LongUnderOut = ge.LongUnderOut;
      so.AlbedoUnder = ge.AlbedoUnder; so.advected_sensible = ge.advected_sensible; so.advection = ge.advection;
      so.deltaCC = ge.deltaCC; so.refreeze_energy = ge.refreeze_energy; so.error = ge.error; so.latent = ge.latent;
      so.latent_sub = ge.latent_sub; so.sensible = ge.sensible; so.NetLongUnder = nlu; so.NetShortUnder = nsu; so.NetShortGrnd = 0;
      se.canopy_advection = 0; se.canopy_latent = 0; se.canopy_latent_sub = 0; se.canopy_sensible = 0; se.canopy_refreeze = 0;
      w.AlbedoOver_avg = 0; w.LongOverIn_avg = 0; w.NetLongOver_avg = 0; w.NetShortOver_avg = 0; w.ShortOverIn_avg = 0;
      w.ShortUnderIn_avg = sui;
      w.deltaCC_glac = ge.deltaCC_glac; w.glacier_flux = ge.glacier_flux; w.glacier_melt_energy = ge.glacier_melt_energy;
      // accumulateGlacierMassBalance.c:13-67: the per-step += (the accumulation-window decision is driver state: the
      // host makes cum_mass_balance valid when the window opens)
      if (!isnan(w.gl.cum_mass_balance) && !isnan(w.gl.mass_balance)) w.gl.cum_mass_balance += w.gl.mass_balance;
    } else {
      ok = surface_fluxes<NN>(o, cv, vl, s3, fc, a.dmy, C, w);
    }
    if (!ok) err |= VICGPU_CELLERR_SOLVER;
  }
  hru_epilogue<NN>(a, g, cv, s3, C, w, err, GLAC);
  PROF_ADD(0, t_kernel);
  PROF_WAVE(0);
  PROF_LANE(1);
}

// Finite-difference pipeline, stage kernel: phase 0 starts the step of every ordinary HRU; phase p >= 1 resumes the
// HRUs whose ground-surface root of sub-step p - 1 has been found.  Either way an HRU leaves with its next sub-step
// set up and parked (appended to the work list) or with its step finished and stored.
// FIRST: the phase-0 instantiation; MULTI: the run has more than one snow sub-step per step (otherwise phase 1 never sets up
// another sub-step and that code is not instantiated).
template <int NN, bool FIRST, bool MULTI>
__global__ __launch_bounds__(64) VIC_WAVES_PER_EU(1, 1) void vic_fd_stage(const KArgs a) {
  const int gi = a.map.index(blockIdx.x, threadIdx.x, a.gcount);
  if (gi < 0) return;
  const int g = a.glist ? a.glist[gi] : gi;
  const Opt& o = a.o;
  const HruId id = hru_id(a, g);
  if (id.run && id.is_glacier) return;              // vic_hru_step<NN, true> owns glacier HRUs
  if (FIRST && !id.run) { store_zero_record(a, g); a.hstate[g] = 0; return; }
  if (!FIRST && (a.hstate[g] & HS_STATE) != 2) return;
  CellView cv{a.cell_params, a.ncell, id.c, o.Nnode, o.Nband};
  VegLib vl{a.veglib};
  Forcing fc{a.forcing, a.snowflag, a.ncell, id.c, o.NR + 1};
  const Soil3 s3 = load_soil3(cv);
  const int Nn = (NN == VIC_MAX_NODES) ? o.Nnode : NN;
  const CtxRef cx = CtxRef::at(a.ctx, ctx_words<NN>(), g);
  HruWork<NN> w;
  StepConst C;
  SubLoop L;
  int err = 0;
  bool more;
  PROF_T0(t_stage);
  if constexpr (FIRST) {
    err = hru_prologue<NN, false>(a, g, id, cv, vl, fc, s3, w, C, /*node_props=*/false);
    PROF_ADD(1, t_stage);
    more = !(err & VICGPU_CELLERR_AERO);
    if (more) sf_begin<NN>(o, fc, C, w, L);
  } else {
    SubStep P;
    SurfEB eb;
    SurfSolve sv;
    ctx_get_words(cx, CO_SV, sv, (int)CW_SV_ITER, (int)CW_SV);                     // the result of the root find, not the Brent state
    ctx_get(cx, CO_EBM, static_cast<SurfEBMut&>(eb));
    ctx_get_words(cx, CO_EBC, static_cast<SurfEBConst&>(eb), 0, EBC_W_POST);      // the bookkeeping reads the flags and T2 only
    surf_cell_fill(eb, cv, vl, s3, fc, eb.hidx, id.veg_idx, a.dmy.month);
    ctx_get(cx, CO_P, P);
    ctx_get_words(cx, CO_L, L, 0, (int)CW_L_HEAD);
    if (MULTI && L.N_steps > 0) ctx_get_words(cx, CO_L, L, (int)CW_L_HEAD, (int)CW_L);
    else zero_substep_sums(L);
    load_untouched_state<NN>(a, g, w);
    if constexpr (MULTI) {
      ctx_get(cx, CO_C, C);
      WCarryMulti<NN> km;
      ctx_get(cx, CO_WM, km);
      carry_in_multi<NN>(km, w);
    } else {
      // one sub-step per step: of StepConst the bookkeeping needs the PET resistances of this sub-step's surface cases (parked),
      // the rest is in the HRU tables
      StepConstPost q;
      ctx_get(cx, CO_C, q);
      step_const_post_in(q, P.UnderStory, C);
      C.veg_idx = id.veg_idx; C.band = id.band; C.is_art_bare = id.is_art_bare ? 1 : 0;
      C.overstory = (vl.f(id.veg_idx, VL_OVERSTORY) != 0.0) ? 1 : 0;
#pragma unroll
      for (int l = 0; l < 3; l++) C.root[l] = (double)(float)a.hpd[(size_t)(HPD_ROOT0 + l) * a.nhru + g];
    }
    {
      WCarry k;
      ctx_get(cx, CO_W, k);
      carry_in<NN>(k, w);
    }
    PROF_ADD(11, t_stage);
    PROF_T0(t_post);
    // the soil profile of the final evaluation
    const double* __restrict__ po = a.pout + (size_t)g * pout_hru_stride(Nn) + sv.final_slot * pout_stride(Nn);
    const int* __restrict__ poc = reinterpret_cast<const int*>(po + Nn + 1);
    double Tprof[NN];
    int cntprof[NN];
    const unsigned long long flags = (unsigned long long)__double_as_longlong(po[Nn]);
#pragma unroll
    for (int n = 0; n < NN; n++) {
      Tprof[n] = (n < Nn) ? po[n] : 0.0;
      cntprof[n] = (n < Nn) ? poc[n] : 0;
    }
    sf_sub_post<NN>(o, cv, vl, s3, fc, a.dmy, C, w, L, P, eb, sv, Tprof, cntprof, (unsigned)(flags & 0xFFFFFFFFull));
    PROF_ADD(12, t_post);
    more = true;
  }
  bool pend = false;
  int key = 0;
  if ((FIRST || MULTI) && more && L.hidx < L.endhidx) {
    if constexpr (FIRST || MULTI) {
      SubStep P;
      SurfEB eb;
      SurfSolve sv;
      PROF_T0(t_pre);
      sf_sub_pre<NN>(o, cv, vl, s3, fc, a.dmy, C, w, L, P, eb, sv);
      PROF_ADD(13, t_pre);
      PROF_T0(t_put);
      ctx_put(cx, CO_SV, sv);
      ctx_put_words(cx, CO_EBM, static_cast<const SurfEBMut&>(eb), 0, EBM_W_KEEP);   // the outputs are the final evaluation's to write
      const int cls = surf_eb_class(eb);
      ebc_put(cx, eb, cls);
      ctx_put(cx, CO_P, P);
      ctx_put_words(cx, CO_L, L, 0, (int)CW_L_HEAD);
      if (MULTI && L.N_steps > 0) ctx_put_words(cx, CO_L, L, (int)CW_L_HEAD, (int)CW_L);
      if constexpr (MULTI) {
        if (FIRST) ctx_put(cx, CO_C, C);
        WCarryMulti<NN> km;
        carry_out_multi<NN>(w, km);
        ctx_put(cx, CO_WM, km);
      } else {
        StepConstPost q;
        step_const_post_out(C, P.UnderStory, q);
        ctx_put(cx, CO_C, q);
      }
      {
        WCarry k;
        carry_out<NN>(w, k);
        ctx_put(cx, CO_W, k);
      }
      // the item block needs the node rows that nothing before it reads: loaded last, so that they are not live (and
      // spilled) across solve_snow
      load_node_props<NN>(a, g, w.nd);
      {
        // Key = number of frozen nodes, plus a second set of segments for HRUs with a node whose Brent bracket
        // T0 +- SOIL_DT contains 0 C: the residual has a kink there (ice vanishes), Brent degrades to bisection and needs
        // 17-28 evaluations instead of 6-11 -- 1.5 % of the node solves, but one such lane holds up its whole wave.
        // (Finer keys -- frozen range, thawed top -- and the measured trip count were tried: no better.)
        int nfrozen = 0;
        bool kink = false;
#pragma unroll
        for (int n = 1; n < NN; n++)
          if (n < Nn && eb.frozen_on) {
            if (w.nd.T[n] < 0) nfrozen++;
            if (fabs(w.nd.T[n]) < SOIL_DT) kink = true;
          }
        key = nfrozen + (kink ? NBUCKET / 2 : 0);
        a.hkey[g] = key;
      }
      profile_item_store<NN>(o, cv, s3, w.nd, eb.delta_t, eb.frozen_on != 0, a.pin + (size_t)g * Nn * PREC);
      if (o.IMPLICIT) {
        double* __restrict__ im = a.pimp + (size_t)g * Nn * PIMP;
#pragma unroll
        for (int n = 0; n < NN; n++)
          if (n < Nn) {
            im[n * PIMP + PI_MOIST] = w.nd.moist[n]; im[n * PIMP + PI_ICE] = (n == 0) ? eb.delta_t : w.nd.ice[n];
            im[n * PIMP + PI_KAPPA] = w.nd.kappa[n]; im[n * PIMP + PI_CS] = w.nd.Cs[n];
          }
        a.lastexp[g] = -1;
      }
      a.ts[g] = sv.x;
      a.pslot[g] = 0;
      if (o.QUICK_SOLVE) {
        // calc_surf_energy_bal.c:289-299: the iteration solves the nodes down to the thaw depth + 4 only
        int tmpNnodes = 0;
#pragma unroll
        for (int n = NN - 1; n >= 0; n--)
          if (n <= Nn - 5 && w.nd.T[n] >= 0 && w.nd.T[(n + 1 < NN) ? n + 1 : n] < 0) tmpNnodes = n + 1;
        if (tmpNnodes == 0) tmpNnodes = (w.nd.T[0] <= 0 && w.nd.T[1] >= 0) ? Nn : 3;
        else tmpNnodes += 4;
        // (the iteration runs with NOFLUX forced off, calc_surf_energy_bal.c:298; without an iteration -- no FULL_ENERGY -- the
        // run's own NOFLUX decides whether the bottom node is solved)
        a.jl[g] = (sv.stage == SurfSolve::ROOT_QUICK) ? tmpNnodes - 1 : (o.NOFLUX ? Nn : Nn - 1);
      }
      a.pout[(size_t)g * pout_hru_stride(Nn) + pout_key(Nn, 0)] = NAN;      // no solve on record yet
      a.pout[(size_t)g * pout_hru_stride(Nn) + pout_key(Nn, 1)] = NAN;
      a.hstate[g] = 1 | (cls << HS_CLS_SHIFT);
      pend = true;
      PROF_ADD(14, t_put);
    }
  } else {
    PROF_T0(t_end);
    if (more && !sf_end<NN>(o, cv, s3, C, w, L)) err |= VICGPU_CELLERR_SOLVER;
    PROF_ADD(15, t_end);
    hru_epilogue<NN>(a, g, cv, s3, C, w, err, false);
    a.hstate[g] = 0;
  }
  list_append(a.list, a.count, a.list_cap, pend, key, g);
  PROF_ADD(0, t_stage);
  PROF_WAVE(0);
  PROF_LANE(1);
}

// Finite-difference pipeline, evaluation kernel: the residual of the ground-surface energy balance at the trial
// temperature whose soil profile has just been solved, then one step of the Brent iteration on Tsurf.
struct EArgs {
  Opt o;
  LaunchMap map;
  int ncell, nhru, Nn;
  const int* glist;
  int gcount;
  const double* cell_params;
  const int* hpi;
  unsigned long long* ctx;
  size_t ctx_words;
  double* pout;
  int* pslot;
  double* ts;
  int* jl;               // QUICK_SOLVE: [nhru] end of the column the profile kernel solves (null otherwise)
  int* hstate;
  int* list_next;        // NBUCKET segments of list_cap entries
  int* count_next;       // [NBUCKET]
  int list_cap;
  const int* hkey;
  int* profile_next;     // work-list cursor of the profile kernel, cleared for its next launch
  int* evalonly;         // HRUs that wait for an evaluation without a solve (final evaluation on record)
  int* eo_list_next;     // ... and which: with the work list this is every HRU the round leaves pending
  // sparse rounds (at most list_thr HRUs pending; the others go through glist / map and test hstate): lane = pending HRU, taken
  // from the work list the profile kernel has just gone through and from the evaluation-only list of the round before
  int list_thr;
  const int* list_cur;
  const int* count_cur;  // [NBUCKET]
  const int* eo_list_cur;
  const int* npend_cur;  // their sum + the length of eo_list_cur, written by the round's profile kernel
  int implicit;          // IMPLICIT: the final evaluation is always solved again (its fallback flags depend on the solves before it)
  const double* veglib;  // for the table-derived part of the residual's inputs (surf_cell_fill)
  const double* forcing; // this step
  int month;
};

#ifndef VIC_EVAL_WAVES
#define VIC_EVAL_WAVES 2
#endif
__global__ __launch_bounds__(64) VIC_WAVES_PER_EU(VIC_EVAL_WAVES, VIC_EVAL_WAVES) void vic_surf_eval(const EArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.profile_next = 0;
  // Sparse rounds.  A dense launch pays a whole wave -- its chain of dependent loads -- for every 64 HRUs of which one is
  // pending.  The number pending is on the device before the host knows it (the work list the profile kernel has just gone
  // through + the evaluation-only list of the round before), so every wave looks at it: from the round in which at most
  // list_thr HRUs are pending, lane = entry of those lists and the waves beyond the lists' end leave at once.
  const int npend = *a.npend_cur;
  int g;
  if (npend <= a.list_thr) {
    if ((int)blockIdx.x * 64 >= npend) return;
    __shared__ int bcount[NBUCKET];
    for (int b = threadIdx.x; b < NBUCKET; b += 64) bcount[b] = a.count_cur[b];
    __syncthreads();
    int nsolve = 0;
#pragma unroll 1
    for (int b = 0; b < NBUCKET; b++) nsolve += bcount[b];
    const int gi = blockIdx.x * 64 + threadIdx.x;
    if (gi >= npend) return;
    if (gi < nsolve) {
      int rem = gi, found = 0;
#pragma unroll 1
      for (int b = 0; b < NBUCKET; b++) {
        const int cb = bcount[b];
        if (rem < cb) { found = b * a.list_cap + rem; break; }
        rem -= cb;
      }
      g = a.list_cur[found];
    } else g = a.eo_list_cur[gi - nsolve];
  } else {
    const int gi = a.map.index(blockIdx.x, threadIdx.x, a.gcount);
    if (gi < 0) return;
    g = a.glist ? a.glist[gi] : gi;
  }
  const int hs = a.hstate[g];
  if ((hs & HS_STATE) != 1) return;
  const int cls = hs >> HS_CLS_SHIFT;
  const size_t nh = a.nhru;
  const int c = a.hpi[(size_t)HPI_CELL * nh + g];
  const int veg_idx = a.hpi[(size_t)HPI_VEG_INDEX * nh + g];
  const int ps = a.pslot[g];
  CellView cv{a.cell_params, a.ncell, c, a.o.Nnode, a.o.Nband};
  const Soil3 s3 = load_soil3(cv);
  SurfSolve sv;
  SurfEB eb;
  const CtxRef cx = CtxRef::at(a.ctx, a.ctx_words, g);
  ctx_get(cx, CO_SV, sv);
  ebc_get(cx, eb, cls);
  const bool is_final = sv.stage == SurfSolve::FINAL;
  // the record the profile kernel has just written, or the one found on record for the final evaluation
  const int slot = sv.on_record ? sv.final_slot : ps;
  {
    // of SurfEBMut an evaluation of the iteration reads what it cannot know otherwise; the final one reads every input, since
    // what it does not assign passes through to the bookkeeping (vic_surface.hpp)
    SurfEBMut& m = eb;
    if (is_final) ctx_get_words(cx, CO_EBM, m, 0, EBM_W_KEEP);
    else {
      if (cls & EBG_INCL) ctx_get_words(cx, CO_EBM, m, 0, EBM_W_FEED);
      ctx_get_words(cx, CO_EBM, m, EBM_W_FEED, EBM_W_IN3);
      if (cls & EBG_SNOWCOV) ctx_get_words(cx, CO_EBM, m, EBM_W_IN3, EBM_W_TSNOW);
      if (cls & EBG_CANOPY) ctx_get_words(cx, CO_EBM, m, EBM_W_RA1, EBM_W_RA1 + 1);
    }
  }
  {
    const VegLib vl{a.veglib};
    const Forcing fc{a.forcing, nullptr, a.ncell, c, a.o.NR + 1};
    surf_cell_fill(eb, cv, vl, s3, fc, eb.hidx, veg_idx, a.month);
  }
  const double* __restrict__ rec = a.pout + (size_t)g * pout_hru_stride(a.Nn);
  const double* __restrict__ po = rec + slot * pout_stride(a.Nn);
  const bool ok = (((unsigned long long)__double_as_longlong(po[a.Nn])) >> 32) & 1ull;
  if (sv.stage == SurfSolve::FINAL) sv.final_slot = slot;
  const double fx = ok ? eb.eval(a.o, s3, sv.x, po[1], po[2]) : ERROR_VAL;
  const bool was_quick = sv.stage == SurfSolve::ROOT_QUICK;
  const int stage_before = sv.stage;
  const double x_eval = sv.x;
  surf_solve_consume(a.o, sv, eb, eb, fx);
  // The iteration has just ended on the point it has just evaluated (the usual end of a Brent iteration: the newest point is
  // the best one): the evaluation "at the root" the reference makes next (calc_surf_energy_bal.c:489-506) would repeat this
  // one -- same trial temperature, same profile record, and for an HRU without a thin snowpack nothing an evaluation leaves
  // behind feeds the next -- so it is booked as done here instead of in another round.  Not taken: thin snowpack (the vapour
  // fluxes are carried from call to call), fallback / error results, QUICK_SOLVE and IMPLICIT (their final evaluation solves
  // another column / is always solved again).
  bool at_root = false;
#if VIC_FINAL_SHORTCUT
  if (stage_before == SurfSolve::ROOT && sv.stage == SurfSolve::FINAL && !a.implicit && !a.o.QUICK_SOLVE && !(cls & EBG_INCL) && sv.ok
      && sv.fbflag == 0 && sv.Tsurf == x_eval && fabs(fx) < 1.e30) {
    sv.final_slot = slot;
    surf_solve_consume(a.o, sv, eb, eb, fx);       // FINAL -> DONE with this evaluation's residual
    at_root = true;
  }
#endif
  if (was_quick && sv.stage != SurfSolve::ROOT_QUICK) {
    // QUICK_SOLVE: from here on the whole column is solved; the records of the shortened column are not its solutions.  NOFLUX
    // comes back with a second iteration only (calc_surf_energy_bal.c:403); the final evaluation keeps what was last set
    a.jl[g] = (sv.stage == SurfSolve::ROOT && a.o.NOFLUX) ? a.Nn : a.Nn - 1;
    a.pout[(size_t)g * pout_hru_stride(a.Nn) + pout_key(a.Nn, 0)] = NAN;
    a.pout[(size_t)g * pout_hru_stride(a.Nn) + pout_key(a.Nn, 1)] = NAN;
  }
  bool need_solve = sv.stage != SurfSolve::DONE;
  if (sv.stage == SurfSolve::FINAL && !a.implicit) {
    // the root has been found: the final evaluation needs the profile at sv.x, which is on record if sv.x is one of the
    // last two trial points; the evaluation itself happens in the next round, together with everybody else's (making it
    // here, in a second pass over eval(), costs the kernel 548 B of scratch per lane and 5 ms per step: measured, dropped)
    if (rec[pout_key(a.Nn, slot)] == sv.x) { sv.final_slot = slot; sv.on_record = 1; need_solve = false; }
    else if (rec[pout_key(a.Nn, slot ^ 1)] == sv.x) { sv.final_slot = slot ^ 1; sv.on_record = 1; need_solve = false; }
  }
  // while the Brent iteration goes on only its own state and the next abscissa change: the tail of SurfSolve (result,
  // flags, stage, record bookkeeping) is written when it does
  if (sv.stage == stage_before && (sv.stage == SurfSolve::ROOT || sv.stage == SurfSolve::ROOT_QUICK)) ctx_put_words(cx, CO_SV, sv, 0, (int)CW_SV_ITER);
  else ctx_put(cx, CO_SV, sv);
  if (sv.stage == SurfSolve::DONE) {
    const SurfEBMut& m = eb;
    if (at_root) {
      // an evaluation of the iteration has not fetched the inputs it only passes through: written back are the outputs and
      // what this HRU's branch of the evaluation assigns (vic_surface.hpp); the rest keeps the values the set-up parked
      ctx_put_words(cx, CO_EBM, m, EBM_W_KEEP, (int)CW_EBM);
      ctx_put_words(cx, CO_EBM, m, EBM_W_TSNOW, EBM_W_TSNOW + 1);                                   // ra_used[0]
      if (cls & EBG_FROZEN) ctx_put_words(cx, CO_EBM, m, EBM_W_IN3 - 1, EBM_W_IN3);                 // fusion
      if (cls & EBG_CANOPY) ctx_put_words(cx, CO_EBM, m, EBM_W_VV, EBM_W_VV + 6);                   // vv, layerevap[3]
      else if (cls & EBG_EVAP) ctx_put_words(cx, CO_EBM, m, EBM_W_VV + 3, EBM_W_VV + 4);            // layerevap[0] (arno_evap)
    } else ctx_put(cx, CO_EBM, m);
  } else if (cls & EBG_INCL) ctx_put_words(cx, CO_EBM, static_cast<const SurfEBMut&>(eb), 0, EBM_W_FEED);
  if (sv.stage == SurfSolve::DONE) a.hstate[g] = 2;
  else if (need_solve) { a.ts[g] = sv.x; a.pslot[g] = slot ^ 1; }     // keep the record just used, overwrite the older one
  list_append(a.list_next, a.count_next, a.list_cap, need_solve, a.hkey[g], g);
  {
    const bool eo = sv.stage != SurfSolve::DONE && !need_solve;
    const unsigned long long m = __ballot(eo);
    if (m != 0) {
      const int lane = (int)__lane_id(), lead = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == lead) base = atomicAdd(a.evalonly, __popcll(m));
      base = __shfl(base, lead);
      if (eo) a.eo_list_next[base + __popcll(m & ((1ull << lane) - 1ull))] = g;
    }
  }
}

// ------------------------------------------------------------------------------------------------ glacier mass-balance fit
// GlacierMassBalanceResult.c:34-73 + GraphingEquation.c:8-125 for every cell at once (lane = cell): the accumulated
// mass balance of the cell's glacier HRUs against band elevation, points merged per elevation in hruList order, closed-form
// normal equations in the reference's order of operations; then resetAccumulationValues
// (accumulateGlacierMassBalance.c:5-11) when asked.
struct GArgs {
  Opt o;
  int ncell, nhru, reset;
  const double* cell_params;
  const int* cell_off;
  const int* cell_list;
  const int* hpi;
  double* sd;
  double* eq;          // [GMB_NROW][ncell]
};

__global__ __launch_bounds__(64) void vic_glacier_fit(const GArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.ncell) return;
  const size_t nh = a.nhru, nc = a.ncell;
  CellView cv{a.cell_params, a.ncell, c, a.o.Nnode, a.o.Nband};
  double X[VIC_MAX_BANDS], Y[VIC_MAX_BANDS];      // at most one point per band elevation
  int np = 0;
  for (int k = a.cell_off[c]; k < a.cell_off[c + 1]; k++) {
    const int g = a.cell_list[k];
    if (a.hpi[(size_t)HPI_IS_GLACIER * nh + g] == 0) continue;
    const double cum = a.sd[(size_t)SD_GLAC_CUM_MASS_BALANCE * nh + g];
    if (!isnan(cum)) {
      const double x = cv.band(CPB_BANDELEV, a.hpi[(size_t)HPI_BAND * nh + g]);
      bool found = false;
      for (int j = 0; j < np; j++)
        if (X[j] == x) { Y[j] += cum; found = true; }
      if (!found && np < VIC_MAX_BANDS) { X[np] = x; Y[np] = cum; np++; }
    }
    if (a.reset) a.sd[(size_t)SD_GLAC_CUM_MASS_BALANCE * nh + g] = 0.0;
  }
  int k2 = 0;
  for (int i = 0; i < np; i++)
    if (!(X[i] == 0)) { X[k2] = X[i]; Y[k2] = Y[i]; k2++; }       // "meaningless" points (GlacierMassBalanceResult.c:58-66)
  np = k2;
  double b0 = 0, b1 = 0, b2 = 0, fit = -1;
  if (np == 1) b0 = Y[0];
  else if (np == 2) {
    const double slope = (Y[1] - Y[0]) / (X[1] - X[0]);
    b0 = Y[0] - slope * X[0]; b1 = slope;
  } else if (np >= 3) {
    double sumx4 = 0, sumx3 = 0, sumx2 = 0, sumx1 = 0;
    const int size = np;
    for (int i = 0; i < np; i++) {
      sumx4 += X[i] * X[i] * X[i] * X[i];
      sumx3 += X[i] * X[i] * X[i];
      sumx2 += X[i] * X[i];
      sumx1 += X[i];
    }
    const double det = (sumx4 * sumx2 * size) + (sumx3 * sumx1 * sumx2) + (sumx2 * sumx3 * sumx1) - (sumx2 * sumx2 * sumx2)
                       - (sumx1 * sumx1 * sumx4) - (size * sumx3 * sumx3);
    const double inv[3][3] = {{size * sumx2 - sumx1 * sumx1, -(size * sumx3 - sumx1 * sumx2), sumx1 * sumx3 - sumx2 * sumx2},
                              {-(size * sumx3 - sumx2 * sumx1), size * sumx4 - sumx2 * sumx2, -(sumx1 * sumx4 - sumx3 * sumx2)},
                              {sumx1 * sumx3 - sumx2 * sumx2, -(sumx1 * sumx4 - sumx2 * sumx3), sumx2 * sumx4 - sumx3 * sumx3}};
    double acoef[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < np; j++) {
        const double stuff = inv[i][0] * (X[j] * X[j]) + inv[i][1] * X[j] + inv[i][2] * 1;
        acoef[i] += stuff * Y[j];
      }
      acoef[i] /= det;
    }
    b0 = acoef[2]; b1 = acoef[1]; b2 = acoef[0];
  }
  if (np > 0) {
    fit = 0;
    for (int i = 0; i < np; i++) fit += fabs((b0 + b1 * X[i] + b2 * (X[i] * X[i])) - Y[i]);
  }
  a.eq[(size_t)GMB_B0 * nc + c] = b0; a.eq[(size_t)GMB_B1 * nc + c] = b1; a.eq[(size_t)GMB_B2 * nc + c] = b2;
  a.eq[(size_t)GMB_FIT_ERROR * nc + c] = fit;
}

// ------------------------------------------------------------------------------------------------ derived cell rows
__global__ __launch_bounds__(256) void vic_derive_cell_params(double* cp, int ncell, int Nn, int Nb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncell) return;
  CellView cv{cp, ncell, c, Nn, Nb};
#pragma unroll
  for (int l = 0; l < VIC_NLAYER; l++) {
    const SoilKLayer k = soil_conductivity_layer_constants(cv.lay(CPL_SOIL_DENS_MIN, l), cv.lay(CPL_BULK_DENS_MIN, l), cv.lay(CPL_QUARTZ, l),
                                                           cv.lay(CPL_SOIL_DENSITY, l), cv.lay(CPL_BULK_DENSITY, l), cv.lay(CPL_ORGANIC, l));
    cp[(size_t)VIC_CPX_ROW(CPX_KDRY, l, Nn, Nb) * ncell + c] = k.Kdry;
    cp[(size_t)VIC_CPX_ROW(CPX_KSP, l, Nn, Nb) * ncell + c] = k.KsP;
    cp[(size_t)VIC_CPX_ROW(CPX_KWP, l, Nn, Nb) * ncell + c] = k.KwP;
    cp[(size_t)VIC_CPX_ROW(CPX_POROSITY, l, Nn, Nb) * ncell + c] = k.porosity;
  }
}

// TreeAdjustFactor of put_data.c:185-208 for every band of every cell (lane = cell; once per vicgpu_set_domain)
__global__ __launch_bounds__(64) void vic_derive_tree_adjust(double* cp, int ncell, int nhru, int Nn, int Nb, const int* cell_off, const int* cell_list,
                                                             const int* hpi, const double* hpd, const double* veglib) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= ncell) return;
  for (int b = 0; b < Nb; b++) {
    double bandCv = 0;
    for (int k = cell_off[c]; k < cell_off[c + 1]; k++) {          // hruList order, like the reference's sum
      const int g = cell_list[k];
      if (hpi[(size_t)HPI_BAND * nhru + g] != b) continue;
      if (veglib[(size_t)hpi[(size_t)HPI_VEG_INDEX * nhru + g] * VL_NFIELD + VL_OVERSTORY] != 0.0) bandCv += hpd[(size_t)HPD_CV * nhru + g];
    }
    const bool atl = cp[(size_t)VICGPU_CP_BAND(CPB_ABOVETREELINE, b, Nn, Nb) * ncell + c] != 0.0;
    cp[(size_t)VIC_CPX_TREE_ROW(b, Nn, Nb) * ncell + c] = atl ? 1. / (1. - bandCv) : 1.;
  }
}

// ------------------------------------------------------------------------------------------------ test hook
struct DArgs { Opt o; const double* cell_params; int ncell, fn, n; const double* in; double* out; };

__global__ __launch_bounds__(64) void vic_debug_pure(const DArgs d) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= d.n) return;
  const double* a = d.in + (size_t)i * VICGPU_PURE_NIN;
  CellView cv{d.cell_params, d.ncell, 0, d.o.Nnode, d.o.Nband};
  double r = NAN;
  switch (d.fn) {
    case VICGPU_PURE_SVP: r = svp(a[0]); break;
    case VICGPU_PURE_SVP_SLOPE: r = svp_slope(a[0]); break;
    case VICGPU_PURE_CALC_RAINONLY: r = calc_rainonly(d.o, a[0], a[1], a[2], a[3]); break;
    case VICGPU_PURE_SNOW_ALBEDO: r = snow_albedo(d.o, cv, a[0], a[1], a[2], a[3], a[4], a[5], (int)a[6], a[7] != 0.0 ? 1 : 0); break;
    case VICGPU_PURE_NEW_SNOW_DENSITY: r = new_snow_density(d.o, a[0]); break;
    case VICGPU_PURE_STABILITY: r = stability_correction(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case VICGPU_PURE_PENMAN: r = penman(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case VICGPU_PURE_CALC_RC: r = calc_rc(a[0], a[1], (float)a[2], a[3], a[4], a[5], a[6], a[7] != 0.0); break;
    case VICGPU_PURE_ESTIMATE_T1: r = estimate_T1(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); break;
    case VICGPU_PURE_SOIL_CONDUCTIVITY: r = soil_conductivity(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
    case VICGPU_PURE_VOL_HEAT_CAPACITY: r = volumetric_heat_capacity(a[0], a[1], a[2], a[3]); break;
    case VICGPU_PURE_MAX_UNFROZEN_WATER: r = maximum_unfrozen_water(a[0], a[1], a[2], a[3]); break;
    case VICGPU_PURE_LINEAR_INTERP: r = linear_interp(a[0], a[1], a[2], a[3], a[4]); break;
    case VICGPU_PURE_VEG_HEIGHT: r = calc_veg_height(a[0], a[1]); break;
    case VICGPU_PURE_SOIL_CONDUCTIVITY_DERIVED: {
      const int l = (int)a[2];
      const SoilKLayer kc{cv.x(CPX_KDRY, l), cv.x(CPX_KSP, l), cv.x(CPX_KWP, l), cv.x(CPX_POROSITY, l)};
      r = soil_conductivity_pre(a[0], a[1], kc);
      break;
    }
    default: break;
  }
  d.out[i] = r;
}

// ------------------------------------------------------------------------------------------------ cell kernel
struct CArgs {
  int ncell, nhru;
  int c0, ccount;        // cells of this launch
  const int* cell_off;
  const int* cell_list;
  const double* hpd;
  const int* hpi_glac;   // row HPI_IS_GLACIER of the int parameter table
  const double* flux;
  const double* sd;
  const int* hru_err;
  double* cell_out;   // [CO_NROW][ncell]
  double* accum;      // [CA_NROW][ncell]
  int* cell_err;      // [ncell], OR-accumulated
};

__global__ __launch_bounds__(256) void vic_cell_reduce(const CArgs a) {
  const int ci = blockIdx.x * 256 + threadIdx.x;
  if (ci >= a.ccount) return;
  const int c = a.c0 + ci;
  const size_t nh = a.nhru, nc = a.ncell;
  double op = 0, orn = 0, os = 0, ro = 0, bf = 0, ev = 0, swe = 0, sm0 = 0, sm1 = 0, sm2 = 0, gmb = 0;
  int err = 0;
  for (int k = a.cell_off[c]; k < a.cell_off[c + 1]; k++) {
    const int g = a.cell_list[k];
    const double Cv = a.hpd[(size_t)HPD_CV * nh + g];
    op += a.flux[(size_t)FX_OUT_PREC * nh + g] * Cv;            // full_energy.c:429-431
    orn += a.flux[(size_t)FX_OUT_RAIN * nh + g] * Cv;
    os += a.flux[(size_t)FX_OUT_SNOW * nh + g] * Cv;
    ro += a.flux[(size_t)FX_RUNOFF * nh + g] * Cv;              // put_data.c:789-800 AreaFactor = Cv (mu = TreeAdjust = 1)
    bf += a.flux[(size_t)FX_BASEFLOW * nh + g] * Cv;
    double e = a.flux[(size_t)FX_EVAP0 * nh + g] + a.flux[(size_t)FX_EVAP1 * nh + g] + a.flux[(size_t)FX_EVAP2 * nh + g]
               + a.flux[(size_t)FX_CANOPYEVAP * nh + g]
               + (a.flux[(size_t)FX_SNOW_VAPOR_FLUX * nh + g] + a.flux[(size_t)FX_SNOW_CANOPY_VAPOR_FLUX * nh + g]) * 1000.;
    ev += e * Cv;
    swe += a.sd[(size_t)SD_SNOW_SWQ * nh + g] * 1000. * Cv;
    sm0 += a.sd[(size_t)SD_MOIST0 * nh + g] * Cv;
    sm1 += a.sd[(size_t)SD_MOIST1 * nh + g] * Cv;
    sm2 += a.sd[(size_t)SD_MOIST2 * nh + g] * Cv;
    { double mb = a.flux[(size_t)FX_GLAC_MASS_BALANCE * nh + g]; if (a.hpi_glac[g] && !isnan(mb)) gmb += mb * Cv; }
    err |= a.hru_err[g];
  }
  a.cell_out[(size_t)CO_OUT_PREC * nc + c] = op;
  a.cell_out[(size_t)CO_OUT_RAIN * nc + c] = orn;
  a.cell_out[(size_t)CO_OUT_SNOW * nc + c] = os;
  a.accum[(size_t)CA_RUNOFF * nc + c] += ro;
  a.accum[(size_t)CA_BASEFLOW * nc + c] += bf;
  a.accum[(size_t)CA_EVAP * nc + c] += ev;
  a.accum[(size_t)CA_PREC * nc + c] += op;
  a.accum[(size_t)CA_SWE_END * nc + c] = swe;
  a.accum[(size_t)CA_SOIL_MOIST_END0 * nc + c] = sm0;
  a.accum[(size_t)CA_SOIL_MOIST_END1 * nc + c] = sm1;
  a.accum[(size_t)CA_SOIL_MOIST_END2 * nc + c] = sm2;
  a.accum[(size_t)CA_GLAC_MASS_BALANCE * nc + c] += gmb;
  a.accum[(size_t)CA_NSTEPS * nc + c] += 1.0;
  a.cell_err[c] |= err;
}

// ------------------------------------------------------------------------------------------------ state-file records
// One lane per HRU in hruList order: the HRU's values in the order processCellForStateFile streams them
// (write_model_state.c:166-285).  GATHER = false is the read side; lanes whose band / vegetation class do not match the
// record count themselves in *mismatch and scatter nothing.
struct RArgs {
  int nhru, Nn;
  const int* cell_list;
  const int* hpi;
  double* sd;
  int* si;
  double* flux;
  double* rec;
  int* mismatch;
};

template <bool GATHER>
__global__ __launch_bounds__(256) void vic_state_records(const RArgs a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.nhru) return;
  const int g = a.cell_list[k], Nn = a.Nn;
  const size_t nh = a.nhru;
  double* r = a.rec + (size_t)k * VICGPU_SR_LEN(Nn);
  const int band = a.hpi[(size_t)HPI_BAND * nh + g], vegc = a.hpi[(size_t)HPI_VEG_CLASS * nh + g];
  if (GATHER) { r[SR_BAND_INDEX] = band; r[SR_VEG_CLASS] = vegc; }
  else if ((int)r[SR_BAND_INDEX] != band || (int)r[SR_VEG_CLASS] != vegc) { atomicAdd(a.mismatch, 1); return; }
#define D(slot, row) do { if (GATHER) r[slot] = a.sd[(size_t)(row) * nh + g]; else a.sd[(size_t)(row) * nh + g] = r[slot]; } while (0)
#define I(slot, row) do { if (GATHER) r[slot] = a.si[(size_t)(row) * nh + g]; else a.si[(size_t)(row) * nh + g] = (int)r[slot]; } while (0)
#define F(slot, row) do { if (GATHER) r[slot] = a.flux[(size_t)(row) * nh + g]; else a.flux[(size_t)(row) * nh + g] = r[slot]; } while (0)
  for (int l = 0; l < 3; l++) { D(SR_MOIST0 + l, SD_MOIST0 + l); D(SR_ICE0 + l, SD_ICE0 + l); }
  D(SR_WDEW, SD_WDEW);
  D(SR_SNOW_CANOPY, SD_SNOW_CANOPY); D(SR_SNOW_DENSITY, SD_SNOW_DENSITY); D(SR_SNOW_DEPTH, SD_SNOW_DEPTH);
  D(SR_SNOW_PACK_WATER, SD_SNOW_PACK_WATER); D(SR_SNOW_SURF_WATER, SD_SNOW_SURF_WATER); D(SR_SNOW_SWQ, SD_SNOW_SWQ);
  D(SR_GLAC_WATER_STORAGE, SD_GLAC_WATER_STORAGE); D(SR_GLAC_CUM_MASS_BALANCE, SD_GLAC_CUM_MASS_BALANCE);
  for (int n = 0; n < Nn; n++) { D(SR_ENERGY_T + n, VICGPU_SD_NODE(SDN_T, n, Nn)); I(VICGPU_SR_T(SRT_T_FBCOUNT, Nn) + n, VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn)); }
  D(VICGPU_SR_T(SRT_TFOLIAGE, Nn), SD_TFOLIAGE); D(VICGPU_SR_T(SRT_GLAC_SURF_TEMP, Nn), SD_GLAC_SURF_TEMP);
  D(VICGPU_SR_T(SRT_SNOW_COLD_CONTENT, Nn), SD_SNOW_COLDCONTENT); D(VICGPU_SR_T(SRT_SNOW_PACK_TEMP, Nn), SD_SNOW_PACK_TEMP);
  D(VICGPU_SR_T(SRT_SNOW_SURF_TEMP, Nn), SD_SNOW_SURF_TEMP); D(VICGPU_SR_T(SRT_SNOW_ALBEDO, Nn), SD_SNOW_ALBEDO);
  I(VICGPU_SR_T(SRT_SNOW_LAST_SNOW, Nn), SI_SNOW_LAST_SNOW); I(VICGPU_SR_T(SRT_SNOW_MELTING, Nn), SI_SNOW_MELTING);
  I(VICGPU_SR_T(SRT_TCANOPY_FBCOUNT, Nn), SI_TCANOPY_FBCOUNT);
  I(VICGPU_SR_U(SRU_TFOLIAGE_FBCOUNT, Nn), SI_TFOLIAGE_FBCOUNT); I(VICGPU_SR_U(SRU_TSURF_FBCOUNT, Nn), SI_TSURF_FBCOUNT);
  I(VICGPU_SR_U(SRU_GLAC_SURF_TEMP_FBCOUNT, Nn), SI_GLAC_SURF_TEMP_FBCOUNT); I(VICGPU_SR_U(SRU_SNOW_SURF_TEMP_FBCOUNT, Nn), SI_SNOW_SURF_TEMP_FBCOUNT);
  I(VICGPU_SR_U(SRU_GLAC_SURF_TEMP_FBFLAG, Nn), SI_GLAC_SURF_TEMP_FBFLAG);
  F(VICGPU_SR_U(SRU_GLAC_VAPOR_FLUX, Nn), FX_GLAC_VAPOR_FLUX);
  if (GATHER) r[VICGPU_SR_U(SRU_SNOW_CANOPY_ALBEDO, Nn)] = 0.0;            // snow.canopy_albedo: initialize_snow.c:62, never assigned again
  F(VICGPU_SR_U(SRU_SNOW_SURFACE_FLUX, Nn), FX_SNOW_SURFACE_FLUX);
  I(VICGPU_SR_U(SRU_SNOW_SURF_TEMP_FBFLAG, Nn), SI_SNOW_SURF_TEMP_FBFLAG);
  D(VICGPU_SR_U(SRU_SNOW_TMP_INT_STORAGE, Nn), SD_SNOW_TMP_INT_STORAGE);
  F(VICGPU_SR_U(SRU_SNOW_VAPOR_FLUX, Nn), FX_SNOW_VAPOR_FLUX);
#undef D
#undef I
#undef F
}

// ------------------------------------------------------------------------------------------------ context
// A chunk of cells with all their HRUs.  Cells never interact, so every chunk runs the whole step sequence on its own
// stream, driven by its own host thread: while one chunk is in the thin tail of its Brent rounds (a few stragglers,
// latency bound) or in a stage kernel (memory / latency bound), the profile solves of the others fill the SIMDs.
// The host reads the round's list sizes back RB_LAG rounds late (fd_step): it stays that many rounds ahead of the device, so the
// thin tail rounds -- two short kernels each -- never wait for a host round trip; the price is RB_LAG rounds on empty lists at
// the end of the iteration (both kernels return at once).
#ifndef VIC_RB_LAG
#define VIC_RB_LAG 3
#endif
constexpr int RB_LAG = VIC_RB_LAG, RB_DEPTH = RB_LAG + 1;
struct FdChunk {
  int c0 = 0, ccount = 0;          // cells [c0, c0 + ccount)
  int* d_glist = nullptr;          // their HRUs, ascending
  int gcount = 0;
  LaunchMap map;                   // XCD-aware launch order when the chunk's list is regular (slot-major, every slot ccount cells)
  int* d_list[2] = {nullptr, nullptr};   // work lists (HRU ids)
  int *d_fb_list = nullptr, *d_fb_count = nullptr;   // IMPLICIT: HRUs whose Newton iteration failed this round
  int* d_count = nullptr;          // counter block (CNT_*): segment sizes of the two lists, profile cursor, evaluation-only counts, pending total
  int* d_elist[2] = {nullptr, nullptr};  // evaluation-only lists (flat, gcount entries): pending HRUs that need no solve
  int list_cap = 0;                // entries per segment
  int* h_count = nullptr;          // pinned read-back, RB_DEPTH slots of CNT_TOTAL
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  hipEvent_t readback[RB_DEPTH] = {};
  std::string err;
  int status = 0;
  long long rounds = 0, steps = 0;
};

struct vicgpu_ctx {
  vicgpu_options opt;
  Opt o;
  int device;
  std::string err;
  int ncell = 0, nhru = 0, nveg_rows = 0;
  bool domain_ready = false;       // set at the end of a successful vicgpu_set_domain, cleared by free_domain
  double *d_veglib = nullptr, *d_cp = nullptr, *d_hpd = nullptr, *d_sd = nullptr, *d_flux = nullptr, *d_forcing = nullptr,
         *d_cell_out = nullptr, *d_accum = nullptr;
  int *d_hpi = nullptr, *d_si = nullptr, *d_cell_off = nullptr, *d_cell_list = nullptr, *d_hru_err = nullptr, *d_cell_err = nullptr;
  unsigned char* d_snowflag = nullptr;
  // forcing: d_forcing / d_snowflag / dmy / chunk_steps describe the CURRENT chunk = slot[cur]; the other slot takes the
  // prefetch of the next one (vicgpu_prefetch_forcing*, vicgpu_swap_forcing)
  std::vector<int> dmy;            // host copy [nsteps][VIC_NDMY]
  int chunk_steps = 0;
  struct ForcingSlot {
    double *d_f = nullptr, *d_raw = nullptr;
    unsigned char* d_s = nullptr;
    size_t fcap = 0, scap = 0, rawcap = 0, stage_cap = 0;
    void* h_stage = nullptr;       // pinned staging for pageable sources
    std::vector<int> dmy;
    int nsteps = 0;
    hipEvent_t uploaded = nullptr; // copy stream: the chunk is in the slot
    hipEvent_t released = nullptr; // context stream: every step that read the slot has been queued before it
    bool upload_pending = false, was_current = false;
  } slot[2];
  int cur = -1, staged = -1;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  bool own_stream = true;
  std::vector<hipEvent_t> ev;      // start/stop pairs of the last vicgpu_step call
  int ev_used = 0;
  int write_fluxes = 1;
  int steps_done = 0;
  bool any_glacier = false;
  // finite-difference pipeline workspace (allocated when QUICK_FLUX is off)
  bool fd = false;
  unsigned long long* d_ctx = nullptr;
  double *d_pin = nullptr, *d_ts = nullptr, *d_pout = nullptr;
  int *d_hstate = nullptr, *d_pslot = nullptr, *d_hkey = nullptr, *d_lastexp = nullptr, *d_jl = nullptr;
  double* d_pimp = nullptr;        // IMPLICIT only
  int profile_waves = 0;           // resident waves of the profile kernel
  int eval_list_pct = 30;          // sparse evaluation rounds (lane = pending HRU) once at most this percentage of the HRUs is pending
  bool node_newton = false;        // frozen-node root finder: safeguarded Newton instead of the reference's Brent iteration
  std::vector<FdChunk> chunks;     // cell chunks, each an independent pipeline on its own stream
  int ev_steps = 0;                // steps covered by the event pair of the last vicgpu_step call
  // put_data (vicgpu_out.h): output tables [nrow][ncell], allocated by vicgpu_put_data_config
  bool put_on = false;
  int out_step_ratio = 1, out_nrow = 0;
  OutLayout out_lay;
  double *d_out_data = nullptr, *d_out_agg = nullptr, *d_pb = nullptr;
  unsigned char* d_rowagg = nullptr;   // [out_nrow] aggregation type of every output row
};

static void free_domain(vicgpu_ctx* c) {
  void* ps[] = {c->d_cp, c->d_hpd, c->d_sd, c->d_flux, c->d_cell_out, c->d_accum, c->d_hpi, c->d_si, c->d_cell_off, c->d_cell_list,
                c->d_hru_err, c->d_cell_err, c->d_ctx, c->d_pin, c->d_ts, c->d_pout, c->d_hstate, c->d_pslot, c->d_hkey,
                c->d_out_data, c->d_out_agg, c->d_pb, c->d_rowagg, c->d_pimp, c->d_lastexp, c->d_jl};
  for (void* p : ps) HIPIGN(hipFree(p));
  c->d_out_data = c->d_out_agg = c->d_pb = nullptr;
  c->d_rowagg = nullptr;
  c->d_pimp = nullptr; c->d_lastexp = nullptr; c->d_jl = nullptr;
  c->put_on = false;
  for (FdChunk& ch : c->chunks) {
    HIPIGN(hipFree(ch.d_glist)); HIPIGN(hipFree(ch.d_list[0])); HIPIGN(hipFree(ch.d_list[1])); HIPIGN(hipFree(ch.d_count));
    HIPIGN(hipFree(ch.d_elist[0])); HIPIGN(hipFree(ch.d_elist[1]));
    HIPIGN(hipFree(ch.d_fb_list)); HIPIGN(hipFree(ch.d_fb_count));
    if (ch.h_count) HIPIGN(hipHostFree(ch.h_count));
    if (ch.done) HIPIGN(hipEventDestroy(ch.done));
    for (hipEvent_t e : ch.readback) if (e) HIPIGN(hipEventDestroy(e));
    if (ch.stream) HIPIGN(hipStreamDestroy(ch.stream));
  }
  c->chunks.clear();
  c->domain_ready = false;
  c->chunk_steps = 0;              // a forcing chunk belongs to the domain it was pushed for (its rows are ncell wide)
  c->dmy.clear();
  c->cur = c->staged = -1;
  c->d_forcing = nullptr; c->d_snowflag = nullptr;
  c->d_ctx = nullptr; c->d_pin = c->d_ts = c->d_pout = nullptr; c->d_hstate = c->d_pslot = c->d_hkey = nullptr;
  c->d_cp = c->d_hpd = c->d_sd = c->d_flux = c->d_cell_out = c->d_accum = nullptr;
  c->d_hpi = c->d_si = c->d_cell_off = c->d_cell_list = c->d_hru_err = c->d_cell_err = nullptr;
}

template <int NN>
static hipError_t launch_hru(const KArgs& ka, hipStream_t st, bool ordinary, bool glacier) {
  const int nblk = ka.map.nblocks(ka.gcount);
  // ordinary HRUs run the monolithic kernel with QUICK_FLUX only (Nnode == 3); the other node counts never instantiate it
  if constexpr (NN == 3) {
    if (ordinary) hipLaunchKernelGGL((vic_hru_step<NN, false>), dim3(nblk), dim3(64), 0, st, ka);
  }
  if (glacier) hipLaunchKernelGGL((vic_hru_step<NN, true>), dim3(nblk), dim3(64), 0, st, ka);
  return hipGetLastError();
}

template <int NN>
static hipError_t launch_fd_stage(const KArgs& ka, bool multi, hipStream_t st) {
  const dim3 grid(ka.map.nblocks(ka.gcount)), block(64);
  if (ka.phase == 0) {
    if (multi) hipLaunchKernelGGL((vic_fd_stage<NN, true, true>), grid, block, 0, st, ka);
    else hipLaunchKernelGGL((vic_fd_stage<NN, true, false>), grid, block, 0, st, ka);
  } else {
    if (multi) hipLaunchKernelGGL((vic_fd_stage<NN, false, true>), grid, block, 0, st, ka);
    else hipLaunchKernelGGL((vic_fd_stage<NN, false, false>), grid, block, 0, st, ka);
  }
  return hipGetLastError();
}

// The profile kernel of a node count: 10 nodes have the register-resident instantiation, every other count the generic one
template <int NN>
static hipError_t launch_profile(const PArgs& pa, int nmax, int resident_waves, bool newton, hipStream_t st) {
  int nblk = (nmax + 63) / 64;
  if (nblk > resident_waves) nblk = resident_waves;      // persistent waves pull from the work list
  if (nblk < 1) nblk = 1;                                // block 0 also clears the counters of the round
  const dim3 g(nblk), b(64);
  if constexpr (NN == 10) {
    if (newton) hipLaunchKernelGGL((vic_profile_solve_reg<NN, true>), g, b, 0, st, pa);
    else hipLaunchKernelGGL((vic_profile_solve_reg<NN, false>), g, b, 0, st, pa);
  } else {
    if (newton) hipLaunchKernelGGL((vic_profile_solve_lockstep<NN, true>), g, b, 0, st, pa);
    else hipLaunchKernelGGL((vic_profile_solve_lockstep<NN, false>), g, b, 0, st, pa);
  }
  return hipGetLastError();
}

template <int NN>
static int profile_resident_waves(int device, bool newton) {
  int per_cu = 0, ncu = 0;
  hipError_t e;
  if constexpr (NN == 10) e = newton ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vic_profile_solve_reg<NN, true>, 64, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vic_profile_solve_reg<NN, false>, 64, 0);
  else e = newton ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vic_profile_solve_lockstep<NN, true>, 64, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vic_profile_solve_lockstep<NN, false>, 64, 0);
  if (e != hipSuccess || per_cu <= 0) per_cu = 8;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
  return per_cu * ncu;
}

#define CHKCH(ch, call)                                                                               \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      (ch)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
      return VICGPU_ERR_HIP;                                                                           \
    }                                                                                                  \
  } while (0)

// Counter block of a chunk.  Every group sits on its own 128-byte lines: the evaluation kernel's waves all read the pending
// count while others append to the next list with atomics, and reads that share a line with those atomics queue behind them
// in the L2 channel (measured: the dense evaluation rounds went from 0.6 to 1.4-2.5 ms when they did).
constexpr int CNT_LIST_STRIDE = 64, CNT_CURSOR = 128, CNT_EVALONLY = 160, CNT_EVALONLY_STRIDE = 32, CNT_NPEND = 224, CNT_TOTAL = 256;
static_assert(NBUCKET <= CNT_LIST_STRIDE, "counter block layout");
static inline int* cnt_list(int* d_count, int l) { return d_count + l * CNT_LIST_STRIDE; }
static inline int* cnt_evalonly(int* d_count, int l) { return d_count + CNT_EVALONLY + l * CNT_EVALONLY_STRIDE; }

// One model step of the finite-difference pipeline for one chunk (see the header of this file).  Blocks the calling
// host thread: the number of Brent rounds is data dependent, so the pending count is read back once the first rounds
// are through.
static int fd_read_count(FdChunk* ch, int which, int* nsolve, int* nevalonly) {
  CHKCH(ch, hipMemcpyAsync(ch->h_count, ch->d_count, sizeof(int) * CNT_TOTAL, hipMemcpyDeviceToHost, ch->stream));
  CHKCH(ch, hipStreamSynchronize(ch->stream));
  int n = 0;
  for (int b = 0; b < NBUCKET; b++) n += ch->h_count[which * CNT_LIST_STRIDE + b];
  *nsolve = n;
  *nevalonly = ch->h_count[CNT_EVALONLY + which * CNT_EVALONLY_STRIDE];
  return VICGPU_OK;
}

static int fd_step(vicgpu_ctx* c, FdChunk* ch, KArgs ka) {
  const int Nn = c->o.Nnode;
  const bool n10 = (Nn == 10);
  hipStream_t st = ch->stream;
  if (c->any_glacier) CHKCH(ch, (n10 ? launch_hru<10>(ka, st, false, true) : launch_hru<VIC_MAX_NODES>(ka, st, false, true)));
  CHKCH(ch, hipMemsetAsync(ch->d_count, 0, sizeof(int) * CNT_TOTAL, st));
  int cur = 0;
  ka.phase = 0; ka.list = ch->d_list[cur]; ka.count = cnt_list(ch->d_count, cur); ka.list_cap = ch->list_cap;
  CHKCH(ch, (n10 ? launch_fd_stage<10>(ka, c->o.NF > 1, st) : launch_fd_stage<VIC_MAX_NODES>(ka, c->o.NF > 1, st)));
  PArgs pa;
  pa.pin = c->d_pin; pa.ts = c->d_ts; pa.pout = c->d_pout; pa.pslot = c->d_pslot; pa.Nn = Nn; pa.NOFLUX = c->o.NOFLUX; pa.EXP_TRANS = c->o.EXP_TRANS;
  pa.TFALLBACK = c->o.TFALLBACK; pa.next = ch->d_count + CNT_CURSOR; pa.cap = ch->list_cap; pa.jl = c->d_jl;
  EArgs ea;
  ea.o = c->o; ea.ncell = c->ncell; ea.nhru = c->nhru; ea.Nn = Nn; ea.glist = ch->d_glist; ea.gcount = ch->gcount; ea.map = ch->map;
  ea.cell_params = c->d_cp; ea.hpi = c->d_hpi; ea.ctx = c->d_ctx;
  ea.ctx_words = n10 ? ctx_words<10>() : ctx_words<VIC_MAX_NODES>();
  ea.pout = c->d_pout; ea.pslot = c->d_pslot; ea.ts = c->d_ts; ea.hstate = c->d_hstate; ea.profile_next = ch->d_count + CNT_CURSOR;
  ea.list_thr = (int)((long long)ch->gcount * c->eval_list_pct / 100);
  ea.list_cap = ch->list_cap; ea.hkey = c->d_hkey; ea.implicit = c->o.IMPLICIT; ea.jl = c->d_jl;
  ea.veglib = c->d_veglib; ea.forcing = ka.forcing; ea.month = ka.dmy.month;
  const bool trace_rounds = getenv("VICGPU_TRACE_ROUNDS") != nullptr;
  const int FREE_ROUNDS = 6;       // a Brent solve needs two bracket evaluations, a few iterations and the final evaluation
  const int nsub = c->o.NF;
  for (int p = 1; p <= nsub; p++) {
    int nmax = ch->gcount;
    int npend = -1;                            // upper bound of the evaluations pending (solves + on-record finals), once known
    int rb_list[RB_DEPTH];                     // the list each read-back slot counts
    int rb_first = -1;                         // first round whose counts were read back
    for (int round = 0;; round++) {
      pa.list = ch->d_list[cur]; pa.count = cnt_list(ch->d_count, cur); pa.count_zero = cnt_list(ch->d_count, cur ^ 1);
      pa.evalonly_zero = cnt_evalonly(ch->d_count, cur ^ 1);
      pa.pend_counts = cnt_list(ch->d_count, cur); pa.pend_eo = cnt_evalonly(ch->d_count, cur); pa.pend_out = ch->d_count + CNT_NPEND;
      if (c->o.IMPLICIT) {
        // the Newton iteration for every listed HRU; those it fails for go on the fall-back list, which the explicit kernel
        // (the same one, on that list) solves right after (func_surf_energy_bal.c:192-222)
        CHKCH(ch, hipMemsetAsync(ch->d_fb_count, 0, sizeof(int) * (NBUCKET + 1), st));      // the fall-back segments and the work-list cursor
        IArgs ia;
        ia.ncell = c->ncell; ia.nhru = c->nhru; ia.Nband = c->o.Nband; ia.pimp = c->d_pimp; ia.hpi = c->d_hpi; ia.cell_params = c->d_cp;
        ia.hkey = c->d_hkey; ia.fb_list = ch->d_fb_list; ia.fb_count = ch->d_fb_count; ia.lastexp = c->d_lastexp; ia.cursor = ch->d_fb_count + NBUCKET;
        {
          // persistent waves, a few per SIMD: a lane takes the next solve when its own ends (vic_implicit.hpp)
          int nblk = (nmax + 63) / 64;
          if (nblk > 4096) nblk = 4096;
          if (nblk < 1) nblk = 1;
          hipLaunchKernelGGL(vic_profile_solve_implicit, dim3(nblk), dim3(64), 0, st, pa, ia);
        }
        CHKCH(ch, hipGetLastError());
        pa.list = ch->d_fb_list; pa.count = ch->d_fb_count;
      }
      CHKCH(ch, (n10 ? launch_profile<10>(pa, nmax, c->profile_waves, c->node_newton, st)
                     : launch_profile<VIC_MAX_NODES>(pa, nmax, c->profile_waves, c->node_newton, st)));
      ea.list_next = ch->d_list[cur ^ 1]; ea.count_next = cnt_list(ch->d_count, cur ^ 1);
      ea.evalonly = cnt_evalonly(ch->d_count, cur ^ 1); ea.eo_list_next = ch->d_elist[cur ^ 1];
      ea.list_cur = ch->d_list[cur]; ea.count_cur = cnt_list(ch->d_count, cur);
      ea.eo_list_cur = ch->d_elist[cur]; ea.npend_cur = ch->d_count + CNT_NPEND;
      // the device switches to the lists by itself; once the host knows (RB_LAG rounds late) that it has, the grid shrinks too
      const bool sparse = npend >= 0 && npend <= ea.list_thr;
      hipLaunchKernelGGL(vic_surf_eval, dim3(sparse ? ((npend + 63) / 64 > 0 ? (npend + 63) / 64 : 1) : ea.map.nblocks(ch->gcount)), dim3(64), 0, st, ea);
      CHKCH(ch, hipGetLastError());
      cur ^= 1;
      ch->rounds++;
      if (trace_rounds) {       // tuning: what every round leaves pending (a host round trip per round)
        int n = 0, ne = 0;
        if (fd_read_count(ch, cur, &n, &ne) != VICGPU_OK) return VICGPU_ERR_HIP;
        fprintf(stderr, "vicgpu rounds: chunk %d sub-step %d round %d leaves %d solves + %d evaluation-only of %d\n", (int)(ch - &c->chunks[0]), p, round, n, ne, ch->gcount);
      }
      // The list sizes of this round travel to the host behind the kernels just launched; the host looks at the copy issued
      // RB_LAG rounds ago, which has long arrived, so waiting for it never leaves the GPU idle.  The counts only shrink from
      // round to round (an HRU either goes on or is through), so a stale count is a valid upper bound for the grid.
      if (round + 2 >= FREE_ROUNDS) {
        const int slot = round % RB_DEPTH;
        if (rb_first < 0) rb_first = round;
        CHKCH(ch, hipMemcpyAsync(ch->h_count + slot * CNT_TOTAL, ch->d_count, sizeof(int) * CNT_TOTAL, hipMemcpyDeviceToHost, st));
        CHKCH(ch, hipEventRecord(ch->readback[slot], st));
        rb_list[slot] = cur;
      }
      if (rb_first >= 0 && round - RB_LAG >= rb_first) {
        const int slot = (round - RB_LAG) % RB_DEPTH;
        CHKCH(ch, hipEventSynchronize(ch->readback[slot]));
        const int* h = ch->h_count + slot * CNT_TOTAL;
        int n = 0;
        for (int b = 0; b < NBUCKET; b++) n += h[rb_list[slot] * CNT_LIST_STRIDE + b];
        const int neo = h[CNT_EVALONLY + rb_list[slot] * CNT_EVALONLY_STRIDE];
        if (n == 0 && neo == 0) break;
        nmax = n;
        npend = n + neo;
      }
    }
    ka.phase = p; ka.list = ch->d_list[cur]; ka.count = cnt_list(ch->d_count, cur);
    CHKCH(ch, (n10 ? launch_fd_stage<10>(ka, c->o.NF > 1, st) : launch_fd_stage<VIC_MAX_NODES>(ka, c->o.NF > 1, st)));
    if (p < nsub) {
      int n = 0, ne = 0;
      const int r = fd_read_count(ch, cur, &n, &ne);
      if (r != VICGPU_OK) return r;
      if (n == 0) break;
    }
  }
  ch->steps++;
  return VICGPU_OK;
}

// put_data for cells [c0, c0 + ccount) after step s of the forcing chunk (s < 0: the initialisation call)
static hipError_t launch_put_data(const vicgpu_ctx* c, hipStream_t st, int c0, int ccount, int s) {
  OArgs a;
  a.o = c->o; a.lay = c->out_lay; a.ncell = c->ncell; a.nhru = c->nhru; a.c0 = c0; a.ccount = ccount;
  a.rec = s < 0 ? -1 : 0; a.out_step_ratio = c->out_step_ratio;
  a.cell_off = c->d_cell_off; a.cell_list = c->d_cell_list; a.cell_params = c->d_cp; a.veglib = c->d_veglib;
  a.hpi = c->d_hpi; a.hpd = c->d_hpd; a.sd = c->d_sd; a.si = c->d_si; a.flux = c->d_flux;
  a.forcing = s < 0 ? nullptr : c->d_forcing + (size_t)s * VIC_NFORCE * (c->o.NR + 1) * c->ncell;
  a.cell_out = c->d_cell_out; a.out_data = c->d_out_data; a.out_agg = c->d_out_agg; a.pb = c->d_pb;
  const unsigned nblk = (unsigned)((ccount + 63) / 64);
  // zero_output_list: the columns of these cells in every row
  hipLaunchKernelGGL(vic_put_zero, dim3(nblk, (c->out_nrow + PUT_AGG_ROWS - 1) / PUT_AGG_ROWS), dim3(64), 0, st, a);
  hipLaunchKernelGGL(vic_put_sum, dim3(nblk, PUT_NPART), dim3(64), 0, st, a);
  hipLaunchKernelGGL(vic_put_finish, dim3(nblk), dim3(64), 0, st, a);
  if (s >= 0)
    hipLaunchKernelGGL(vic_put_aggregate, dim3(nblk, (c->out_nrow + PUT_AGG_ROWS - 1) / PUT_AGG_ROWS), dim3(64), 0, st, a, c->d_rowagg);
  return hipGetLastError();
}

struct StepPlan {
  vicgpu_ctx* c;
  KArgs ka;
  CArgs ca;
  int step0, nsteps;
};


static void set_step_inputs(const vicgpu_ctx* c, KArgs& ka, int s) {
  const size_t nsub = c->o.NR + 1;
  ka.forcing = c->d_forcing + (size_t)s * VIC_NFORCE * nsub * c->ncell;
  ka.snowflag = c->d_snowflag + (size_t)s * nsub * c->ncell;
  const int* d = &c->dmy[(size_t)s * VIC_NDMY];
  ka.dmy.month = d[VIC_DMY_MONTH]; ka.dmy.day_in_year = d[VIC_DMY_DAY_IN_YEAR]; ka.dmy.hour = d[VIC_DMY_HOUR];
  ka.dmy.day = d[VIC_DMY_DAY]; ka.dmy.year = d[VIC_DMY_YEAR];
}

// all steps of one vicgpu_step call for one chunk
static int fd_chunk_run(const StepPlan& plan, FdChunk* ch) {
  vicgpu_ctx* c = plan.c;
  CHKCH(ch, hipSetDevice(c->device));
  KArgs ka = plan.ka;
  CArgs ca = plan.ca;
  ka.glist = ch->d_glist; ka.gcount = ch->gcount; ka.map = ch->map;
  ca.c0 = ch->c0; ca.ccount = ch->ccount;
  const bool trace = getenv("VICGPU_TRACE") != nullptr;      // tuning: per-step wall time and Brent rounds (adds a sync per step)
  for (int s = plan.step0; s < plan.step0 + plan.nsteps; s++) {
    set_step_inputs(c, ka, s);
    const long long r0 = ch->rounds;
    const auto t0 = std::chrono::steady_clock::now();
    const int r = fd_step(c, ch, ka);
    if (r != VICGPU_OK) return r;
    if (trace) {
      CHKCH(ch, hipStreamSynchronize(ch->stream));
      fprintf(stderr, "[vicgpu] step %d hour %d: %.2f ms, %lld rounds\n", s, ka.dmy.hour,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ch->rounds - r0);
    }
    hipLaunchKernelGGL(vic_cell_reduce, dim3((ch->ccount + 255) / 256), dim3(256), 0, ch->stream, ca);
    CHKCH(ch, hipGetLastError());
    if (c->put_on) CHKCH(ch, launch_put_data(c, ch->stream, ch->c0, ch->ccount, s));
  }
  CHKCH(ch, hipEventRecord(ch->done, ch->stream));
  return VICGPU_OK;
}

extern "C" {

#ifdef VIC_PROF
// tuning build only (not part of include/vicgpu.h): read and clear the section counters
int vicgpu_prof_read(unsigned long long* cyc, unsigned long long* cnt) {
  if (hipMemcpyFromSymbol(cyc, HIP_SYMBOL(vic_prof_cyc), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(vic_prof_cnt), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  unsigned long long z[32] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(vic_prof_cyc), z, sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(vic_prof_cnt), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

int vicgpu_abi_version(void) { return VICGPU_ABI_VERSION; }

const char* vicgpu_last_error(const vicgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vicgpu_create(const vicgpu_options* opt, int device, vicgpu_ctx** out) {
  if (!opt || !out) return VICGPU_ERR_ARG;
  *out = nullptr;
  if (opt->abi_version != VICGPU_ABI_VERSION) return VICGPU_ERR_ARG;
  if (opt->Nlayer != VIC_NLAYER || opt->Nnode < 3 || opt->Nnode > VIC_MAX_NODES || opt->Nband < 1 || opt->Nband > VIC_MAX_BANDS)
    return VICGPU_ERR_UNSUPPORTED;
  if (opt->dt <= 0 || opt->snow_step <= 0 || opt->dt % opt->snow_step != 0) return VICGPU_ERR_ARG;
  if (opt->QUICK_FLUX && opt->Nnode != 3) return VICGPU_ERR_ARG;             // get_global_param.c:1151-1155
  if (opt->FROZEN_SOIL && opt->QUICK_FLUX) return VICGPU_ERR_ARG;            // get_global_param.c:376-381
  // options of the reference this library does not implement are refused, never silently replaced
  // QUICK_SOLVE (calc_surf_energy_bal.c:289-309, 400-480; ignored with QUICK_FLUX like in the reference): the reference forces
  // NOFLUX and EXP_TRANS off for the iteration and keeps whatever it last set for the final evaluation (NOFLUX returns with a
  // second iteration, EXP_TRANS never does): reproduced; not combined with IMPLICIT
  if (opt->QUICK_SOLVE && !opt->QUICK_FLUX && opt->IMPLICIT) return VICGPU_ERR_UNSUPPORTED;
  // IMPLICIT (newt_raph_func_fast.c): the finite-difference soil profile with the node freezing parameters of the node
  // arrays; the reference as shipped reads the 3-element layer arrays out of bounds there (frozen_soil.c:283-284)
  if (opt->IMPLICIT && (opt->QUICK_FLUX || opt->frozen_compat)) return VICGPU_ERR_UNSUPPORTED;
  if (opt->NODE_SOLVER != VIC_NODE_SOLVER_BRENT && opt->NODE_SOLVER != VIC_NODE_SOLVER_NEWTON) return VICGPU_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VICGPU_ERR_HIP;   // no CPU fallback: fail loudly
  if (device < 0 || device >= ndev) return VICGPU_ERR_ARG;
  vicgpu_ctx* c = new vicgpu_ctx();
  c->opt = *opt;
  c->device = device;
  Opt& o = c->o;
  o.Nnode = opt->Nnode; o.Nband = opt->Nband; o.dt = opt->dt; o.snow_step = opt->snow_step;
  o.NF = VICGPU_NF(opt); o.NR = VICGPU_NR(opt);
  o.FULL_ENERGY = opt->FULL_ENERGY; o.FROZEN_SOIL = opt->FROZEN_SOIL; o.QUICK_FLUX = opt->QUICK_FLUX; o.NOFLUX = opt->NOFLUX;
  o.EXP_TRANS = opt->EXP_TRANS; o.GRND_FLUX_TYPE = opt->GRND_FLUX_TYPE; o.TFALLBACK = opt->TFALLBACK;
  o.AERO_RESIST_CANSNOW = opt->AERO_RESIST_CANSNOW; o.SNOW_ALBEDO = opt->SNOW_ALBEDO; o.SNOW_DENSITY = opt->SNOW_DENSITY;
  o.TEMP_TH_TYPE = opt->TEMP_TH_TYPE; o.GLACIER_ID = opt->GLACIER_ID; o.GLACIER_DYNAMICS = opt->GLACIER_DYNAMICS;
  o.frozen_compat = opt->frozen_compat; o.nveg_types = opt->nveg_types; o.wind_h = opt->wind_h; o.CORRPREC = opt->CORRPREC;
  o.BLOWING = opt->BLOWING ? 1 : 0; o.IMPLICIT = opt->IMPLICIT; o.QUICK_SOLVE = (opt->QUICK_SOLVE && !opt->QUICK_FLUX) ? 1 : 0;
  // calc_surf_energy_bal.c:300-308: with QUICK_SOLVE and a surface energy balance the solver's EXP_TRANS is FALSE from the first
  // iteration to the final evaluation (the linear-spacing coefficients on the run's node geometry, whatever it is)
  if (o.QUICK_SOLVE && o.FULL_ENERGY) o.EXP_TRANS = 0;
  if (hipSetDevice(device) != hipSuccess) { delete c; return VICGPU_ERR_HIP; }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess
      || hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess
      || hipEventCreateWithFlags(&c->slot[0].uploaded, hipEventDisableTiming) != hipSuccess
      || hipEventCreateWithFlags(&c->slot[1].uploaded, hipEventDisableTiming) != hipSuccess
      || hipEventCreateWithFlags(&c->slot[0].released, hipEventDisableTiming) != hipSuccess
      || hipEventCreateWithFlags(&c->slot[1].released, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return VICGPU_ERR_HIP;
  }
  *out = c;
  return VICGPU_OK;
}

void vicgpu_destroy(vicgpu_ctx* c) {
  if (!c) return;
  HIPIGN(hipSetDevice(c->device));
  if (c->stream) HIPIGN(hipStreamSynchronize(c->stream));
  if (getenv("VICGPU_STATS"))
    for (size_t k = 0; k < c->chunks.size(); k++)
      if (c->chunks[k].steps)
        fprintf(stderr, "[vicgpu] chunk %zu: %d cells, %d HRUs, %lld steps, %.1f Brent rounds per step\n", k, c->chunks[k].ccount,
                c->chunks[k].gcount, c->chunks[k].steps, (double)c->chunks[k].rounds / c->chunks[k].steps);
  free_domain(c);
  HIPIGN(hipFree(c->d_veglib));
  if (c->copy_stream) HIPIGN(hipStreamSynchronize(c->copy_stream));
  for (auto& sl : c->slot) {
    HIPIGN(hipFree(sl.d_f)); HIPIGN(hipFree(sl.d_s)); HIPIGN(hipFree(sl.d_raw));
    if (sl.h_stage) HIPIGN(hipHostFree(sl.h_stage));
    if (sl.uploaded) HIPIGN(hipEventDestroy(sl.uploaded));
    if (sl.released) HIPIGN(hipEventDestroy(sl.released));
  }
  for (auto e : c->ev) HIPIGN(hipEventDestroy(e));
  if (c->own_stream && c->stream) HIPIGN(hipStreamDestroy(c->stream));
  if (c->copy_stream) HIPIGN(hipStreamDestroy(c->copy_stream));
  delete c;
}

int vicgpu_set_veglib(vicgpu_ctx* c, int nrow, const double* veglib) {
  if (!c || !veglib || nrow != c->opt.nveg_types + 4) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPIGN(hipFree(c->d_veglib));
  c->d_veglib = nullptr;
  HIPCHK(c, hipMalloc(&c->d_veglib, sizeof(double) * nrow * VL_NFIELD));
  HIPCHK(c, copy_on(c->stream, c->d_veglib, veglib, sizeof(double) * nrow * VL_NFIELD, hipMemcpyHostToDevice));
  c->nveg_rows = nrow;
  return VICGPU_OK;
}

static int set_domain_impl(vicgpu_ctx* c, int ncell, int nhru, const double* cell_params, const int* hpi, const double* hpd,
                           const int* cell_hru_offset, const int* cell_hru_list);

int vicgpu_set_domain(vicgpu_ctx* c, int ncell, int nhru, const double* cell_params, const int* hpi, const double* hpd,
                      const int* cell_hru_offset, const int* cell_hru_list) {
  if (!c) return VICGPU_ERR_ARG;
  const int r = set_domain_impl(c, ncell, nhru, cell_params, hpi, hpd, cell_hru_offset, cell_hru_list);
  if (r == VICGPU_OK) c->domain_ready = true;
  else if (r == VICGPU_ERR_HIP || r == VICGPU_ERR_NOMEM) {     // failed half-way: leave no partially built domain behind
    HIPIGN(hipSetDevice(c->device));
    free_domain(c);
  }
  return r;
}

static int set_domain_impl(vicgpu_ctx* c, int ncell, int nhru, const double* cell_params, const int* hpi, const double* hpd,
                           const int* cell_hru_offset, const int* cell_hru_list) {
  if (!c || ncell <= 0 || nhru <= 0 || !cell_params || !hpi || !hpd || !cell_hru_offset || !cell_hru_list) return VICGPU_ERR_ARG;
  // host-side shape checks: every index the kernels dereference is validated here, once
  if (cell_hru_offset[0] != 0 || cell_hru_offset[ncell] != nhru) return VICGPU_ERR_ARG;
  for (int i = 0; i < ncell; i++) if (cell_hru_offset[i + 1] < cell_hru_offset[i]) return VICGPU_ERR_ARG;
  {
    std::vector<char> seen(nhru, 0);
    for (int i = 0; i < ncell; i++)
      for (int k = cell_hru_offset[i]; k < cell_hru_offset[i + 1]; k++) {
        int g = cell_hru_list[k];
        if (g < 0 || g >= nhru || seen[g] || hpi[(size_t)HPI_CELL * nhru + g] != i) return VICGPU_ERR_ARG;
        seen[g] = 1;
      }
    for (int g = 0; g < nhru; g++) {
      if (!seen[g]) return VICGPU_ERR_ARG;
      int b = hpi[(size_t)HPI_BAND * nhru + g], v = hpi[(size_t)HPI_VEG_INDEX * nhru + g];
      if (b < 0 || b >= c->opt.Nband || v < 0 || v >= c->opt.nveg_types + 4) return VICGPU_ERR_ARG;
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  free_domain(c);
  c->ncell = ncell; c->nhru = nhru;
  c->any_glacier = false;
  for (int g = 0; g < nhru; g++) if (hpi[(size_t)HPI_IS_GLACIER * nhru + g]) c->any_glacier = true;
  const size_t cp_n = (size_t)VICGPU_CP_NROW(c->opt.Nnode, c->opt.Nband) * ncell;
  const size_t cpx_n = (size_t)VIC_CPX_NROW(c->opt.Nnode, c->opt.Nband) * ncell;      // + the derived rows
  const size_t sd_n = (size_t)VICGPU_SD_NROW(c->opt.Nnode) * nhru, si_n = (size_t)VICGPU_SI_NROW(c->opt.Nnode) * nhru;
  HIPCHK(c, hipMalloc(&c->d_cp, sizeof(double) * cpx_n));
  HIPCHK(c, hipMalloc(&c->d_hpi, sizeof(int) * HPI_NROW * nhru));
  HIPCHK(c, hipMalloc(&c->d_hpd, sizeof(double) * HPD_NROW * nhru));
  HIPCHK(c, hipMalloc(&c->d_cell_off, sizeof(int) * (ncell + 1)));
  HIPCHK(c, hipMalloc(&c->d_cell_list, sizeof(int) * nhru));
  HIPCHK(c, hipMalloc(&c->d_sd, sizeof(double) * sd_n));
  HIPCHK(c, hipMalloc(&c->d_si, sizeof(int) * si_n));
  HIPCHK(c, hipMalloc(&c->d_flux, sizeof(double) * FX_NROW * nhru));
  HIPCHK(c, hipMalloc(&c->d_cell_out, sizeof(double) * CO_NROW * ncell));
  HIPCHK(c, hipMalloc(&c->d_accum, sizeof(double) * CA_NROW * ncell));
  HIPCHK(c, hipMalloc(&c->d_hru_err, sizeof(int) * nhru));
  HIPCHK(c, hipMalloc(&c->d_cell_err, sizeof(int) * ncell));
  HIPCHK(c, copy_on(c->stream, c->d_cp, cell_params, sizeof(double) * cp_n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(vic_derive_cell_params, dim3((ncell + 255) / 256), dim3(256), 0, c->stream, c->d_cp, ncell, c->opt.Nnode, c->opt.Nband);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, c->d_hpi, hpi, sizeof(int) * HPI_NROW * nhru, hipMemcpyHostToDevice));
  HIPCHK(c, copy_on(c->stream, c->d_hpd, hpd, sizeof(double) * HPD_NROW * nhru, hipMemcpyHostToDevice));
  HIPCHK(c, copy_on(c->stream, c->d_cell_off, cell_hru_offset, sizeof(int) * (ncell + 1), hipMemcpyHostToDevice));
  HIPCHK(c, copy_on(c->stream, c->d_cell_list, cell_hru_list, sizeof(int) * nhru, hipMemcpyHostToDevice));
  HIPCHK(c, fill_on(c->stream, c->d_sd, 0, sizeof(double) * sd_n));
  HIPCHK(c, fill_on(c->stream, c->d_si, 0, sizeof(int) * si_n));
  HIPCHK(c, fill_on(c->stream, c->d_flux, 0, sizeof(double) * FX_NROW * nhru));
  HIPCHK(c, fill_on(c->stream, c->d_cell_out, 0, sizeof(double) * CO_NROW * ncell));
  HIPCHK(c, fill_on(c->stream, c->d_accum, 0, sizeof(double) * CA_NROW * ncell));
  HIPCHK(c, fill_on(c->stream, c->d_hru_err, 0, sizeof(int) * nhru));
  HIPCHK(c, fill_on(c->stream, c->d_cell_err, 0, sizeof(int) * ncell));
  c->fd = !c->o.QUICK_FLUX;
  if (c->fd) {
    const int Nn = c->o.Nnode;
    const size_t words = (Nn == 10) ? ctx_words<10>() : ctx_words<VIC_MAX_NODES>();
    HIPCHK(c, hipMalloc(&c->d_ctx, sizeof(unsigned long long) * ctx_padded_words(words) * (((size_t)nhru + 63) / 64 * 64)));
    HIPCHK(c, hipMalloc(&c->d_pin, sizeof(double) * (size_t)Nn * PREC * nhru));
    HIPCHK(c, hipMalloc(&c->d_ts, sizeof(double) * nhru));
    HIPCHK(c, hipMalloc(&c->d_pout, sizeof(double) * (size_t)pout_hru_stride(Nn) * nhru));
    HIPCHK(c, hipMalloc(&c->d_pslot, sizeof(int) * nhru));
    HIPCHK(c, hipMalloc(&c->d_hkey, sizeof(int) * nhru));
    HIPCHK(c, fill_on(c->stream, c->d_hkey, 0, sizeof(int) * nhru));
    HIPCHK(c, fill_on(c->stream, c->d_pslot, 0, sizeof(int) * nhru));
    HIPCHK(c, hipMalloc(&c->d_hstate, sizeof(int) * nhru));
    HIPCHK(c, fill_on(c->stream, c->d_hstate, 0, sizeof(int) * nhru));
    HIPCHK(c, fill_on(c->stream, c->d_pin, 0, sizeof(double) * (size_t)Nn * PREC * nhru));
    HIPCHK(c, fill_on(c->stream, c->d_pout, 0, sizeof(double) * (size_t)pout_hru_stride(Nn) * nhru));
    if (c->o.QUICK_SOLVE) {
      HIPCHK(c, hipMalloc(&c->d_jl, sizeof(int) * nhru));
      HIPCHK(c, fill_on(c->stream, c->d_jl, 0, sizeof(int) * nhru));
    }
    if (c->o.IMPLICIT) {
      HIPCHK(c, hipMalloc(&c->d_pimp, sizeof(double) * (size_t)Nn * PIMP * nhru));
      HIPCHK(c, hipMalloc(&c->d_lastexp, sizeof(int) * nhru));
      HIPCHK(c, fill_on(c->stream, c->d_pimp, 0, sizeof(double) * (size_t)Nn * PIMP * nhru));
      HIPCHK(c, fill_on(c->stream, c->d_lastexp, 0xFF, sizeof(int) * nhru));
    }
    // frozen-node root finder (vic_profile.hpp): the option, overridable for A/B runs
    c->node_newton = c->opt.NODE_SOLVER == VIC_NODE_SOLVER_NEWTON;
    if (const char* ev = getenv("VICGPU_NODE_SOLVER")) c->node_newton = (strcmp(ev, "newton") == 0);
    c->profile_waves = (Nn == 10) ? profile_resident_waves<10>(c->device, c->node_newton)
                                  : profile_resident_waves<VIC_MAX_NODES>(c->device, c->node_newton);
    // tuning: the pending share (percent of the chunk's HRUs) from which the evaluation rounds run from the pending list; 0 = never
    if (const char* ev = getenv("VICGPU_EVAL_LIST_PCT")) {
      const int pct = atoi(ev);
      if (pct >= 0 && pct <= 100) c->eval_list_pct = pct;
    }
    // cell chunks (VICGPU_CHUNKS): independent pipelines on their own streams and host threads.  Every kernel of the pipeline
    // is stalled most of its time (dependent fp64 chains in the profile kernel, memory latency in the others: 15 % VALU-active
    // per wave), so two pipelines side by side fill each other's gaps and thin tail rounds: -6 % step time at 2.5 M HRUs
    // (27.3 vs 29.0 ms, same-box A/B); three or more lose again.  Default: two chunks for domains of 20k cells or more
    // (VICGPU_CHUNKS=1 gives per-kernel profiles whose durations add up to the step).
    int nchunk = (ncell >= 20000) ? 2 : 1;
    if (const char* ev = getenv("VICGPU_CHUNKS")) nchunk = atoi(ev);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 16) nchunk = 16;
    if (nchunk > ncell) nchunk = ncell;
    // A chunk's profile kernel takes half of the resident wave slots when chunks run side by side, so that the other chunk's
    // kernels find free SIMD slots beside it (26.5 vs 26.9 ms per step with two chunks, same box, both repetitions);
    // VICGPU_PROFILE_WAVES_PCT overrides (tuning)
    int waves_pct = nchunk > 1 ? 50 : 100;
    if (const char* ev = getenv("VICGPU_PROFILE_WAVES_PCT")) {
      const int pct = atoi(ev);
      if (pct >= 5 && pct <= 100) waves_pct = pct;
    }
    c->profile_waves = c->profile_waves * waves_pct / 100 > 0 ? c->profile_waves * waves_pct / 100 : 1;
    c->chunks.resize(nchunk);
    for (int k = 0; k < nchunk; k++) {
      FdChunk& ch = c->chunks[k];
      ch.c0 = (int)((long long)ncell * k / nchunk);
      ch.ccount = (int)((long long)ncell * (k + 1) / nchunk) - ch.c0;
      std::vector<int> gl(cell_hru_list + cell_hru_offset[ch.c0], cell_hru_list + cell_hru_offset[ch.c0 + ch.ccount]);
      std::sort(gl.begin(), gl.end());
      ch.gcount = (int)gl.size();
      ch.map = LaunchMap();
      if (ch.ccount > 0 && ch.gcount % ch.ccount == 0 && !getenv("VICGPU_NO_XCD_MAP")) {
        const int nslot = ch.gcount / ch.ccount;
        bool regular = true;
        for (int sl = 0; sl < nslot && regular; sl++)
          for (int i = 0; i < ch.ccount; i++)
            if (gl[(size_t)sl * ch.ccount + i] != sl * ncell + ch.c0 + i || hpi[(size_t)HPI_CELL * nhru + gl[(size_t)sl * ch.ccount + i]] != ch.c0 + i) {
              regular = false;
              break;
            }
        if (regular) { ch.map.nslot = nslot; ch.map.ccount = ch.ccount; }
      }
      const size_t gb = sizeof(int) * (size_t)(ch.gcount > 0 ? ch.gcount : 1);
      HIPCHK(c, hipMalloc(&ch.d_glist, gb));
      ch.list_cap = ch.gcount > 0 ? ch.gcount : 1;
      HIPCHK(c, hipMalloc(&ch.d_list[0], gb * NBUCKET));
      HIPCHK(c, hipMalloc(&ch.d_list[1], gb * NBUCKET));
      HIPCHK(c, hipMalloc(&ch.d_count, sizeof(int) * CNT_TOTAL));
      HIPCHK(c, hipMalloc(&ch.d_elist[0], gb));
      HIPCHK(c, hipMalloc(&ch.d_elist[1], gb));
      if (c->o.IMPLICIT) {
        HIPCHK(c, hipMalloc(&ch.d_fb_list, gb * NBUCKET));
        HIPCHK(c, hipMalloc(&ch.d_fb_count, sizeof(int) * (NBUCKET + 1)));
      }
      HIPCHK(c, hipHostMalloc(&ch.h_count, sizeof(int) * CNT_TOTAL * RB_DEPTH, hipHostMallocDefault));
      if (ch.gcount) HIPCHK(c, copy_on(c->stream, ch.d_glist, gl.data(), sizeof(int) * ch.gcount, hipMemcpyHostToDevice));
      HIPCHK(c, hipStreamCreateWithFlags(&ch.stream, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&ch.done, hipEventDisableTiming));
      for (hipEvent_t& e : ch.readback) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
  }
  return VICGPU_OK;
}

int vicgpu_set_state(vicgpu_ctx* c, const double* sd, const int* si) {
  if (!c || !sd || !si) return VICGPU_ERR_ARG;
  if (!c->domain_ready) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, c->d_sd, sd, sizeof(double) * VICGPU_SD_NROW(c->opt.Nnode) * c->nhru, hipMemcpyHostToDevice));
  HIPCHK(c, copy_on(c->stream, c->d_si, si, sizeof(int) * VICGPU_SI_NROW(c->opt.Nnode) * c->nhru, hipMemcpyHostToDevice));
  return VICGPU_OK;
}

int vicgpu_get_state(vicgpu_ctx* c, double* sd, int* si) {
  if (!c || !sd || !si) return VICGPU_ERR_ARG;
  if (!c->domain_ready) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, sd, c->d_sd, sizeof(double) * VICGPU_SD_NROW(c->opt.Nnode) * c->nhru, hipMemcpyDeviceToHost));
  HIPCHK(c, copy_on(c->stream, si, c->d_si, sizeof(int) * VICGPU_SI_NROW(c->opt.Nnode) * c->nhru, hipMemcpyDeviceToHost));
  return VICGPU_OK;
}

// initialize_atmos.c, the derivation of atmos[rec] from the hourly forcing of one record (see include/vicgpu.h): one lane
// per (step, cell)
struct FArgs {
  int nsteps, ncell, dt, snow_step, NF, NR, temp_th_type, Nband, Nnode, plapse;
  double min_wind;
  const double* raw;
  const double* cell_params;
  double* forcing;
  unsigned char* snowflag;
};

__global__ __launch_bounds__(256) void vic_derive_forcing(const FArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)a.nsteps * a.ncell) return;
  const int s = (int)(i / a.ncell), c = (int)(i % a.ncell);
  const size_t nc = a.ncell;
  const int ns = a.NR + 1, NF = a.NF;
  const double* raw = a.raw + (size_t)s * VIC_NRAW * a.dt * nc + c;
  double* f = a.forcing + (size_t)s * VIC_NFORCE * ns * nc + c;
  unsigned char* sf = a.snowflag + (size_t)s * ns * nc + c;
#define RAW(v, h) raw[((size_t)(v) * a.dt + (h)) * nc]
#define F(v, j) f[((size_t)(v) * ns + (j)) * nc]
  CellView cv{a.cell_params, a.ncell, c, a.Nnode, a.Nband};
  double min_Tfactor = cv.band(CPB_TFACTOR, 0);                                       // initialize_atmos.c:1275-1280
  for (int b = 1; b < a.Nband; b++) { const double t = cv.band(CPB_TFACTOR, b); if (t < min_Tfactor) min_Tfactor = t; }
  const double max_snow = cv.s(CP_MAX_SNOW_TEMP), min_rain = cv.s(CP_MIN_RAIN_TEMP);
  const double thr = (a.temp_th_type == VIC_TEMP_TH_KIENZLE) ? (max_snow + min_rain / 2) : max_snow;
  double sT = 0, sP = 0, sPr = 0, sVp = 0, sVpd = 0, sD = 0, sSw = 0, sLw = 0, sW = 0;
  bool any_snow = false;
  for (int j = 0; j < NF; j++) {
    double T = 0, prec = 0, pr = 0, vp = 0, sw = 0, lw = 0, wind = 0;
    for (int h = j * a.snow_step; h < (j + 1) * a.snow_step; h++) {                   // the snow_step-hour aggregation (:886-893 et al.)
      T += RAW(VIC_RAW_AIR_TEMP, h); prec += RAW(VIC_RAW_PREC, h);
      pr += RAW(VIC_RAW_PRESSURE_KPA, h) * 1000.0; vp += RAW(VIC_RAW_VP_KPA, h) * 1000.0;      // kPa2Pa, :290-295
      sw += RAW(VIC_RAW_SHORTWAVE, h); lw += RAW(VIC_RAW_LONGWAVE, h);
      const double w = RAW(VIC_RAW_WIND, h);
      wind += (w < a.min_wind) ? a.min_wind : w;                                      // :527-530
    }
    T /= a.snow_step; pr /= a.snow_step; vp /= a.snow_step; sw /= a.snow_step; lw /= a.snow_step; wind /= a.snow_step;
    const double dens = a.plapse ? pr / (287.0 * (KELVIN + T)) : 0.003486 * pr / (275.0 + T);   // :988-998 (Rd = 287)
    double vpd = svp(T) - vp;                                                         // :1179-1183
    if (vpd < 0) { vpd = 0; vp = svp(T); }
    F(VIC_F_AIR_TEMP, j) = T; F(VIC_F_PREC, j) = prec; F(VIC_F_PRESSURE, j) = pr; F(VIC_F_VP, j) = vp; F(VIC_F_VPD, j) = vpd;
    F(VIC_F_DENSITY, j) = dens; F(VIC_F_SHORTWAVE, j) = sw; F(VIC_F_LONGWAVE, j) = lw; F(VIC_F_WIND, j) = wind;
    const bool snow = ((T + min_Tfactor) < thr) && (prec > 0);                        // :1283-1300
    sf[(size_t)j * nc] = snow ? 1 : 0;
    any_snow = any_snow || snow;
    sT += T; sP += prec; sPr += pr; sVp += vp; sVpd += vpd; sD += dens; sSw += sw; sLw += lw; sW += wind;
  }
  if (NF > 1) {                                                                       // x[NR] = sum / (float)NF; prec[NR] = sum
    const double n = (double)(float)NF;
    F(VIC_F_AIR_TEMP, a.NR) = sT / n; F(VIC_F_PREC, a.NR) = sP; F(VIC_F_PRESSURE, a.NR) = sPr / n; F(VIC_F_VP, a.NR) = sVp / n;
    F(VIC_F_VPD, a.NR) = sVpd / n; F(VIC_F_SHORTWAVE, a.NR) = sSw / n; F(VIC_F_LONGWAVE, a.NR) = sLw / n;
    F(VIC_F_WIND, a.NR) = sW / n;
    // density[NR] is derived from pressure[NR] and air_temp[NR] like every other slot (initialize_atmos.c:984-998), not averaged
    F(VIC_F_DENSITY, a.NR) = a.plapse ? (sPr / n) / (287.0 * (KELVIN + sT / n)) : 0.003486 * (sPr / n) / (275.0 + sT / n);
    sf[(size_t)a.NR * nc] = any_snow ? 1 : 0;
  }
#undef RAW
#undef F
}

static bool is_pinned(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { HIPIGN(hipGetLastError()); return false; }
  return at.type == hipMemoryTypeHost;
}

// source -> device on the copy stream: straight from pinned memory, through the slot's staging area otherwise
static hipError_t upload(vicgpu_ctx* c, vicgpu_ctx::ForcingSlot& sl, void* dst, const void* src, size_t bytes, size_t stage_off) {
  const void* from = src;
  if (!is_pinned(src)) {
    memcpy((char*)sl.h_stage + stage_off, src, bytes);
    from = (char*)sl.h_stage + stage_off;
  }
  return hipMemcpyAsync(dst, from, bytes, hipMemcpyHostToDevice, c->copy_stream);
}

static int prefetch_impl(vicgpu_ctx* c, int nsteps, const double* forcing, const unsigned char* snowflag, const double* raw,
                         const int* dmy, double min_wind, int plapse) {
  if (!c || nsteps <= 0 || !dmy || (!raw && (!forcing || !snowflag))) return VICGPU_ERR_ARG;
  if (!c->domain_ready) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  for (int s = 0; s < nsteps; s++) {
    int m = dmy[(size_t)s * VIC_NDMY + VIC_DMY_MONTH];
    if (m < 1 || m > 12) return VICGPU_ERR_ARG;          // month indexes the veg library tables
  }
  const size_t nsub = c->o.NR + 1;
  const size_t fbytes = sizeof(double) * (size_t)nsteps * VIC_NFORCE * nsub * c->ncell;
  const size_t sbytes = (size_t)nsteps * nsub * c->ncell;
  const size_t rbytes = raw ? sizeof(double) * (size_t)nsteps * VIC_NRAW * c->o.dt * c->ncell : 0;
  const int t = (c->cur == 0) ? 1 : 0;
  vicgpu_ctx::ForcingSlot& sl = c->slot[t];
  // the slot's previous upload may still be reading its staging area; the steps that read the slot's device buffers were
  // queued before `released` was recorded (vicgpu_swap_forcing): the copy stream waits for that, not the host
  if (sl.upload_pending) { HIPCHK(c, hipEventSynchronize(sl.uploaded)); sl.upload_pending = false; }
  if (sl.was_current) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, sl.released, 0));
  const bool grow = fbytes > sl.fcap || sbytes > sl.scap || rbytes > sl.rawcap;
  if (grow) {                                            // re-allocation: nothing may still use the old buffers
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    if (sl.was_current) HIPCHK(c, hipEventSynchronize(sl.released));
    if (fbytes > sl.fcap) { HIPIGN(hipFree(sl.d_f)); sl.d_f = nullptr; sl.fcap = 0; HIPCHK(c, hipMalloc(&sl.d_f, fbytes)); sl.fcap = fbytes; }
    if (sbytes > sl.scap) { HIPIGN(hipFree(sl.d_s)); sl.d_s = nullptr; sl.scap = 0; HIPCHK(c, hipMalloc(&sl.d_s, sbytes)); sl.scap = sbytes; }
    if (rbytes > sl.rawcap) { HIPIGN(hipFree(sl.d_raw)); sl.d_raw = nullptr; sl.rawcap = 0; HIPCHK(c, hipMalloc(&sl.d_raw, rbytes)); sl.rawcap = rbytes; }
  }
  const size_t need_stage = raw ? (is_pinned(raw) ? 0 : rbytes) : ((is_pinned(forcing) ? 0 : fbytes) + (is_pinned(snowflag) ? 0 : sbytes));
  if (need_stage > sl.stage_cap) {
    if (sl.h_stage) HIPIGN(hipHostFree(sl.h_stage));
    sl.h_stage = nullptr; sl.stage_cap = 0;
    HIPCHK(c, hipHostMalloc(&sl.h_stage, need_stage, hipHostMallocDefault));
    sl.stage_cap = need_stage;
  }
  if (raw) {
    HIPCHK(c, upload(c, sl, sl.d_raw, raw, rbytes, 0));
    FArgs a;
    a.nsteps = nsteps; a.ncell = c->ncell; a.dt = c->o.dt; a.snow_step = c->o.snow_step; a.NF = c->o.NF; a.NR = c->o.NR;
    a.temp_th_type = c->o.TEMP_TH_TYPE; a.Nband = c->o.Nband; a.Nnode = c->o.Nnode; a.plapse = plapse;
    a.min_wind = (double)(float)min_wind;        // options.MIN_WIND_SPEED is a float (vicNl_def.h:713)
    a.raw = sl.d_raw; a.cell_params = c->d_cp; a.forcing = sl.d_f; a.snowflag = sl.d_s;
    const size_t n = (size_t)nsteps * c->ncell;
    hipLaunchKernelGGL(vic_derive_forcing, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->copy_stream, a);
    HIPCHK(c, hipGetLastError());
  } else {
    HIPCHK(c, upload(c, sl, sl.d_f, forcing, fbytes, 0));
    HIPCHK(c, upload(c, sl, sl.d_s, snowflag, sbytes, is_pinned(forcing) ? 0 : fbytes));
  }
  HIPCHK(c, hipEventRecord(sl.uploaded, c->copy_stream));
  sl.upload_pending = true;
  sl.was_current = false;
  sl.dmy.assign(dmy, dmy + (size_t)nsteps * VIC_NDMY);
  sl.nsteps = nsteps;
  c->staged = t;
  return VICGPU_OK;
}

int vicgpu_prefetch_forcing(vicgpu_ctx* c, int nsteps, const double* forcing, const unsigned char* snowflag, const int* dmy) {
  return prefetch_impl(c, nsteps, forcing, snowflag, nullptr, dmy, 0.0, 1);
}
int vicgpu_prefetch_forcing_raw(vicgpu_ctx* c, int nsteps, const double* raw, const int* dmy, double min_wind_speed, int plapse) {
  if (!raw) return VICGPU_ERR_ARG;
  return prefetch_impl(c, nsteps, nullptr, nullptr, raw, dmy, min_wind_speed, plapse);
}

int vicgpu_swap_forcing(vicgpu_ctx* c) {
  if (!c) return VICGPU_ERR_ARG;
  if (c->staged < 0) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->cur >= 0) {                                      // every step queued so far read the old chunk: fence it
    HIPCHK(c, hipEventRecord(c->slot[c->cur].released, c->stream));
    c->slot[c->cur].was_current = true;
  }
  vicgpu_ctx::ForcingSlot& sl = c->slot[c->staged];
  // the source buffer (pinned user memory or the staging area) is free again once the upload has finished
  HIPCHK(c, hipEventSynchronize(sl.uploaded));
  sl.upload_pending = false;
  c->cur = c->staged; c->staged = -1;
  c->d_forcing = sl.d_f; c->d_snowflag = sl.d_s; c->dmy = sl.dmy; c->chunk_steps = sl.nsteps;
  return VICGPU_OK;
}

int vicgpu_push_forcing(vicgpu_ctx* c, int nsteps, const double* forcing, const unsigned char* snowflag, const int* dmy) {
  const int r = vicgpu_prefetch_forcing(c, nsteps, forcing, snowflag, dmy);
  return r == VICGPU_OK ? vicgpu_swap_forcing(c) : r;
}

int vicgpu_get_forcing(vicgpu_ctx* c, int step, double* forcing, unsigned char* snowflag) {
  if (!c || !forcing || !snowflag) return VICGPU_ERR_ARG;
  if (c->cur < 0 || step < 0 || step >= c->chunk_steps) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nsub = c->o.NR + 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, forcing, c->d_forcing + (size_t)step * VIC_NFORCE * nsub * c->ncell, sizeof(double) * VIC_NFORCE * nsub * c->ncell, hipMemcpyDeviceToHost));
  HIPCHK(c, copy_on(c->stream, snowflag, c->d_snowflag + (size_t)step * nsub * c->ncell, nsub * c->ncell, hipMemcpyDeviceToHost));
  return VICGPU_OK;
}

void* vicgpu_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void vicgpu_host_free(void* p) { if (p) HIPIGN(hipHostFree(p)); }

int vicgpu_step(vicgpu_ctx* c, int step0, int nsteps) {
  if (!c) return VICGPU_ERR_ARG;
  if (!c->domain_ready || !c->d_veglib || !c->d_forcing || c->chunk_steps <= 0) return VICGPU_ERR_STATE;
  if (step0 < 0 || nsteps <= 0 || step0 + nsteps > c->chunk_steps) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  while ((int)c->ev.size() < 2 * nsteps) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreate(&e));
    c->ev.push_back(e);
  }
  c->ev_used = 0;
  c->ev_steps = 0;
  StepPlan plan;
  plan.c = c; plan.step0 = step0; plan.nsteps = nsteps;
  KArgs& ka = plan.ka;
  ka.o = c->o; ka.ncell = c->ncell; ka.nhru = c->nhru; ka.nveg_rows = c->nveg_rows;
  ka.write_fluxes = (c->write_fluxes || c->put_on) ? 1 : 0;      // put_data reads every row of the flux table
  ka.veglib = c->d_veglib; ka.cell_params = c->d_cp; ka.hpi = c->d_hpi; ka.hpd = c->d_hpd;
  ka.sd = c->d_sd; ka.si = c->d_si; ka.flux = c->d_flux; ka.hru_err = c->d_hru_err;
  ka.glist = nullptr; ka.gcount = c->nhru;
  ka.ctx = c->d_ctx; ka.pin = c->d_pin; ka.ts = c->d_ts; ka.pout = c->d_pout; ka.pslot = c->d_pslot; ka.hstate = c->d_hstate; ka.hkey = c->d_hkey; ka.pimp = c->d_pimp; ka.lastexp = c->d_lastexp; ka.jl = c->d_jl; ka.list = nullptr; ka.count = nullptr; ka.list_cap = 0;
  ka.phase = 0;
  CArgs& ca = plan.ca;
  ca.ncell = c->ncell; ca.nhru = c->nhru; ca.c0 = 0; ca.ccount = c->ncell;
  ca.cell_off = c->d_cell_off; ca.cell_list = c->d_cell_list; ca.hpd = c->d_hpd;
  ca.hpi_glac = c->d_hpi + (size_t)HPI_IS_GLACIER * c->nhru;
  ca.flux = c->d_flux; ca.sd = c->d_sd; ca.hru_err = c->d_hru_err; ca.cell_out = c->d_cell_out; ca.accum = c->d_accum;
  ca.cell_err = c->d_cell_err;
  if (!c->fd) {
    // QUICK_FLUX (implies Nnode == 3, vicgpu_create): one kernel per step, enqueued without blocking
    for (int s = step0; s < step0 + nsteps; s++) {
      set_step_inputs(c, ka, s);
      HIPCHK(c, hipEventRecord(c->ev[2 * (s - step0)], c->stream));
      HIPCHK(c, launch_hru<3>(ka, c->stream, true, c->any_glacier));
      HIPCHK(c, hipEventRecord(c->ev[2 * (s - step0) + 1], c->stream));
      hipLaunchKernelGGL(vic_cell_reduce, dim3((c->ncell + 255) / 256), dim3(256), 0, c->stream, ca);
      HIPCHK(c, hipGetLastError());
      if (c->put_on) HIPCHK(c, launch_put_data(c, c->stream, 0, c->ncell, s));
      c->steps_done++;
    }
    c->ev_used = nsteps;
    c->ev_steps = nsteps;
    return VICGPU_OK;
  }
  // finite-difference pipeline: every chunk runs all nsteps on its own stream (ordered after what is queued on the
  // context's stream, which in turn waits for every chunk before anything queued later)
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  for (FdChunk& ch : c->chunks) {
    HIPCHK(c, hipStreamWaitEvent(ch.stream, c->ev[0], 0));
    ch.status = VICGPU_OK;
    ch.err.clear();
  }
  if (c->chunks.size() == 1) c->chunks[0].status = fd_chunk_run(plan, &c->chunks[0]);
  else {
    std::vector<std::thread> th;
    for (FdChunk& ch : c->chunks) th.emplace_back([&plan, &ch]() { ch.status = fd_chunk_run(plan, &ch); });
    for (std::thread& t : th) t.join();
  }
  int status = VICGPU_OK;
  for (FdChunk& ch : c->chunks) {
    if (ch.status != VICGPU_OK && status == VICGPU_OK) { status = ch.status; c->err = ch.err; }
  }
  if (status != VICGPU_OK) {
    for (FdChunk& ch : c->chunks) HIPIGN(hipStreamSynchronize(ch.stream));
    return status;
  }
  for (FdChunk& ch : c->chunks) HIPCHK(c, hipStreamWaitEvent(c->stream, ch.done, 0));
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  c->ev_used = 1;
  c->ev_steps = nsteps;
  c->steps_done += nsteps;
  return VICGPU_OK;
}

int vicgpu_synchronize(vicgpu_ctx* c) {
  if (!c) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VICGPU_OK;
}

int vicgpu_last_kernel_ms(vicgpu_ctx* c, double* ms_per_launch, int* nlaunch) {
  if (!c || !ms_per_launch) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double tot = 0;
  for (int i = 0; i < c->ev_used; i++) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
    tot += ms;
  }
  *ms_per_launch = c->ev_steps ? tot / c->ev_steps : 0.0;
  if (nlaunch) *nlaunch = c->ev_steps;
  return VICGPU_OK;
}

static int d2h(vicgpu_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || !dst || !src) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, dst, src, bytes, hipMemcpyDeviceToHost));
  return VICGPU_OK;
}

int vicgpu_get_fluxes(vicgpu_ctx* c, double* flux) { return c ? d2h(c, flux, c->d_flux, sizeof(double) * FX_NROW * c->nhru) : VICGPU_ERR_ARG; }
int vicgpu_get_cell_outputs(vicgpu_ctx* c, double* o) { return c ? d2h(c, o, c->d_cell_out, sizeof(double) * CO_NROW * c->ncell) : VICGPU_ERR_ARG; }
int vicgpu_get_accum(vicgpu_ctx* c, double* a) { return c ? d2h(c, a, c->d_accum, sizeof(double) * CA_NROW * c->ncell) : VICGPU_ERR_ARG; }
int vicgpu_get_cell_errors(vicgpu_ctx* c, int* f) { return c ? d2h(c, f, c->d_cell_err, sizeof(int) * c->ncell) : VICGPU_ERR_ARG; }

int vicgpu_reset_accum(vicgpu_ctx* c) {
  if (!c || !c->d_accum) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemsetAsync(c->d_accum, 0, sizeof(double) * CA_NROW * c->ncell, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_cell_err, 0, sizeof(int) * c->ncell, c->stream));
  return VICGPU_OK;
}

int vicgpu_glacier_mass_balance_fit(vicgpu_ctx* c, double* eq, int reset) {
  if (!c || !c->d_cp || !eq) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  double* d_eq = nullptr;
  HIPCHK(c, hipMalloc(&d_eq, sizeof(double) * GMB_NROW * c->ncell));
  GArgs g;
  g.o = c->o; g.ncell = c->ncell; g.nhru = c->nhru; g.reset = reset ? 1 : 0; g.cell_params = c->d_cp; g.cell_off = c->d_cell_off;
  g.cell_list = c->d_cell_list; g.hpi = c->d_hpi; g.sd = c->d_sd; g.eq = d_eq;
  hipLaunchKernelGGL(vic_glacier_fit, dim3((c->ncell + 63) / 64), dim3(64), 0, c->stream, g);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = copy_on(c->stream, eq, d_eq, sizeof(double) * GMB_NROW * c->ncell, hipMemcpyDeviceToHost);
  HIPIGN(hipFree(d_eq));
  HIPCHK(c, e);
  return VICGPU_OK;
}

int vicgpu_debug_pure(vicgpu_ctx* c, int fn, int n, const double* in, double* out) {
  if (!c || !c->d_cp || fn < 0 || fn >= VICGPU_PURE_NFN_DEVICE || n <= 0 || !in || !out) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  double *d_in = nullptr, *d_out = nullptr;
  HIPCHK(c, hipMalloc(&d_in, sizeof(double) * (size_t)n * VICGPU_PURE_NIN));
  HIPCHK(c, hipMalloc(&d_out, sizeof(double) * (size_t)n));
  HIPCHK(c, copy_on(c->stream, d_in, in, sizeof(double) * (size_t)n * VICGPU_PURE_NIN, hipMemcpyHostToDevice));
  DArgs d;
  d.o = c->o; d.cell_params = c->d_cp; d.ncell = c->ncell; d.fn = fn; d.n = n; d.in = d_in; d.out = d_out;
  hipLaunchKernelGGL(vic_debug_pure, dim3((n + 63) / 64), dim3(64), 0, c->stream, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = copy_on(c->stream, out, d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
  HIPIGN(hipFree(d_in)); HIPIGN(hipFree(d_out));
  HIPCHK(c, e);
  return VICGPU_OK;
}

int vicgpu_set_stream(vicgpu_ctx* c, void* hip_stream) {
  if (!c) return VICGPU_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (hip_stream) {
    if (c->own_stream) HIPIGN(hipStreamDestroy(c->stream));
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
  } else if (!c->own_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return VICGPU_OK;
}

int vicgpu_set_write_fluxes(vicgpu_ctx* c, int on) {
  if (!c) return VICGPU_ERR_ARG;
  c->write_fluxes = on ? 1 : 0;
  return VICGPU_OK;
}

void* vicgpu_device_ptr(vicgpu_ctx* c, int which) {
  if (!c) return nullptr;
  switch (which) {
    case VICGPU_PTR_STATE_D: return c->d_sd;
    case VICGPU_PTR_STATE_I: return c->d_si;
    case VICGPU_PTR_FLUX: return c->d_flux;
    case VICGPU_PTR_FORCING: return c->d_forcing;
    case VICGPU_PTR_ACCUM: return c->d_accum;
    case VICGPU_PTR_CELL_OUT: return c->d_cell_out;
    default: return nullptr;
  }
}

static int state_records(vicgpu_ctx* c, double* host, bool gather) {
  if (!c || !host) return VICGPU_ERR_ARG;
  if (!c->domain_ready) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t bytes = sizeof(double) * (size_t)VICGPU_SR_LEN(c->opt.Nnode) * c->nhru;
  double* d_rec = nullptr;
  int* d_mis = nullptr;
  HIPCHK(c, hipMalloc(&d_rec, bytes));
  hipError_t e = hipMalloc(&d_mis, sizeof(int));
  int mismatch = 0;
  if (e == hipSuccess) e = fill_on(c->stream, d_mis, 0, sizeof(int));
  if (e == hipSuccess && !gather) e = copy_on(c->stream, d_rec, host, bytes, hipMemcpyHostToDevice);
  RArgs a;
  a.nhru = c->nhru; a.Nn = c->opt.Nnode; a.cell_list = c->d_cell_list; a.hpi = c->d_hpi; a.sd = c->d_sd; a.si = c->d_si;
  a.flux = c->d_flux; a.rec = d_rec; a.mismatch = d_mis;
  if (e == hipSuccess && !gather) {
    // read side, pass 1: validate every record before anything is scattered (a reader that throws changes nothing)
    std::vector<int> hpi_band(c->nhru), hpi_veg(c->nhru), list(c->nhru);
    e = copy_on(c->stream, hpi_band.data(), c->d_hpi + (size_t)HPI_BAND * c->nhru, sizeof(int) * c->nhru, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = copy_on(c->stream, hpi_veg.data(), c->d_hpi + (size_t)HPI_VEG_CLASS * c->nhru, sizeof(int) * c->nhru, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = copy_on(c->stream, list.data(), c->d_cell_list, sizeof(int) * c->nhru, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
      const size_t L = VICGPU_SR_LEN(c->opt.Nnode);
      for (int k = 0; k < c->nhru && !mismatch; k++)
        if ((int)host[k * L + SR_BAND_INDEX] != hpi_band[list[k]] || (int)host[k * L + SR_VEG_CLASS] != hpi_veg[list[k]]) mismatch = k + 1;
    }
  }
  if (e == hipSuccess && !mismatch) {
    const unsigned nblk = (unsigned)((c->nhru + 255) / 256);
    if (gather) hipLaunchKernelGGL(vic_state_records<true>, dim3(nblk), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(vic_state_records<false>, dim3(nblk), dim3(256), 0, c->stream, a);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && gather) e = copy_on(c->stream, host, d_rec, bytes, hipMemcpyDeviceToHost);
  }
  HIPIGN(hipFree(d_rec)); HIPIGN(hipFree(d_mis));
  HIPCHK(c, e);
  if (mismatch) {
    c->err = "state record " + std::to_string(mismatch - 1) + ": band / vegetation class do not match the domain (write_model_state.c:179-188)";
    return VICGPU_ERR_ARG;
  }
  return VICGPU_OK;
}
int vicgpu_get_state_records(vicgpu_ctx* c, double* rec) { return state_records(c, rec, true); }
int vicgpu_set_state_records(vicgpu_ctx* c, const double* rec) { return state_records(c, const_cast<double*>(rec), false); }

// ------------------------------------------------------------------------------------------------ put_data (vicgpu_out.h)
int vicgpu_out_nvar(void) { return VOUT_NVAR; }
const char* vicgpu_out_var_name(int id) { return (id >= 0 && id < VOUT_NVAR) ? vout_name_host[id] : nullptr; }
int vicgpu_out_var_id(const char* name) {
  if (!name) return -1;
  for (int v = 0; v < VOUT_NVAR; v++)
    if (strcmp(name, vout_name_host[v]) == 0) return v;
  return -1;
}
int vicgpu_out_var_kind(int id) { return (id >= 0 && id < VOUT_NVAR) ? vout_kind_host[id] : -1; }
int vicgpu_out_var_agg(int id) { return (id >= 0 && id < VOUT_NVAR) ? vout_agg_host[id] : -1; }
int vicgpu_out_var_nelem(const vicgpu_options* opt, int id) {
  if (!opt || id < 0 || id >= VOUT_NVAR) return -1;
  return vout_kind_nelem(vout_kind_host[id], opt->Nnode, opt->Nband, opt->FROZEN_SOIL);
}

int vicgpu_put_data_config(vicgpu_ctx* c, int out_step_ratio) {
  if (!c || out_step_ratio < 1) return VICGPU_ERR_ARG;
  if (!c->domain_ready || !c->d_veglib) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the tree-line adjustment factor of every band (a function of the domain and the vegetation library's overstory flags)
  hipLaunchKernelGGL(vic_derive_tree_adjust, dim3((c->ncell + 63) / 64), dim3(64), 0, c->stream, c->d_cp, c->ncell, c->nhru, c->opt.Nnode,
                     c->opt.Nband, c->d_cell_off, c->d_cell_list, c->d_hpi, c->d_hpd, c->d_veglib);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int r = 0;
  for (int v = 0; v < VOUT_NVAR; v++) {
    c->out_lay.off[v] = r; c->out_lay.agg[v] = vout_agg_host[v];
    r += vout_kind_nelem(vout_kind_host[v], c->opt.Nnode, c->opt.Nband, c->opt.FROZEN_SOIL);
  }
  c->out_lay.off[VOUT_NVAR] = r;
  c->out_nrow = r;
  c->out_step_ratio = out_step_ratio;
  std::vector<unsigned char> rowagg(r);
  for (int v = 0; v < VOUT_NVAR; v++) {
    const bool by_finish = (v == VOUT_AERO_RESIST || v == VOUT_AERO_RESIST1 || v == VOUT_AERO_RESIST2);   // vic_put_finish
    for (int k = c->out_lay.off[v]; k < c->out_lay.off[v + 1]; k++) rowagg[k] = (unsigned char)(by_finish ? VOUT_AGG_SKIP : vout_agg_host[v]);
  }
  if (!c->d_out_data) {
    HIPCHK(c, hipMalloc(&c->d_out_data, sizeof(double) * (size_t)r * c->ncell));
    HIPCHK(c, hipMalloc(&c->d_out_agg, sizeof(double) * (size_t)r * c->ncell));
    HIPCHK(c, hipMalloc(&c->d_pb, sizeof(double) * (size_t)PBX_NROW * c->ncell));
    HIPCHK(c, hipMalloc(&c->d_rowagg, (size_t)r));
  }
  HIPCHK(c, copy_on(c->stream, c->d_rowagg, rowagg.data(), (size_t)r, hipMemcpyHostToDevice));
  HIPCHK(c, fill_on(c->stream, c->d_out_data, 0, sizeof(double) * (size_t)r * c->ncell));
  HIPCHK(c, fill_on(c->stream, c->d_out_agg, 0, sizeof(double) * (size_t)r * c->ncell));
  HIPCHK(c, fill_on(c->stream, c->d_pb, 0, sizeof(double) * (size_t)PBX_NROW * c->ncell));
  c->put_on = true;
  return VICGPU_OK;
}

int vicgpu_put_data_init(vicgpu_ctx* c) {
  if (!c) return VICGPU_ERR_ARG;
  if (!c->domain_ready || !c->put_on || !c->d_veglib) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, launch_put_data(c, c->stream, 0, c->ncell, -1));
  return VICGPU_OK;
}

__global__ __launch_bounds__(256) void vic_out_rows_f32(const double* __restrict__ src, const int* __restrict__ rows, int nrows, int ncell,
                                                         float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nrows * ncell) return;
  const int r = (int)(i / ncell), cc = (int)(i % ncell);
  dst[i] = (float)src[(size_t)rows[r] * ncell + cc];                       // WriteOutputNetCDF.c:387-455 writes floats
}

// the rows of the listed variables, in the order asked for; -1 when an id is out of range
static int out_rows(const vicgpu_ctx* c, int nvar, const int* ids, std::vector<int>& rows) {
  rows.clear();
  for (int k = 0; k < nvar; k++) {
    if (ids[k] < 0 || ids[k] >= VOUT_NVAR) return -1;
    for (int r = c->out_lay.off[ids[k]]; r < c->out_lay.off[ids[k] + 1]; r++) rows.push_back(r);
  }
  return (int)rows.size();
}

int vicgpu_get_outputs(vicgpu_ctx* c, int nvar, const int* var_ids, float* out, int reset) {
  if (!c || nvar < 0 || (nvar > 0 && (!var_ids || !out))) return VICGPU_ERR_ARG;
  if (!c->put_on) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int> rows;
  const int nr = out_rows(c, nvar, var_ids, rows);
  if (nr < 0) return VICGPU_ERR_ARG;
  if (nr > 0) {
    int* d_rows = nullptr;
    float* d_f = nullptr;
    const size_t n = (size_t)nr * c->ncell;
    HIPCHK(c, hipMalloc(&d_rows, sizeof(int) * nr));
    hipError_t e = hipMalloc(&d_f, sizeof(float) * n);
    if (e == hipSuccess) e = copy_on(c->stream, d_rows, rows.data(), sizeof(int) * nr, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(vic_out_rows_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_out_agg, d_rows, nr, c->ncell, d_f);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = copy_on(c->stream, out, d_f, sizeof(float) * n, hipMemcpyDeviceToHost);
    HIPIGN(hipFree(d_rows)); HIPIGN(hipFree(d_f));
    HIPCHK(c, e);
  }
  if (reset) HIPCHK(c, fill_on(c->stream, c->d_out_agg, 0, sizeof(double) * (size_t)c->out_nrow * c->ncell));   // vicNl.c:599-606
  return VICGPU_OK;
}

int vicgpu_get_output_data(vicgpu_ctx* c, int nvar, const int* var_ids, int which, double* out) {
  if (!c || nvar <= 0 || !var_ids || !out || which < 0 || which > 1) return VICGPU_ERR_ARG;
  if (!c->put_on) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  size_t at = 0;
  for (int k = 0; k < nvar; k++) {
    if (var_ids[k] < 0 || var_ids[k] >= VOUT_NVAR) return VICGPU_ERR_ARG;
    const int r0 = c->out_lay.off[var_ids[k]], ne = c->out_lay.off[var_ids[k] + 1] - r0;
    HIPCHK(c, copy_on(c->stream, out + at, (which ? c->d_out_agg : c->d_out_data) + (size_t)r0 * c->ncell, sizeof(double) * (size_t)ne * c->ncell, hipMemcpyDeviceToHost));
    at += (size_t)ne * c->ncell;
  }
  return VICGPU_OK;
}

int vicgpu_get_balance(vicgpu_ctx* c, double* pb) {
  if (!c || !pb) return VICGPU_ERR_ARG;
  if (!c->put_on) return VICGPU_ERR_STATE;
  return d2h(c, pb, c->d_pb, sizeof(double) * (size_t)PB_NROW * c->ncell);
}

int vicgpu_set_fluxes(vicgpu_ctx* c, const double* flux) {
  if (!c || !flux) return VICGPU_ERR_ARG;
  if (!c->domain_ready) return VICGPU_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_on(c->stream, c->d_flux, flux, sizeof(double) * FX_NROW * c->nhru, hipMemcpyHostToDevice));
  return VICGPU_OK;
}

}  // extern "C"
