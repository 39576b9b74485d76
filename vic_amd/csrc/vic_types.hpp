// vic_types.hpp — device-side data types of the MI355X VIC hot path (gfx950 only).
//
// One lane owns one HRU.  Everything a lane needs is either a scalar in registers,
// a small fixed array (soil layers, thermal nodes: unrolled for the instantiated
// node counts) or read on demand through the struct-of-arrays views below, whose
// loads are coalesced because consecutive lanes are consecutive cells of one
// (veg tile, snow band) slot (vic_amd/domain.py numbering).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <limits.h>
#include "vicgpu.h"

#define VIC_DEV __device__ __forceinline__
#define VIC_DEVN __device__ __forceinline__
#ifdef VIC_HOSTEMU   // sanitizer build for the host, tools/hostemu
#define VIC_WAVES_PER_EU(lo, hi)
#else
#define VIC_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

namespace vic {

// ---- constants shared with the reference (vicNl_def.h:138-302, snow.h:34-79) ----
constexpr double ERROR_VAL = -999.0;
constexpr double HUGE_RESIST = 1.e20;
constexpr double SMALL = 1.e-12;
constexpr int INVALID_INT = INT_MIN;
constexpr double ICE_DENSITY = 917.0;
constexpr double VON_K = 0.40;
constexpr double KELVIN = 273.15;
constexpr double STEFAN_B = 5.6696e-8;
constexpr double LF = 3.337e5;
constexpr double RHO_W = 999.842594;
constexpr double CP_AIR = 1013.0;
constexpr double CH_ICE = 2100.0e3;
constexpr double CH_WATER = 4186.8e3;
constexpr double K_SNOW = 2.9302e-6;
constexpr double EPS_MW = 0.62196351;
constexpr double G_GRAV = 9.81;
constexpr double JOULESPCAL = 4.1868;
constexpr double GRAMSPKG = 1000.0;
constexpr double SECPHOUR = 3600;
constexpr double SEC_PER_DAY = 86400.;
constexpr double GLAC_TEMP = 0.0;
constexpr double GLAC_K_ICE = 2.14;
constexpr double SNOW_SURF_DENSITY = 350;
constexpr double CUTOFF_DENSITY = 830;
constexpr double SNOW_DT = 5.0;
constexpr double SURF_DT = 1.0;
constexpr double SOIL_DT = 0.25;
constexpr double COEF_DRAG = 0.2;
constexpr double LIQUID_WATER_CAPACITY = 0.035;
constexpr double LAI_SNOW_MULTIPLIER = 0.0005;
constexpr double MIN_INTERCEPTION_STORAGE = 0.005;
constexpr double MAX_SURFACE_SWE = 0.125;
constexpr double NEW_SNOW_DENSITY = 50.;
constexpr double SNDENS_DMLIMIT = 100.;
constexpr double SNDENS_ETA0 = 3.6e6;
constexpr double SNDENS_C1 = 0.04;
constexpr double SNDENS_C2 = 2.778e-6;
constexpr double SNDENS_C5 = 0.08;
constexpr double SNDENS_C6 = 0.021;
constexpr double SNDENS_F = 0.6;
constexpr double MIN_SWQ_EB_THRES = 0.0010;
constexpr double TRACESNOW = 0.03;
constexpr int NPET = 6;
constexpr int NPET_NON_NAT = 4;
constexpr int PET_VEGNOCR = 5;

// surface cases of VegConditions (VegConditions.h:4-20)
enum { SNOW_FREE = 0, CANOPY = 1, SNOW_COVERED = 2, GLACIER_SURF = 3, NCASE = 4 };

// Section timers / trip counters of the tuning build (-DVIC_PROF, tools/prof_sections.py); compiled out otherwise.
#ifdef VIC_PROF
__device__ unsigned long long vic_prof_cyc[32];
__device__ unsigned long long vic_prof_cnt[32];
VIC_DEV bool prof_leader() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }
#define PROF_T0(name) const long long name = (long long)__builtin_readcyclecounter()
#define PROF_ADD(id, name) do { const long long _d = (long long)__builtin_readcyclecounter() - (name); \
    if (prof_leader()) atomicAdd(&vic_prof_cyc[id], (unsigned long long)_d); } while (0)
#define PROF_LANE(id) atomicAdd(&vic_prof_cnt[id], 1ull)
#define PROF_WAVE(id) do { if (prof_leader()) atomicAdd(&vic_prof_cnt[id], 1ull); } while (0)
#define PROF_VOTE(id, pred) do { const unsigned long long m_ = __ballot(pred); if (prof_leader()) atomicAdd(&vic_prof_cnt[id], (unsigned long long)__popcll(m_)); } while (0)
#else
#define PROF_T0(name) do { } while (0)
#define PROF_ADD(id, name) do { } while (0)
#define PROF_LANE(id) do { } while (0)
#define PROF_WAVE(id) do { } while (0)
#define PROF_VOTE(id, pred) do { } while (0)
#endif

VIC_DEV bool is_error(double x) { return x <= -998.0; }   // RootBrent::resultIsError

// ---- run-time options, passed by value to every kernel (lives in SGPRs / kernarg) ----
struct Opt {
  int Nnode, Nband, dt, snow_step, NF, NR;
  int FULL_ENERGY, FROZEN_SOIL, QUICK_FLUX, NOFLUX, EXP_TRANS, GRND_FLUX_TYPE, TFALLBACK, AERO_RESIST_CANSNOW,
      SNOW_ALBEDO, SNOW_DENSITY, TEMP_TH_TYPE, GLACIER_ID, GLACIER_DYNAMICS, frozen_compat, nveg_types, CORRPREC, IMPLICIT, QUICK_SOLVE, BLOWING;
  double wind_h;
};

// ---- read-only views ----
struct VegLib {
  const double* __restrict__ t;   // [nrow][VL_NFIELD]
  VIC_DEV double f(int idx, int field) const { return t[idx * VL_NFIELD + field]; }
};

// Rows the library appends to its device copy of the cell parameter table (vic_derive_cell_params, once per
// vicgpu_set_domain): the factors of soil_conductivity (soil_conduction.c:7-105) that depend on the layer's soil only.
enum { CPX_KDRY = 0, CPX_KSP, CPX_KWP, CPX_POROSITY, CPX_NFIELD };
#define VIC_CPX_ROW(f, l, Nn, Nb) (VICGPU_CP_NROW(Nn, Nb) + (f) * VIC_NLAYER + (l))
// ... and put_data's tree-line adjustment factor of every band (put_data.c:185-208: a function of the HRU areas, the
// vegetation classes and AboveTreeLine, all fixed with the domain)
#define VIC_CPX_TREE_ROW(b, Nn, Nb) (VICGPU_CP_NROW(Nn, Nb) + CPX_NFIELD * VIC_NLAYER + (b))
#define VIC_CPX_NROW(Nn, Nb) (VICGPU_CP_NROW(Nn, Nb) + CPX_NFIELD * VIC_NLAYER + (Nb))

struct CellView {
  const double* __restrict__ cp;   // [CP_NROW + derived rows][ncell]
  int ncell, c, Nn, Nb;
  VIC_DEV double s(int row) const { return cp[(size_t)row * ncell + c]; }
  VIC_DEV double x(int f, int l) const { return s(VIC_CPX_ROW(f, l, Nn, Nb)); }
  VIC_DEV double lay(int f, int l) const { return s(VICGPU_CP_LAYER(f, l)); }
  VIC_DEV double node(int f, int n) const { return s(VICGPU_CP_NODE(f, n, Nn)); }
  VIC_DEV double band(int f, int b) const { return s(VICGPU_CP_BAND(f, b, Nn, Nb)); }
  VIC_DEV double zwt_zwt(int l, int i) const { return s(VICGPU_CP_ZWT_ZWT(l, i, Nn, Nb)); }
  VIC_DEV double zwt_moist(int l, int i) const { return s(VICGPU_CP_ZWT_MOIST(l, i, Nn, Nb)); }
};

// forcing of one record for this lane's cell: [VIC_NFORCE][NF+1][ncell]
struct Forcing {
  const double* __restrict__ f;
  const unsigned char* __restrict__ snowflag;   // [NF+1][ncell]
  int ncell, c, nsub;
  VIC_DEV double v(int var, int sub) const { return f[((size_t)var * nsub + sub) * ncell + c]; }
  VIC_DEV int flag(int sub) const { return snowflag[(size_t)sub * ncell + c]; }
};

struct Dmy { int month, day_in_year, hour, day, year; };

// the soil-layer parameters every part of the step touches, held in registers
struct Soil3 {
  double depth[3], max_moist[3], Wcr[3], Wpwp[3], resid_moist[3];
};

struct Vc { double v[NCASE]; };
// v[k] for a run-time k without indexing memory: a dynamically indexed member would pin the whole enclosing struct
// (HruWork, StepConst, ...) to scratch memory instead of registers
VIC_DEV double sel_by_value(double a, double b, double c, double d, int k) { return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d)); }
VIC_DEV double sel3(const double* a, int k) { return sel_by_value(a[0], a[1], a[2], a[2], k); }
VIC_DEV double vsel(const Vc& x, int k) { return sel_by_value(x.v[0], x.v[1], x.v[2], x.v[3], k); }

// ---- per-HRU working state ----
struct Snow {
  double albedo, coldcontent, coverage, density, depth, max_swq, pack_temp, pack_water, snow_canopy, store_coverage,
         store_swq, surf_temp, surf_water, swq, swq_slope, tmp_int_storage;
  double blowing_flux, canopy_vapor_flux, mass_error, melt, Qnet, surface_flux, vapor_flux;
  int last_snow, MELTING, snow, store_snow, surf_temp_fbcount, surf_temp_fbflag;
};

struct VegVar { double canopyevap, throughfall, Wdew; };

// energy_bal_struct fields the snow side writes (solve_snow / snow_intercept / snow_melt); the reference keeps a
// whole struct copy "snow_energy" for these (surface_fluxes.c:301)
struct SnowEnergy {
  double AlbedoOver, LongOverIn, NetLongOver, NetShortOver, ShortOverIn, Tfoliage;
  double canopy_advection, canopy_latent, canopy_latent_sub, canopy_refreeze, canopy_sensible;
  double advected_sensible, advection, deltaCC, latent, latent_sub, refreeze_energy, sensible, snow_flux, error;
  int Tfoliage_fbflag, Tfoliage_fbcount;
};

// fields the soil side writes (calc_surf_energy_bal); the reference's "soil_energy" copy (surface_fluxes.c:302)
struct SoilEnergy {
  double advection, deltaCC, refreeze_energy, deltaH, fusion, grnd_flux, latent, latent_sub, sensible, snow_flux, error,
         advected_sensible;
  double NetShortGrnd, NetLongUnder, NetShortUnder, LongUnderOut, AlbedoUnder, melt_energy, Tsurf;
  double kappa[2], Cs[2];
  double fdepth[3], tdepth[3];
  int Tsurf_fbflag, Tsurf_fbcount, frozen, Nfrost, Nthaw;
};

template <int NN>
struct Nodes {
  double T[NN], moist[NN], ice[NN], kappa[NN], Cs[NN];
  int fbflag[NN], fbcount[NN];
};

struct Glac {
  double cold_content, surf_temp, Qnet, mass_balance, ice_mass_balance, cum_mass_balance, accumulation, melt, vapor_flux,
         water_storage, outflow, outflow_coef, inflow;
  int surf_temp_fbcount, surf_temp_fbflag;
};

}  // namespace vic
