// vic_putdata.hpp — put_data on the device (device only, gfx950): the per-cell output aggregation of put_data.c:7-760
// with collect_wb_terms (:762-948) and collect_eb_terms (:950-1232), the balance checks of
// calc_water_energy_balance_errors.c:7-94 and the temporal aggregation of put_data.c:663-685.
//
// Three kernels per step, all with consecutive lanes = consecutive cells, so every access to the [row][cell] tables
// (state, fluxes, forcing, outputs) is a coalesced 512-byte row segment:
//   vic_put_sum<PART>   one lane = one cell x one third of the variables (water-balance terms / energy-balance terms / band
//                       variables).  The cell's HRUs are visited in hruList order (the order put_data.c:260 iterates in), so
//                       every area-weighted sum is formed in the reference's order of additions.  (vic_put_zero clears the rows first.)  Scalar sums live in
//                       registers and are written once; the band variables, whose row depends on the HRU's band, are
//                       read-modify-written in one batch per HRU (all loads, then all stores: the rows' offsets are run-time
//                       values, so the compiler may not reorder a load above an earlier store itself).
//   vic_put_finish      one lane = one cell: forcing echoes, the divisions by partial areas, totals, storage changes, the
//                       water and energy balance errors and their bookkeeping (put_data.c:549-633).
//   vic_put_aggregate   one lane = one cell x 32 rows: temporal aggregation (put_data.c:663-685).
// Option subset as everywhere in this library: no lakes, Ndist = 1, SPATIAL_FROST / EXCESS_ICE off, MOISTFRACT and
// ALMA_OUTPUT off.  The tree-line adjustment factors (put_data.c:185-208) are derived once per domain (VIC_CPX_TREE_ROW).
#pragma once
#include "vic_types.hpp"
#include "vicgpu_out.h"

namespace vic {

#define VIC_OUT_KIND_(name, kind, agg) kind,
#define VIC_OUT_AGG_(name, kind, agg) agg,
#define VIC_OUT_NAME_(name, kind, agg) "OUT_" #name,
static const int vout_kind_host[VOUT_NVAR] = {VICGPU_OUT_VARS(VIC_OUT_KIND_)};
static const int vout_agg_host[VOUT_NVAR] = {VICGPU_OUT_VARS(VIC_OUT_AGG_)};
static const char* const vout_name_host[VOUT_NVAR] = {VICGPU_OUT_VARS(VIC_OUT_NAME_)};

inline int vout_kind_nelem(int kind, int Nnode, int Nband, int FROZEN_SOIL) {
  switch (kind) {
    case VOUT_KLAYER: return VIC_NLAYER;
    case VOUT_KNODE: return Nnode;
    case VOUT_KBAND: return Nband;
    case VOUT_KFRONT: return FROZEN_SOIL ? VIC_MAX_FRONTS : 1;      // output_list_utils.c:298-301
    default: return 1;
  }
}

// first row of every variable in the [row][cell] output tables, and the aggregation type of every row's variable
struct OutLayout {
  int off[VOUT_NVAR + 1];
  int agg[VOUT_NVAR];
};
constexpr int VOUT_MAX_ROWS = 1280;   // > 110 + 7*3 + 2*MAX_NODES + 34*MAX_BANDS + 2*MAX_FRONTS

struct OArgs {
  Opt o;
  OutLayout lay;
  int ncell, nhru, c0, ccount, rec, out_step_ratio;
  const int* cell_off;
  const int* cell_list;
  const double* cell_params;
  const double* veglib;
  const int* hpi;
  const double* hpd;
  const double* sd;
  const int* si;
  const double* flux;
  const double* forcing;        // this step: [VIC_NFORCE][NF+1][ncell] (unused when rec < 0)
  const double* cell_out;       // [CO_NROW][ncell]
  double* out_data;             // [nrow][ncell]
  double* out_agg;              // [nrow][ncell]
  double* pb;                   // [PBX_NROW][ncell]: the public rows + this file's own
};

// rows this file appends to the bookkeeping table: the partial areas of put_data.c:262-300
enum { PBX_CV_BARESOIL = PB_NROW, PBX_CV_VEG, PBX_CV_OVERSTORY, PBX_CV_SNOW, PBX_CV_GLACIER, PBX_NROW };
enum { VOUT_AGG_SKIP = 3 };     // rows the aggregation kernel leaves alone (vic_put_finish writes their aggregates)
constexpr int PUT_NPART = 3, PUT_AGG_ROWS = 32;

// what the three kernels share per HRU: whether put_data visits it and its factors
struct PutHru { bool run, HasVeg, HasGlac, overstory; int band; double Cv, ThisAreaFract, TreeAdjust; };
VIC_DEV PutHru put_hru(const OArgs& a, const CellView& cv, int g) {
  const size_t nh = a.nhru;
  PutHru h;
  h.Cv = a.hpd[(size_t)HPD_CV * nh + g];
  h.band = a.hpi[(size_t)HPI_BAND * nh + g];
  h.ThisAreaFract = cv.band(CPB_AREAFRACT, h.band);
  const bool is_glac = a.hpi[(size_t)HPI_IS_GLACIER * nh + g] != 0, art_bare = a.hpi[(size_t)HPI_IS_ARTIFICIAL_BARE * nh + g] != 0;
  h.HasVeg = !(art_bare || is_glac);
  h.HasGlac = is_glac;
  h.overstory = a.veglib[(size_t)a.hpi[(size_t)HPI_VEG_INDEX * nh + g] * VL_NFIELD + VL_OVERSTORY] != 0.0;
  const bool atl = cv.band(CPB_ABOVETREELINE, h.band) != 0.0;
  h.TreeAdjust = cv.s(VIC_CPX_TREE_ROW(h.band, cv.Nn, cv.Nb));                       // put_data.c:199-208, derived with the domain
  h.run = (h.Cv > 0) && (h.ThisAreaFract > 0.) && (art_bare || (!atl || (atl && !h.overstory)));   // :289-290
  return h;
}

#define SD(row) a.sd[(size_t)(row) * nh + g]
#define SI(row) a.si[(size_t)(row) * nh + g]
#define FX(row) a.flux[(size_t)(row) * nh + g]
#define ROW(var, i) od[(size_t)(a.lay.off[VOUT_##var] + (i)) * nc]
#define PB(r) a.pb[(size_t)(r) * nc + c]

// PART 0: collect_wb_terms (put_data.c:762-948, mu = 1, lakefactor = 1, TreeAdjustFactor = 1) and the partial areas
// PART 1: collect_eb_terms, the cell-wide part (put_data.c:950-1135)
// PART 2: collect_eb_terms, the band variables (put_data.c:1137-1232) and OUT_ELEV_BAND
template <int PART>
VIC_DEV void put_sum_part(const OArgs& a, int c) {
  const size_t nh = a.nhru, nc = a.ncell;
  const Opt& o = a.o;
  const int Nn = o.Nnode;
  double* od = a.out_data + c;
  CellView cv{a.cell_params, a.ncell, c, Nn, o.Nband};
  const int k0 = a.cell_off[c], k1 = a.cell_off[c + 1];
  if (PART == 0) {
    double cv_baresoil = 0, cv_veg = 0, cv_overstory = 0, cv_snow = 0, cv_glacier = 0;
    double TRANSP_VEG = 0, EVAP_BARE = 0, SUB_SNOW = 0, SUB_SURFACE = 0, SUB_BLOWING = 0, SUB_CANOP = 0, EVAP_CANOP = 0, EVAP = 0;
    double PET[6] = {0, 0, 0, 0, 0, 0};
    double ASAT = 0, RUNOFF = 0, BASEFLOW = 0, INFLOW = 0, WDEW = 0, AERO_COND1 = 0, AERO_COND2 = 0, AERO_COND = 0;
    double SOIL_LIQ[3] = {0, 0, 0}, SOIL_ICE[3] = {0, 0, 0}, ZWTL[3] = {0, 0, 0}, SOIL_TEMP[3] = {0, 0, 0};
    double SOIL_WET = 0, ROOTMOIST = 0, ZWT = 0, ZWT2 = 0, ZWT3 = 0, SWE = 0, SNOW_DEPTH = 0, SALBEDO = 0, SNOW_SURF_TEMP = 0,
           SNOW_PACK_TEMP = 0, SNOW_CANOPY = 0, SNOW_MELT = 0, SNOW_COVER = 0;
    double GLAC_WAT_STOR = 0, GLAC_AREA = 0, GLAC_MBAL = 0, GLAC_IMBAL = 0, GLAC_ACCUM = 0, GLAC_MELT = 0, GLAC_SUB = 0, GLAC_INFLOW = 0,
           GLAC_OUTFLOW = 0, GLAC_OUTFLOW_COEF = 0;
    for (int k = k0; k < k1; k++) {
      const int g = a.cell_list[k];
      const PutHru h = put_hru(a, cv, g);
      if (!h.run) continue;
      const double Cv = h.Cv, ThisTreeAdjust = h.TreeAdjust;
      const double swq = SD(SD_SNOW_SWQ);
      if (h.HasVeg) cv_veg += Cv * 1. * ThisTreeAdjust; else cv_baresoil += Cv * 1. * ThisTreeAdjust;
      if (h.overstory) cv_overstory += Cv * 1. * ThisTreeAdjust;
      if (swq > 0.0) cv_snow += Cv * 1. * ThisTreeAdjust;
      if (h.HasGlac) cv_glacier += Cv * 1. * ThisTreeAdjust;
      const double AreaFactor = Cv * 1. * ThisTreeAdjust * 1.;
      double tmp_evap = 0.0;
#pragma unroll
      for (int l = 0; l < 3; l++) tmp_evap += FX(FX_EVAP0 + l);
      if (h.HasVeg) TRANSP_VEG += tmp_evap * AreaFactor; else EVAP_BARE += tmp_evap * AreaFactor;
      const double vapor_flux = FX(FX_SNOW_VAPOR_FLUX), canopy_vapor_flux = FX(FX_SNOW_CANOPY_VAPOR_FLUX);
      tmp_evap += vapor_flux * 1000.;
      SUB_SNOW += vapor_flux * 1000. * AreaFactor;
      SUB_SURFACE += FX(FX_SNOW_SURFACE_FLUX) * 1000. * AreaFactor;
      SUB_BLOWING += FX(FX_SNOW_BLOWING_FLUX) * 1000. * AreaFactor;
      if (h.HasVeg) { tmp_evap += canopy_vapor_flux * 1000.; SUB_CANOP += canopy_vapor_flux * 1000. * AreaFactor; }
      if (h.HasVeg) { const double ce = FX(FX_CANOPYEVAP); tmp_evap += ce; EVAP_CANOP += ce * AreaFactor; }
      const double gl_vapor = FX(FX_GLAC_VAPOR_FLUX);
      if (h.HasGlac) tmp_evap += gl_vapor * 1000.;
      EVAP += tmp_evap * AreaFactor;
#pragma unroll
      for (int p = 0; p < 6; p++) PET[p] += FX(FX_POT_EVAP0 + p) * AreaFactor;
      ASAT += FX(FX_ASAT) * AreaFactor;
      RUNOFF += FX(FX_RUNOFF) * AreaFactor;
      BASEFLOW += FX(FX_BASEFLOW) * AreaFactor;
      INFLOW += (FX(FX_INFLOW)) * AreaFactor;
      if (h.HasVeg) WDEW += SD(SD_WDEW) * AreaFactor;
      double tmp_cond1, tmp_cond2;
      const double ars = FX(FX_AERO_RESIST_SURFACE), aro = FX(FX_AERO_RESIST_OVERSTORY);
      if (ars > SMALL) tmp_cond1 = (1 / ars) * AreaFactor; else tmp_cond1 = HUGE_RESIST;
      AERO_COND1 += tmp_cond1;
      if (h.overstory) {
        if (aro > SMALL) tmp_cond2 = (1 / aro) * AreaFactor; else tmp_cond2 = HUGE_RESIST;
      } else tmp_cond2 = HUGE_RESIST;
      AERO_COND2 += tmp_cond2;
      if (h.overstory) AERO_COND += tmp_cond2; else AERO_COND += tmp_cond1;
#pragma unroll
      for (int l = 0; l < 3; l++) {
        double tmp_moist = SD(SD_MOIST0 + l);
        const double tmp_ice = SD(SD_ICE0 + l);
        tmp_moist -= tmp_ice;
        SOIL_LIQ[l] += tmp_moist * AreaFactor;
        SOIL_ICE[l] += tmp_ice * AreaFactor;
        ZWTL[l] += FX(FX_ZWTL0 + l) * AreaFactor;
        SOIL_TEMP[l] += SD(SD_LAYER_T0 + l) * AreaFactor;
      }
      SOIL_WET += FX(FX_WETNESS) * AreaFactor;
      ROOTMOIST += FX(FX_ROOTMOIST) * AreaFactor;
      ZWT += FX(FX_ZWT) * AreaFactor; ZWT2 += FX(FX_ZWT2) * AreaFactor; ZWT3 += FX(FX_ZWT3) * AreaFactor;
      SWE += swq * AreaFactor * 1000.;
      SNOW_DEPTH += SD(SD_SNOW_DEPTH) * AreaFactor * 100.;
      if (swq > 0.0) {
        SALBEDO += SD(SD_SNOW_ALBEDO) * AreaFactor;
        SNOW_SURF_TEMP += SD(SD_SNOW_SURF_TEMP) * AreaFactor;
        SNOW_PACK_TEMP += SD(SD_SNOW_PACK_TEMP) * AreaFactor;
      }
      if (h.HasVeg) SNOW_CANOPY += (SD(SD_SNOW_CANOPY)) * AreaFactor * 1000.;
      SNOW_MELT += FX(FX_SNOW_MELT) * AreaFactor * 1000.;                             // sic
      SNOW_COVER += SD(SD_SNOW_COVERAGE) * AreaFactor;
      if (h.HasGlac) {
        GLAC_WAT_STOR += SD(SD_GLAC_WATER_STORAGE) * AreaFactor * 1000.;
        GLAC_AREA += AreaFactor;
        GLAC_MBAL += FX(FX_GLAC_MASS_BALANCE) * AreaFactor * 1000.;
        GLAC_IMBAL += FX(FX_GLAC_ICE_MASS_BALANCE) * AreaFactor * 1000.;
        GLAC_ACCUM += FX(FX_GLAC_ACCUMULATION) * AreaFactor * 1000.;
        GLAC_MELT += FX(FX_GLAC_MELT) * AreaFactor * 1000.;
        GLAC_SUB += gl_vapor * AreaFactor * 1000.;
        GLAC_INFLOW += FX(FX_GLAC_INFLOW) * AreaFactor * 1000.;
        GLAC_OUTFLOW += FX(FX_GLAC_OUTFLOW) * AreaFactor * 1000.;
        GLAC_OUTFLOW_COEF += FX(FX_GLAC_OUTFLOW_COEF) * AreaFactor;
      }
    }
    PB(PBX_CV_BARESOIL) = cv_baresoil; PB(PBX_CV_VEG) = cv_veg; PB(PBX_CV_OVERSTORY) = cv_overstory; PB(PBX_CV_SNOW) = cv_snow;
    PB(PBX_CV_GLACIER) = cv_glacier;
    ROW(TRANSP_VEG, 0) = TRANSP_VEG; ROW(EVAP_BARE, 0) = EVAP_BARE; ROW(SUB_SNOW, 0) = SUB_SNOW; ROW(SUB_SURFACE, 0) = SUB_SURFACE;
    ROW(SUB_BLOWING, 0) = SUB_BLOWING; ROW(SUB_CANOP, 0) = SUB_CANOP; ROW(EVAP_CANOP, 0) = EVAP_CANOP; ROW(EVAP, 0) = EVAP;
    ROW(PET_SATSOIL, 0) = PET[0]; ROW(PET_H2OSURF, 0) = PET[1]; ROW(PET_SHORT, 0) = PET[2]; ROW(PET_TALL, 0) = PET[3];
    ROW(PET_NATVEG, 0) = PET[4]; ROW(PET_VEGNOCR, 0) = PET[5];
    ROW(ASAT, 0) = ASAT; ROW(RUNOFF, 0) = RUNOFF; ROW(BASEFLOW, 0) = BASEFLOW; ROW(INFLOW, 0) = INFLOW; ROW(WDEW, 0) = WDEW;
    ROW(AERO_COND1, 0) = AERO_COND1; ROW(AERO_COND2, 0) = AERO_COND2; ROW(AERO_COND, 0) = AERO_COND;
#pragma unroll
    for (int l = 0; l < 3; l++) { ROW(SOIL_LIQ, l) = SOIL_LIQ[l]; ROW(SOIL_ICE, l) = SOIL_ICE[l]; ROW(ZWTL, l) = ZWTL[l]; ROW(SOIL_TEMP, l) = SOIL_TEMP[l]; }
    ROW(SOIL_WET, 0) = SOIL_WET; ROW(ROOTMOIST, 0) = ROOTMOIST; ROW(ZWT, 0) = ZWT; ROW(ZWT2, 0) = ZWT2; ROW(ZWT3, 0) = ZWT3;
    ROW(SWE, 0) = SWE; ROW(SNOW_DEPTH, 0) = SNOW_DEPTH; ROW(SALBEDO, 0) = SALBEDO; ROW(SNOW_SURF_TEMP, 0) = SNOW_SURF_TEMP;
    ROW(SNOW_PACK_TEMP, 0) = SNOW_PACK_TEMP; ROW(SNOW_CANOPY, 0) = SNOW_CANOPY; ROW(SNOW_MELT, 0) = SNOW_MELT; ROW(SNOW_COVER, 0) = SNOW_COVER;
    ROW(GLAC_WAT_STOR, 0) = GLAC_WAT_STOR; ROW(GLAC_AREA, 0) = GLAC_AREA; ROW(GLAC_MBAL, 0) = GLAC_MBAL; ROW(GLAC_IMBAL, 0) = GLAC_IMBAL;
    ROW(GLAC_ACCUM, 0) = GLAC_ACCUM; ROW(GLAC_MELT, 0) = GLAC_MELT; ROW(GLAC_SUB, 0) = GLAC_SUB; ROW(GLAC_INFLOW, 0) = GLAC_INFLOW;
    ROW(GLAC_OUTFLOW, 0) = GLAC_OUTFLOW; ROW(GLAC_OUTFLOW_COEF, 0) = GLAC_OUTFLOW_COEF;
  }
  if (PART == 1) {
    double FDEPTH[3] = {0, 0, 0}, TDEPTH[3] = {0, 0, 0}, TNODE[VIC_MAX_NODES], FBNODE[VIC_MAX_NODES];
#pragma unroll
    for (int n = 0; n < VIC_MAX_NODES; n++) { TNODE[n] = 0; FBNODE[n] = 0; }
    double SURF_FROST_FRAC = 0, BARESOILT = 0, VEGT = 0, SURF_TEMP = 0, SURFT_FBFLAG = 0, SNOWT_FBFLAG = 0, TFOL_FBFLAG = 0, TCAN_FBFLAG = 0,
           GLAC_TSURF_FBFLAG = 0, NET_SHORT = 0, NET_LONG = 0, IN_LONG = 0, ALBEDO = 0, LATENT = 0, LATENT_SUB = 0, SENSIBLE = 0, GRND_FLUX = 0,
           DELTAH = 0, FUSION = 0, ENERGY_ERROR = 0, RAD_TEMP = 0, DELTACC = 0, ADVECTION = 0, SNOW_FLUX = 0, RFRZ_ENERGY = 0, MELT_ENERGY = 0,
           ADV_SENS = 0, GLAC_SURF_TEMP = 0, GLAC_DELTACC = 0, GLAC_FLUX = 0, GLAC_MELT_ENERGY = 0;
    double fb_tsurf = 0, fb_tsoil = 0, fb_tsnow = 0, fb_tfol = 0, fb_tcan = 0, fb_tglac = 0;
    for (int k = k0; k < k1; k++) {
      const int g = a.cell_list[k];
      const PutHru h = put_hru(a, cv, g);
      if (!h.run) continue;
      const double AreaFactor = h.Cv * h.TreeAdjust * 1.;
      if (o.FROZEN_SOIL) {
#pragma unroll
        for (int l = 0; l < VIC_MAX_FRONTS; l++) {
          const double fd = FX(FX_FDEPTH0 + l), td = FX(FX_TDEPTH0 + l);
          if (!isnan(fd)) FDEPTH[l] += fd * AreaFactor * 100.;
          if (!isnan(td)) TDEPTH[l] += td * AreaFactor * 100.;
        }
      }
      double tmp_fract = 0;
      if (SD(SD_ICE0) > 0) tmp_fract = 1.;
      SURF_FROST_FRAC += tmp_fract * AreaFactor;
      const bool snowing = SI(SI_SNOW_SNOW) != 0;
      const double Tsurf = SD(SD_TSURF);
      double rad_temp;
      if (h.overstory && snowing) rad_temp = SD(SD_TCANOPY) + KELVIN; else rad_temp = Tsurf + KELVIN;
      if (h.HasVeg) BARESOILT += (rad_temp - KELVIN) * AreaFactor;                    // sic: inverted (Appendix C #10)
      else {
        if (h.overstory && !snowing) VEGT += SD(SD_TFOLIAGE) * AreaFactor;
        else VEGT += (rad_temp - KELVIN) * AreaFactor;
      }
      SURF_TEMP += Tsurf * AreaFactor;
#pragma unroll
      for (int n = 0; n < VIC_MAX_NODES; n++) {
        if (n < Nn) {
          TNODE[n] += SD(VICGPU_SD_NODE(SDN_T, n, Nn)) * AreaFactor;
          FBNODE[n] += SI(VICGPU_SI_NODE(SIN_T_FBFLAG, n, Nn)) * AreaFactor;
          fb_tsoil += SI(VICGPU_SI_NODE(SIN_T_FBCOUNT, n, Nn));
        }
      }
      SURFT_FBFLAG += SI(SI_TSURF_FBFLAG) * AreaFactor; fb_tsurf += SI(SI_TSURF_FBCOUNT);
      SNOWT_FBFLAG += SI(SI_SNOW_SURF_TEMP_FBFLAG) * AreaFactor; fb_tsnow += SI(SI_SNOW_SURF_TEMP_FBCOUNT);
      TFOL_FBFLAG += SI(SI_TFOLIAGE_FBFLAG) * AreaFactor; fb_tfol += SI(SI_TFOLIAGE_FBCOUNT);
      TCAN_FBFLAG += SI(SI_TCANOPY_FBFLAG) * AreaFactor; fb_tcan += SI(SI_TCANOPY_FBCOUNT);
      GLAC_TSURF_FBFLAG += SI(SI_GLAC_SURF_TEMP_FBFLAG) * AreaFactor; fb_tglac += SI(SI_GLAC_SURF_TEMP_FBCOUNT);
      NET_SHORT += FX(FX_NET_SHORT_ATMOS) * AreaFactor;
      NET_LONG += FX(FX_NET_LONG_ATMOS) * AreaFactor;
      if (snowing && h.overstory) IN_LONG += SD(SD_LONGOVERIN) * AreaFactor; else IN_LONG += FX(FX_LONG_UNDER_IN) * AreaFactor;
      if (snowing && h.overstory) ALBEDO += SD(SD_ALBEDO_OVER) * AreaFactor; else ALBEDO += SD(SD_ALBEDO_UNDER) * AreaFactor;
      LATENT -= FX(FX_ATMOS_LATENT) * AreaFactor;
      LATENT_SUB -= FX(FX_ATMOS_LATENT_SUB) * AreaFactor;
      SENSIBLE -= FX(FX_ATMOS_SENSIBLE) * AreaFactor;
      GRND_FLUX -= SD(SD_GRND_FLUX) * AreaFactor;
      DELTAH -= SD(SD_DELTAH) * AreaFactor;
      FUSION -= SD(SD_FUSION) * AreaFactor;
      ENERGY_ERROR += SD(SD_ERROR) * AreaFactor;
      RAD_TEMP += ((rad_temp) * (rad_temp) * (rad_temp) * (rad_temp)) * AreaFactor;
      DELTACC += SD(SD_DELTACC) * AreaFactor;
      if (snowing && h.overstory) ADVECTION += SD(SD_CANOPY_ADVECTION) * AreaFactor;
      ADVECTION += SD(SD_ADVECTION) * AreaFactor;
      SNOW_FLUX += SD(SD_SNOW_FLUX) * AreaFactor;
      if (snowing && h.overstory) RFRZ_ENERGY += SD(SD_CANOPY_REFREEZE) * AreaFactor;
      RFRZ_ENERGY += SD(SD_REFREEZE_ENERGY) * AreaFactor;
      MELT_ENERGY += SD(SD_MELT_ENERGY) * AreaFactor;
      if (!h.overstory) ADV_SENS -= SD(SD_ADVECTED_SENSIBLE) * AreaFactor;
      if (h.HasGlac) {
        GLAC_SURF_TEMP += SD(SD_GLAC_SURF_TEMP) * AreaFactor;
        GLAC_DELTACC += FX(FX_DELTACC_GLAC) * AreaFactor;
        GLAC_FLUX += FX(FX_GLACIER_FLUX) * AreaFactor;
        GLAC_MELT_ENERGY += FX(FX_GLACIER_MELT_ENERGY) * AreaFactor;
      }
    }
    if (o.FROZEN_SOIL) {
#pragma unroll
      for (int l = 0; l < VIC_MAX_FRONTS; l++) { ROW(FDEPTH, l) = FDEPTH[l]; ROW(TDEPTH, l) = TDEPTH[l]; }
    }
#pragma unroll
    for (int n = 0; n < VIC_MAX_NODES; n++)
      if (n < Nn) { ROW(SOIL_TNODE, n) = TNODE[n]; ROW(SOILT_FBFLAG, n) = FBNODE[n]; }
    ROW(SURF_FROST_FRAC, 0) = SURF_FROST_FRAC; ROW(BARESOILT, 0) = BARESOILT; ROW(VEGT, 0) = VEGT; ROW(SURF_TEMP, 0) = SURF_TEMP;
    ROW(SURFT_FBFLAG, 0) = SURFT_FBFLAG; ROW(SNOWT_FBFLAG, 0) = SNOWT_FBFLAG; ROW(TFOL_FBFLAG, 0) = TFOL_FBFLAG; ROW(TCAN_FBFLAG, 0) = TCAN_FBFLAG;
    ROW(GLAC_TSURF_FBFLAG, 0) = GLAC_TSURF_FBFLAG; ROW(NET_SHORT, 0) = NET_SHORT; ROW(NET_LONG, 0) = NET_LONG; ROW(IN_LONG, 0) = IN_LONG;
    ROW(ALBEDO, 0) = ALBEDO; ROW(LATENT, 0) = LATENT; ROW(LATENT_SUB, 0) = LATENT_SUB; ROW(SENSIBLE, 0) = SENSIBLE; ROW(GRND_FLUX, 0) = GRND_FLUX;
    ROW(DELTAH, 0) = DELTAH; ROW(FUSION, 0) = FUSION; ROW(ENERGY_ERROR, 0) = ENERGY_ERROR; ROW(RAD_TEMP, 0) = RAD_TEMP; ROW(DELTACC, 0) = DELTACC;
    ROW(ADVECTION, 0) = ADVECTION; ROW(SNOW_FLUX, 0) = SNOW_FLUX; ROW(RFRZ_ENERGY, 0) = RFRZ_ENERGY; ROW(MELT_ENERGY, 0) = MELT_ENERGY;
    ROW(ADV_SENS, 0) = ADV_SENS; ROW(GLAC_SURF_TEMP, 0) = GLAC_SURF_TEMP; ROW(GLAC_DELTACC, 0) = GLAC_DELTACC; ROW(GLAC_FLUX, 0) = GLAC_FLUX;
    ROW(GLAC_MELT_ENERGY, 0) = GLAC_MELT_ENERGY;
    PB(PB_FB_TSURF) += fb_tsurf; PB(PB_FB_TSOIL) += fb_tsoil; PB(PB_FB_TSNOWSURF) += fb_tsnow; PB(PB_FB_TFOLIAGE) += fb_tfol;
    PB(PB_FB_TCANOPY) += fb_tcan; PB(PB_FB_TGLACSURF) += fb_tglac;
  }
  if (PART == 2) {
    // the HRU's band picks the row: one batched read-modify-write per HRU.  The rows start from the zeros of the memset.
    for (int k = k0; k < k1; k++) {
      const int g = a.cell_list[k];
      const PutHru h = put_hru(a, cv, g);
      if (!h.run) continue;
      const int band = h.band;
      const double Cv = h.Cv, bandFactor = Cv * 1. / h.ThisAreaFract;
      const bool snowing = SI(SI_SNOW_SNOW) != 0;
      const double nsa = FX(FX_NET_SHORT_ATMOS), nla = FX(FX_NET_LONG_ATMOS);
      const double alb = (snowing && h.overstory) ? SD(SD_ALBEDO_OVER) : SD(SD_ALBEDO_UNDER);
      enum { B_AREA, B_SWE, B_DEPTH, B_CANOPY, B_MELT, B_COVER, B_DELTACC, B_ADVECTION, B_SNOW_FLUX, B_RFRZ, B_MELT_ENERGY, B_ADV_SENS,
             B_SURFT, B_PACKT, B_LATENT_SUB, B_NET_SHORT, B_NET_LONG, B_ALBEDO, B_LATENT, B_SENSIBLE, B_GRND, BN };
      double inc[BN], cur[BN];
      inc[B_AREA] = (Cv * 1.);
      inc[B_SWE] = SD(SD_SNOW_SWQ) * bandFactor * 1000.;
      inc[B_DEPTH] = SD(SD_SNOW_DEPTH) * bandFactor * 100.;
      inc[B_CANOPY] = h.HasVeg ? (SD(SD_SNOW_CANOPY)) * bandFactor * 1000. : 0.;
      inc[B_MELT] = FX(FX_SNOW_MELT) * bandFactor;
      inc[B_COVER] = SD(SD_SNOW_COVERAGE) * bandFactor;
      inc[B_DELTACC] = SD(SD_DELTACC) * bandFactor;
      inc[B_ADVECTION] = SD(SD_ADVECTION) * bandFactor;
      inc[B_SNOW_FLUX] = SD(SD_SNOW_FLUX) * bandFactor;
      inc[B_RFRZ] = SD(SD_REFREEZE_ENERGY) * bandFactor;
      inc[B_MELT_ENERGY] = SD(SD_MELT_ENERGY) * bandFactor;
      inc[B_ADV_SENS] = SD(SD_ADVECTED_SENSIBLE) * bandFactor;
      inc[B_SURFT] = SD(SD_SNOW_SURF_TEMP) * bandFactor;
      inc[B_PACKT] = SD(SD_SNOW_PACK_TEMP) * bandFactor;
      inc[B_LATENT_SUB] = SD(SD_LATENT_SUB) * bandFactor;
      inc[B_NET_SHORT] = nsa * bandFactor;
      inc[B_NET_LONG] = nla * bandFactor;
      inc[B_ALBEDO] = alb * bandFactor;
      inc[B_LATENT] = SD(SD_LATENT) * bandFactor;
      inc[B_SENSIBLE] = SD(SD_SENSIBLE) * bandFactor;
      inc[B_GRND] = SD(SD_GRND_FLUX) * bandFactor;
      double* rows[BN];
      rows[B_AREA] = &ROW(AREA_BAND, band); rows[B_SWE] = &ROW(SWE_BAND, band); rows[B_DEPTH] = &ROW(SNOW_DEPTH_BAND, band);
      rows[B_CANOPY] = &ROW(SNOW_CANOPY_BAND, band); rows[B_MELT] = &ROW(SNOW_MELT_BAND, band); rows[B_COVER] = &ROW(SNOW_COVER_BAND, band);
      rows[B_DELTACC] = &ROW(DELTACC_BAND, band); rows[B_ADVECTION] = &ROW(ADVECTION_BAND, band); rows[B_SNOW_FLUX] = &ROW(SNOW_FLUX_BAND, band);
      rows[B_RFRZ] = &ROW(RFRZ_ENERGY_BAND, band); rows[B_MELT_ENERGY] = &ROW(MELT_ENERGY_BAND, band); rows[B_ADV_SENS] = &ROW(ADV_SENS_BAND, band);
      rows[B_SURFT] = &ROW(SNOW_SURFT_BAND, band); rows[B_PACKT] = &ROW(SNOW_PACKT_BAND, band); rows[B_LATENT_SUB] = &ROW(LATENT_SUB_BAND, band);
      rows[B_NET_SHORT] = &ROW(NET_SHORT_BAND, band); rows[B_NET_LONG] = &ROW(NET_LONG_BAND, band); rows[B_ALBEDO] = &ROW(ALBEDO_BAND, band);
      rows[B_LATENT] = &ROW(LATENT_BAND, band); rows[B_SENSIBLE] = &ROW(SENSIBLE_BAND, band); rows[B_GRND] = &ROW(GRND_FLUX_BAND, band);
#pragma unroll
      for (int i = 0; i < BN; i++) cur[i] = *rows[i];
      // += for the sums, -= for the four the reference subtracts (put_data.c:1163,1203-1211); SNOW_CANOPY_BAND only with vegetation
#pragma unroll
      for (int i = 0; i < BN; i++) {
        const bool minus = (i == B_ADV_SENS || i == B_LATENT || i == B_SENSIBLE || i == B_GRND);
        const bool skip = (i == B_CANOPY && !h.HasVeg);
        *rows[i] = skip ? cur[i] : (minus ? cur[i] - inc[i] : cur[i] + inc[i]);
      }
      ROW(ELEV_BAND, band) = (double)(float)cv.band(CPB_BANDELEV, band);
      if (h.HasGlac) {
        enum { G_DELTACC, G_FLUX, G_WAT, G_AREA, G_MBAL, G_IMBAL, G_ACCUM, G_MELT, G_SUB, G_IN, G_OUT, GN };
        double ginc[GN], gcur[GN];
        double* grow[GN];
        ginc[G_DELTACC] = FX(FX_DELTACC_GLAC); ginc[G_FLUX] = FX(FX_GLACIER_FLUX); ginc[G_WAT] = SD(SD_GLAC_WATER_STORAGE) * 1000.;
        ginc[G_AREA] = Cv; ginc[G_MBAL] = FX(FX_GLAC_MASS_BALANCE) * 1000.; ginc[G_IMBAL] = FX(FX_GLAC_ICE_MASS_BALANCE) * 1000.;
        ginc[G_ACCUM] = FX(FX_GLAC_ACCUMULATION) * 1000.; ginc[G_MELT] = FX(FX_GLAC_MELT) * 1000.; ginc[G_SUB] = FX(FX_GLAC_VAPOR_FLUX) * 1000.;
        ginc[G_IN] = FX(FX_GLAC_INFLOW) * 1000.; ginc[G_OUT] = FX(FX_GLAC_OUTFLOW) * 1000.;
        grow[G_DELTACC] = &ROW(GLAC_DELTACC_BAND, band); grow[G_FLUX] = &ROW(GLAC_FLUX_BAND, band); grow[G_WAT] = &ROW(GLAC_WAT_STOR_BAND, band);
        grow[G_AREA] = &ROW(GLAC_AREA_BAND, band); grow[G_MBAL] = &ROW(GLAC_MBAL_BAND, band); grow[G_IMBAL] = &ROW(GLAC_IMBAL_BAND, band);
        grow[G_ACCUM] = &ROW(GLAC_ACCUM_BAND, band); grow[G_MELT] = &ROW(GLAC_MELT_BAND, band); grow[G_SUB] = &ROW(GLAC_SUB_BAND, band);
        grow[G_IN] = &ROW(GLAC_INFLOW_BAND, band); grow[G_OUT] = &ROW(GLAC_OUTFLOW_BAND, band);
#pragma unroll
        for (int i = 0; i < GN; i++) gcur[i] = *grow[i];
#pragma unroll
        for (int i = 0; i < GN; i++) *grow[i] = gcur[i] + ginc[i];
      }
    }
  }
}

__global__ __launch_bounds__(64) void vic_put_sum(const OArgs a) {
  const int ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= a.ccount) return;
  const int c = a.c0 + ci;
  if (blockIdx.y == 0) put_sum_part<0>(a, c);
  else if (blockIdx.y == 1) put_sum_part<1>(a, c);
  else put_sum_part<2>(a, c);
}

__global__ __launch_bounds__(64) void vic_put_finish(const OArgs a) {
  const int ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= a.ccount) return;
  const int c = a.c0 + ci;
  const size_t nc = a.ncell;
  const Opt& o = a.o;
  const int NR = o.NR, ns = NR + 1;
  const double dt_sec = (double)o.dt * 3600.;
  double* od = a.out_data + c;
  // everything this kernel reads from the sums, before any store
  const double cv_baresoil = PB(PBX_CV_BARESOIL), cv_veg = PB(PBX_CV_VEG), cv_overstory = PB(PBX_CV_OVERSTORY), cv_snow = PB(PBX_CV_SNOW),
               cv_glacier = PB(PBX_CV_GLACIER);
  double baresoilt = ROW(BARESOILT, 0), vegt = ROW(VEGT, 0), ac2 = ROW(AERO_COND2, 0), salbedo = ROW(SALBEDO, 0), sst = ROW(SNOW_SURF_TEMP, 0),
         spt = ROW(SNOW_PACK_TEMP, 0), gst = ROW(GLAC_SURF_TEMP, 0);
  const double rad4 = ROW(RAD_TEMP, 0), ac1 = ROW(AERO_COND1, 0), ac = ROW(AERO_COND, 0);
  double liq[3], ice[3];
#pragma unroll
  for (int l = 0; l < 3; l++) { liq[l] = ROW(SOIL_LIQ, l); ice[l] = ROW(SOIL_ICE, l); }
  const double swe = ROW(SWE, 0), snow_canopy = ROW(SNOW_CANOPY, 0), wdew = ROW(WDEW, 0), surfstor = 0.0, rfrz = ROW(RFRZ_ENERGY, 0),
               net_short = ROW(NET_SHORT, 0), net_long = ROW(NET_LONG, 0), evap = ROW(EVAP, 0), runoff = ROW(RUNOFF, 0),
               baseflow = ROW(BASEFLOW, 0), glac_imbal = ROW(GLAC_IMBAL, 0), glac_wat = ROW(GLAC_WAT_STOR, 0), latent = ROW(LATENT, 0),
               latent_sub = ROW(LATENT_SUB, 0), sensible = ROW(SENSIBLE, 0), adv_sens = ROW(ADV_SENS, 0), grnd_flux = ROW(GRND_FLUX, 0),
               deltah = ROW(DELTAH, 0), fusion = ROW(FUSION, 0), advection = ROW(ADVECTION, 0), deltacc = ROW(DELTACC, 0),
               snow_flux = ROW(SNOW_FLUX, 0), glac_deltacc = ROW(GLAC_DELTACC, 0), glac_melt_energy = ROW(GLAC_MELT_ENERGY, 0);
  const double save_moist = PB(PB_SAVE_TOTAL_SOIL_MOIST), save_swe = PB(PB_SAVE_SWE), save_wdew = PB(PB_SAVE_WDEW),
               save_surfstor = PB(PB_SAVE_SURFSTOR), last_storage = PB(PB_WATER_LAST_STORAGE);
  double prec = 0;
  if (a.rec >= 0) {                                                                   // put_data.c:229-256
#define FV(var) a.forcing[((size_t)(var) * ns + NR) * nc + c]
    const double vp = FV(VIC_F_VP), vpd = FV(VIC_F_VPD), pr = FV(VIC_F_PRESSURE);
    prec = a.cell_out[(size_t)CO_OUT_PREC * nc + c];
    ROW(AIR_TEMP, 0) = FV(VIC_F_AIR_TEMP); ROW(DENSITY, 0) = FV(VIC_F_DENSITY); ROW(LONGWAVE, 0) = FV(VIC_F_LONGWAVE);
    ROW(PREC, 0) = prec;
    ROW(PRESSURE, 0) = pr / 1000.;
    ROW(QAIR, 0) = EPS_MW * vp / pr;
    ROW(RAINF, 0) = a.cell_out[(size_t)CO_OUT_RAIN * nc + c];
    ROW(REL_HUMID, 0) = 100. * vp / (vp + vpd);
    ROW(SHORTWAVE, 0) = FV(VIC_F_SHORTWAVE);
    ROW(SNOWF, 0) = a.cell_out[(size_t)CO_OUT_SNOW * nc + c];
    ROW(VP, 0) = vp / 1000.; ROW(VPD, 0) = vpd / 1000.; ROW(WIND, 0) = FV(VIC_F_WIND);
#undef FV
  }
  // ---- special cases and derived variables, put_data.c:549-606
  if (cv_baresoil > 0) { baresoilt /= cv_baresoil; ROW(BARESOILT, 0) = baresoilt; }
  if (cv_veg > 0) { vegt /= cv_veg; ROW(VEGT, 0) = vegt; }
  if (cv_overstory > 0) { ac2 /= cv_overstory; ROW(AERO_COND2, 0) = ac2; }
  if (cv_snow > 0) {
    salbedo /= cv_snow; sst /= cv_snow; spt /= cv_snow;
    ROW(SALBEDO, 0) = salbedo; ROW(SNOW_SURF_TEMP, 0) = sst; ROW(SNOW_PACK_TEMP, 0) = spt;
  }
  if (cv_glacier > 0) { gst /= cv_glacier; ROW(GLAC_SURF_TEMP, 0) = gst; }
  ROW(RAD_TEMP, 0) = pow(rad4, 0.25);
  const double ar1 = (ac1 > SMALL) ? 1 / ac1 : HUGE_RESIST, ar2 = (ac2 > SMALL) ? 1 / ac2 : HUGE_RESIST, ar = (ac > SMALL) ? 1 / ac : HUGE_RESIST;
  ROW(AERO_RESIST1, 0) = ar1; ROW(AERO_RESIST2, 0) = ar2; ROW(AERO_RESIST, 0) = ar;
  double delsoil = 0, liq_tot = 0, ice_tot = 0, moist_tot = 0, storage = 0.;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    liq_tot += liq[l]; ice_tot += ice[l];
    const double m = liq[l] + ice[l];
    ROW(SOIL_MOIST, l) = m;
    delsoil += m;
    const double lf = liq[l] / m;
    ROW(SMLIQFRAC, l) = lf;
    ROW(SMFROZFRAC, l) = 1 - lf;
    moist_tot += m;
    storage += liq[l] + ice[l];
  }
  ROW(SOIL_LIQ_TOT, 0) = liq_tot; ROW(SOIL_ICE_TOT, 0) = ice_tot;
  if (a.rec >= 0) {
    delsoil -= save_moist;
    ROW(DELSWE, 0) = swe + snow_canopy - save_swe;
    ROW(DELINTERCEPT, 0) = wdew - save_wdew;
    ROW(DELSURFSTOR, 0) = surfstor - save_surfstor;
  }
  ROW(DELSOILMOIST, 0) = delsoil;
  ROW(REFREEZE, 0) = (rfrz / LF) * dt_sec;
  ROW(R_NET, 0) = net_short + net_long;
  PB(PB_SAVE_TOTAL_SOIL_MOIST) = moist_tot;
  ROW(SOIL_MOIST_TOT, 0) = moist_tot;
  PB(PB_SAVE_SURFSTOR) = surfstor;
  PB(PB_SAVE_SWE) = swe + snow_canopy;
  PB(PB_SAVE_WDEW) = wdew;
  {                                                                                   // water balance, put_data.c:611-619
    const double inflow = prec + 0.;
    const double outflow = evap + runoff + baseflow;
    storage += swe + snow_canopy + wdew + surfstor + glac_wat;
    if (a.rec < 0) { PB(PB_WATER_LAST_STORAGE) = storage; PB(PB_WATER_CUM_ERROR) = 0.; PB(PB_WATER_MAX_ERROR) = 0.; ROW(WATER_ERROR, 0) = 0.0; }
    else {                                                                            // calc_water_balance_error
      const double error = inflow - outflow - (storage - last_storage) - glac_imbal;
      const double mx = PB(PB_WATER_MAX_ERROR);
      PB(PB_WATER_CUM_ERROR) += error;
      if (fabs(error) > fabs(mx) && fabs(error) > 1e-5) PB(PB_WATER_MAX_ERROR) = error;
      PB(PB_WATER_LAST_STORAGE) = storage;
      ROW(WATER_ERROR, 0) = error;
    }
  }
  if (o.FULL_ENERGY) {                                                                // put_data.c:624-633, calc_energy_balance_error
    if (a.rec < 0) { PB(PB_ENERGY_CUM_ERROR) = 0; PB(PB_ENERGY_MAX_ERROR) = 0; }
    else {
      const double net_rad = net_short + net_long, lat = latent + latent_sub, sens = sensible + adv_sens, grnd = grnd_flux + deltah + fusion,
                   snowf = advection - deltacc - snow_flux + rfrz, glacf = -glac_deltacc - glac_melt_energy;
      const double error = net_rad - lat - sens - grnd + snowf + glacf;
      const double mx = PB(PB_ENERGY_MAX_ERROR);
      PB(PB_ENERGY_CUM_ERROR) += error;
      if (fabs(error) > fabs(mx) && fabs(error) > 0.001) PB(PB_ENERGY_MAX_ERROR) = error;
    }
  }
  if (a.rec < 0) return;
  // the aggregates of the three resistances are the reciprocals of the aggregated conductances (put_data.c:683-685):
  // formed here from what vic_put_aggregate is about to store; it skips these three rows
  double* ag = a.out_agg + c;
#define AG(var) ag[(size_t)a.lay.off[VOUT_##var] * nc]
  const double g0 = AG(AERO_COND), g1 = AG(AERO_COND1), g2 = AG(AERO_COND2);
  AG(AERO_RESIST) = 1 / (g0 + ac / a.out_step_ratio);
  AG(AERO_RESIST1) = 1 / (g1 + ac1 / a.out_step_ratio);
  AG(AERO_RESIST2) = 1 / (g2 + ac2 / a.out_step_ratio);
#undef AG
}

// zero_output_list for the cells of the launch: lane = cell, blockIdx.y = a chunk of 32 rows
__global__ __launch_bounds__(64) void vic_put_zero(const OArgs a) {
  const int ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= a.ccount) return;
  const size_t nc = a.ncell;
  const int nrow = a.lay.off[VOUT_NVAR];
  double* __restrict__ od = a.out_data + a.c0 + ci;
  const int r0 = blockIdx.y * PUT_AGG_ROWS;
#pragma unroll
  for (int i = 0; i < PUT_AGG_ROWS; i++)
    if (r0 + i < nrow) od[(size_t)(r0 + i) * nc] = 0.0;
}

// temporal aggregation, put_data.c:663-682; rowagg[r] = aggregation type of row r's variable
__global__ __launch_bounds__(64) void vic_put_aggregate(const OArgs a, const unsigned char* __restrict__ rowagg) {
  const int ci = blockIdx.x * 64 + threadIdx.x;
  if (ci >= a.ccount) return;
  const int c = a.c0 + ci;
  const size_t nc = a.ncell;
  const int nrow = a.lay.off[VOUT_NVAR];
  const int r0 = blockIdx.y * PUT_AGG_ROWS;
  const double* __restrict__ od = a.out_data + c;
  double* __restrict__ ag = a.out_agg + c;
  double x[PUT_AGG_ROWS], y[PUT_AGG_ROWS];
#pragma unroll
  for (int i = 0; i < PUT_AGG_ROWS; i++) {
    const int r = r0 + i;
    if (r < nrow) { x[i] = od[(size_t)r * nc]; y[i] = ag[(size_t)r * nc]; }
  }
#pragma unroll
  for (int i = 0; i < PUT_AGG_ROWS; i++) {
    const int r = r0 + i;
    if (r < nrow) {
      const int kind = rowagg[r];
      if (kind == VOUT_AGG_END) ag[(size_t)r * nc] = x[i];
      else if (kind == VOUT_AGG_SUM) ag[(size_t)r * nc] = y[i] + x[i];
      else if (kind == VOUT_AGG_AVG) ag[(size_t)r * nc] = y[i] + x[i] / a.out_step_ratio;
    }
  }
}

#undef ROW
#undef PB
#undef SD
#undef SI
#undef FX

}  // namespace vic
