/*
 * orc.h — TEST INFRASTRUCTURE: types of the CPU oracle.
 *
 * oracle/ holds a plain-C restatement of the reference's per-HRU water/energy
 * balance step (pacificclimate/VIC: dist_prec -> full_energy -> surface_fluxes[_glac]
 * and everything below).  It is the checker for the HIP product and is never
 * imported, linked or executed by the product path (vic_amd/): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED against the real reference compiled here
 * (oracle/ref_build/build_ref.sh -> oracle/_ref/libvicref*.so; tests/test_oracle.py)
 * and against committed golden vectors generated from that build (tests/golden/).
 * The reference itself ships no golden outputs (SURVEY.md Finding 3).
 *
 * Scope restrictions mirror include/vicgpu.h: Nlayer 3, Ndist 1 (DIST_PRCP FALSE, mu = 1).
 */
#ifndef ORC_H_
#define ORC_H_

#include <math.h>
#include <limits.h>
#include <string.h>
#include "vicgpu.h"

/* ---- constants shared with the reference (vicNl_def.h:138-302, snow.h:34-79) ---- */
#define ORC_ERROR        (-999.0)
#define ORC_HUGE_RESIST  1.e20
#define ORC_SMALL        1.e-12
#define ORC_INVALID_INT  INT_MIN
#define ORC_ICE_DENSITY  917.0
#define ORC_VON_K        0.40
#define ORC_KELVIN       273.15
#define ORC_STEFAN_B     5.6696e-8
#define ORC_LF           3.337e5
#define ORC_RHO_W        999.842594
#define ORC_CP           1013.0
#define ORC_CH_ICE       2100.0e3
#define ORC_CH_WATER     4186.8e3
#define ORC_K_SNOW       2.9302e-6
#define ORC_EPS          0.62196351
#define ORC_G            9.81
#define ORC_JOULESPCAL   4.1868
#define ORC_GRAMSPKG     1000.0
#define ORC_SECPHOUR     3600
#define ORC_SEC_PER_DAY  86400.
#define ORC_GLAC_TEMP    0.0
#define ORC_GLAC_K_ICE   2.14
#define ORC_SNOW_SURF_DENSITY 350
#define ORC_CUTOFF_DENSITY    830
#define ORC_SNOW_DT      5.0
#define ORC_SURF_DT      1.0
#define ORC_SOIL_DT      0.25
#define ORC_COEF_DRAG    0.2
#define ORC_LIQUID_WATER_CAPACITY    0.035
#define ORC_LAI_SNOW_MULTIPLIER      0.0005
#define ORC_MIN_INTERCEPTION_STORAGE 0.005
#define ORC_MAX_SURFACE_SWE          0.125
#define ORC_NEW_SNOW_DENSITY         50.
#define ORC_SNDENS_DMLIMIT 100.
#define ORC_SNDENS_ETA0    (3.6e6)
#define ORC_SNDENS_C1      0.04
#define ORC_SNDENS_C2      (2.778e-6)
#define ORC_SNDENS_C5      0.08
#define ORC_SNDENS_C6      0.021
#define ORC_SNDENS_F       0.6
#define ORC_MIN_SWQ_EB_THRES 0.0010
#define ORC_TRACESNOW        0.03

#define ORC_NPET 6
#define ORC_NPET_NON_NAT 4
#define ORC_PET_VEGNOCR 5

/* the four surface cases of VegConditions (VegConditions.h:4-20) */
enum { ORC_SNOW_FREE = 0, ORC_CANOPY = 1, ORC_SNOW_COVERED = 2, ORC_GLACIER_SURF = 3, ORC_NCASE = 4 };
typedef struct { double v[ORC_NCASE]; } orc_vc;

static inline int orc_is_error(double x) { return x <= -998.0; }   /* RootBrent::resultIsError, root_brent.h */

/* ---- per-cell parameters (soil_con_struct subset) ---- */
typedef struct {
  double Ds, Dsmax, Ws, c, b_infilt, dp, avg_temp, rough, snow_rough, elevation, lat;
  int FS_ACTIVE;
  double NEW_SNOW_ALB, SNOW_ALB_ACCUM_A, SNOW_ALB_ACCUM_B, SNOW_ALB_THAW_A, SNOW_ALB_THAW_B;
  double MIN_RAIN_TEMP, MAX_SNOW_TEMP, PADJ_R, PADJ_S;
  double GLAC_SURF_THICK, GLAC_SURF_WE, GLAC_KMIN, GLAC_DK, GLAC_A, GLAC_ALBEDO, GLAC_ROUGH;
  double Ksat[3], Wcr[3], Wpwp[3], expt[3], bubble[3], depth[3], max_moist[3], resid_moist[3], porosity[3],
         quartz[3], organic[3], bulk_density[3], soil_density[3], bulk_dens_min[3], soil_dens_min[3];
  double Zsum_node[VIC_MAX_NODES], dz_node[VIC_MAX_NODES], alpha[VIC_MAX_NODES], beta[VIC_MAX_NODES],
         gamma[VIC_MAX_NODES], max_moist_node[VIC_MAX_NODES], expt_node[VIC_MAX_NODES], bubble_node[VIC_MAX_NODES];
  double AreaFract[VIC_MAX_BANDS], Tfactor[VIC_MAX_BANDS], Pfactor[VIC_MAX_BANDS], BandElev[VIC_MAX_BANDS];
  int AboveTreeLine[VIC_MAX_BANDS];     /* COMPUTE_TREELINE result (read by put_data only) */
  double zwt_zwt[VIC_NLAYER + 2][VIC_MAX_ZWTVMOIST], zwt_moist[VIC_NLAYER + 2][VIC_MAX_ZWTVMOIST];
} orc_soil;

/* ---- one record of forcing for one cell, [NF+1] each ---- */
#define ORC_MAX_SUB 25
typedef struct {
  double air_temp[ORC_MAX_SUB], prec[ORC_MAX_SUB], pressure[ORC_MAX_SUB], vp[ORC_MAX_SUB], vpd[ORC_MAX_SUB],
         density[ORC_MAX_SUB], shortwave[ORC_MAX_SUB], longwave[ORC_MAX_SUB], wind[ORC_MAX_SUB];
  int snowflag[ORC_MAX_SUB];
  double out_prec, out_rain, out_snow;
  double gauge_correction[2];   /* [0] rain, [1] snow: set by orc_full_energy (full_energy.c:188-194) */
} orc_atmos;

typedef struct { int month, day_in_year, hour, day, year; } orc_dmy;

/* ---- HRU state ---- */
typedef struct { double moist, ice, T, evap, kappa, Cs, zwt; } orc_layer;

typedef struct {
  double AlbedoOver, AlbedoUnder;
  double Cs[2], kappa[2];
  double T[VIC_MAX_NODES], moist[VIC_MAX_NODES], ice[VIC_MAX_NODES], kappa_node[VIC_MAX_NODES], Cs_node[VIC_MAX_NODES];
  int T_fbflag[VIC_MAX_NODES], T_fbcount[VIC_MAX_NODES];
  double fdepth[3], tdepth[3];
  int frozen, Nfrost, Nthaw;
  double Tcanopy, Tfoliage, Tsurf;
  int Tcanopy_fbflag, Tcanopy_fbcount, Tfoliage_fbflag, Tfoliage_fbcount, Tsurf_fbflag, Tsurf_fbcount;
  double advected_sensible, advection, AtmosError, AtmosLatent, AtmosLatentSub, AtmosSensible;
  double canopy_advection, canopy_latent, canopy_latent_sub, canopy_refreeze, canopy_sensible;
  double deltaCC, deltaH, error, fusion, grnd_flux, latent, latent_sub;
  double LongOverIn, LongUnderIn, LongUnderOut, melt_energy;
  double NetLongAtmos, NetLongOver, NetLongUnder, NetShortAtmos, NetShortGrnd, NetShortOver, NetShortUnder;
  double refreeze_energy, sensible, ShortOverIn, ShortUnderIn, snow_flux;
  double glacier_flux, deltaCC_glac, glacier_melt_energy;
} orc_energy;

typedef struct {
  double albedo, coldcontent, coverage, density, depth;
  int last_snow, MELTING;
  double max_swq, pack_temp, pack_water;
  int snow;
  double snow_canopy, store_coverage;
  int store_snow;
  double store_swq, surf_temp;
  int surf_temp_fbcount, surf_temp_fbflag;
  double surf_water, swq, swq_slope, tmp_int_storage;
  double blowing_flux, canopy_vapor_flux, mass_error, melt, Qnet, surface_flux, transport, vapor_flux;
} orc_snow;

typedef struct { double canopyevap, throughfall, Wdew; } orc_vegvar;

typedef struct {
  double cold_content, surf_temp;
  int surf_temp_fbcount, surf_temp_fbflag;
  double Qnet, mass_balance, ice_mass_balance, cum_mass_balance, accumulation, melt, vapor_flux,
         water_storage, outflow, outflow_coef, inflow;
} orc_glac;

typedef struct orc_hru_s {
  /* parameters */
  int cell, band, veg_index, veg_class, is_glacier, is_artificial_bare;
  double Cv, root[3];
  float sigma_slope, lag_one, fetch;       /* veg_con (vicNl_def.h:1023-1027): blowing snow only */
  /* state + per-step outputs */
  orc_layer layer[3];
  orc_energy energy;
  orc_snow snow;
  orc_vegvar veg;
  orc_glac glac;
  double aero_resist_surface, aero_resist_overstory;
  double asat, baseflow, inflow, runoff, excess_moist, pot_evap[ORC_NPET], rootmoist, wetness, zwt, zwt2, zwt3;
  double out_prec, out_rain, out_snow;
} orc_hru;

typedef struct {
  vicgpu_options opt;
  int NF, NR;
  int nveg_rows;
  const double *veglib;   /* [nveg_rows][VL_NFIELD] */
  /* stopping tolerance 2*macheps*|b| + ttol of the frozen-NODE root finds only (root_brent.c:32-36,274: 3e-8, 1e-7).
   * A test-only knob (vicorc_set_node_tolerance): tightened, the restatement converges those roots fully, which separates
   * the reference's own stopping error from implementation error when the product's Newton node solver is checked. */
  double node_macheps, node_ttol;
  long implicit_ok, implicit_failed;   /* func_surf_energy_bal.c:198-202 error_cnt0 / error_cnt1 (test statistics; racy under OpenMP, only > 0 is asserted) */
} orc_model;

/* handle behind the vicorc_* entry points (orc_driver.c, orc_putdata.c) */
typedef struct {
  orc_model model;
  int ncell, nhru;
  double *veglib;
  orc_soil *soil;
  struct orc_hru_s *hru;
  int *cell_off, *cell_list;
  /* put_data (orc_putdata.c): OutputData.data / .aggdata of every provided variable, [row][cell], and the per-cell
   * bookkeeping rows of include/vicgpu_out.h */
  double *out_data, *out_agg, *pb;
} vicorc_handle;

static inline const double *orc_veg(const orc_model *m, int idx) { return m->veglib + (size_t)idx * VL_NFIELD; }

/* ---- orc_base.c ---- */
double orc_svp(double T);
/* orc_blowing.c: CalcBlowingSnow.c:101-310 */
double orc_calc_blowing_snow(double Dt, double Tair, int LastSnow, double SurfaceLiquidWater, double Wind, double Ls, double AirDens,
                             double EactAir, double ZO, double Zrh, double snowdepth, float lag_one, float sigma_slope,
                             double Tsnow, int isArtificialBareSoil, float fe, double displacement, double roughness,
                             double *TotalTransport);
double orc_svp_slope(double T);
double orc_calc_rc(double rs, double net_short, float RGL, double tair, double vpd, double lai, double gsm_inv, int ref_crop);
double orc_penman(double tair, double elevation, double rad, double vpd, double ra, double rc, double rarc);
double orc_stability_correction(double Z, double d, double TSurf, double Tair, double Wind, double Z0);
typedef double (*orc_fn)(double x, void *ctx);
double orc_root_brent(double lower, double upper, orc_fn f, void *ctx);
double orc_root_brent_tol(double lower, double upper, orc_fn f, void *ctx, double MACHEPS, double TTOL);
double orc_calc_veg_height(double displacement, double L);
int    orc_calc_aerodynamic(int overstory, double height, double trunk, double z0_snow, double z0_soil, double n,
                            orc_vc *aero_resist, orc_vc *wind_speed, orc_vc *displacement, orc_vc *ref_height, orc_vc *roughness);
double orc_soil_conductivity(double moist, double Wu, double soil_dens_min, double bulk_dens_min, double quartz,
                             double soil_density, double bulk_density, double organic);
double orc_volumetric_heat_capacity(double soil_fract, double water_fract, double ice_fract, double organic_fract);
double orc_maximum_unfrozen_water(double T, double max_moist, double bubble, double expt);
double orc_linear_interp(double x, double lx, double ux, double ly, double uy);
void   orc_layer_thermal_properties(orc_layer *layer, const orc_soil *sc);
int    orc_distribute_node_moisture_properties(const orc_model *m, orc_energy *e, const orc_soil *sc, const double *moist);
int    orc_estimate_layer_ice_content(const orc_model *m, orc_layer *layer, const double *T, const orc_soil *sc);
void   orc_estimate_layer_ice_content_quick_flux(const orc_model *m, orc_layer *layer, double Tsurf, double T1, const orc_soil *sc);
void   orc_find_0_degree_fronts(orc_energy *e, const double *Zsum, const double *T, int Nnodes);
void   orc_wrap_compute_zwt(const orc_soil *sc, orc_hru *h);
void   orc_compute_runoff_and_asat(const orc_soil *sc, const double *moist, double inflow, double *A, double *runoff);
int    orc_runoff(const orc_model *m, orc_hru *h, const orc_soil *sc, double ppt);
double orc_canopy_evap(const orc_model *m, orc_layer *layer, orc_vegvar *vv, int calc_evap, int veg_idx, int month,
                       double *Wdew, double delta_t, double rad, double vpd, double net_short, double air_temp, double ra,
                       double elevation, double ppt, const orc_soil *sc, const double *root);
double orc_arno_evap(orc_layer *layer, double rad, double air_temp, double vpd, double depth1, double max_moist,
                     double elevation, double b_infilt, double ra, double delta_t, double moist_resid);
void   orc_compute_pot_evap(const orc_model *m, int veg_idx, int month, int dt, double shortwave, double net_longwave,
                            double tair, double vpd, double elevation, const double *ra_surface, const double *ra_overstory,
                            double *pot_evap);

/* ---- orc_snow.c ---- */
double orc_calc_rainonly(const orc_model *m, double air_temp, double prec, double MAX_SNOW_TEMP, double MIN_RAIN_TEMP);
double orc_solve_snow(const orc_model *m, int overstory, double BareAlbedo, double LongUnderOut, double Tcanopy, double Tgrnd,
                      double air_temp, double prec, double snow_grnd_flux, double *AlbedoUnder, double *Le,
                      double *LongUnderIn, double *NetLongSnow, double *NetShortGrnd, double *NetShortSnow,
                      double *ShortUnderIn, double *Torg_snow, orc_vc *aero_resist, double *ra_used /*[2]*/,
                      double *coverage, double *delta_coverage, orc_vc *displacement, double *melt_energy,
                      double *out_prec, double *out_rain, double *out_snow, double *ppt, double *rainfall,
                      orc_vc *ref_height, orc_vc *roughness, double *snow_inflow, double *snowfall, double *surf_atten,
                      orc_vc *wind_speed, const double *root, int UNSTABLE_SNOW, int dt, int hidx, int veg_idx,
                      int is_artificial_bare, int *UnderStory, const orc_dmy *dmy, const orc_atmos *atmos,
                      orc_energy *energy, orc_layer *layer, orc_snow *snow, const orc_soil *sc, orc_vegvar *vv);

/* ---- orc_surface.c ---- */
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
                                const orc_soil *sc, orc_vegvar *vv);

/* ---- orc_glacier.c ---- */
int orc_surface_fluxes_glac(const orc_model *m, orc_hru *h, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy,
                            double BareAlbedo, double ice0, double moist0, orc_vc *aero_resist /*[7]*/, orc_vc *displacement,
                            orc_vc *ref_height, orc_vc *roughness, orc_vc *wind_speed, double *out_prec, double *out_rain,
                            double *out_snow);

/* ---- orc_driver.c ---- */
int orc_full_energy(const orc_model *m, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy, orc_hru **hrus, int nhru);

#endif /* ORC_H_ */
