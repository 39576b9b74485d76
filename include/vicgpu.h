/*
 * vicgpu.h — C-ABI boundary of the MI355X-native VIC hot path.
 *
 * What this replaces.  The reference (pacificclimate/VIC) has no plugin/FFI
 * interface; its seam is the body of the OpenMP loop over cells in
 * vicNl.c:514-593, i.e. per cell and per record
 *
 *     dist_prec(&cell, dmy, &filep, outputFormat, outputData, rec, FALSE, state)   vicNl.c:543
 *       -> full_energy(NEWCELL, rec, atmos, prcp, dmy, lake_con, soil_con, ...)     dist_prec.c:159, full_energy.c:8
 *            -> surface_fluxes / surface_fluxes_glac per HRU                        full_energy.c:399-424
 *
 * The entry points below are what a reference-side binding
 * calls instead of that loop body, once for ALL cells of a domain: plain
 * pointers and sizes, caller-owned host buffers, int return codes
 * (0 = ok, negative = error; VICGPU_ERR_*), never throws, never exits.
 *
 * Data contract.  Everything is struct-of-arrays ("tables"): a table is a
 * dense row-major double (or int) array [nrow][ncol] where the column is the
 * cell (or the HRU) and the row is a field listed in the enums below.  One
 * HRU = one vegetation tile in one snow band of one cell (struct HRU,
 * vicNl_def.h:1374-1388).  All arithmetic is fp64 like the reference.
 *
 * Option coverage: Nlayer = 3, DIST_PRCP = FALSE
 * (Ndist = 1, mu = 1), no lakes, no EXCESS_ICE / SPATIAL_FROST / SPATIAL_SNOW /
 * QUICK_FS / LOW_RES_MOIST / CLOSE_ENERGY (all compiled out in the reference,
 * user_def.h:36-92).  CORRPREC, BLOWING, IMPLICIT (finite-difference soil profile, node-array
 * freezing parameters: frozen_compat = 0) and QUICK_SOLVE (with NOFLUX / EXP_TRANS as the reference handles them)
 * are implemented.  What the device code does not implement (see vicgpu_create) is
 * REJECTED with VICGPU_ERR_UNSUPPORTED, never silently replaced.
 */
#ifndef VICGPU_H_
#define VICGPU_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VICGPU_ABI_VERSION 3

#define VIC_NLAYER        3    /* MAX_LAYERS, user_def.h:95 */
#define VIC_MAX_NODES    24    /* device build limit for options.Nnode (reference MAX_NODES = 50, user_def.h:96) */
#define VIC_MAX_BANDS    30    /* MAX_BANDS, user_def.h:97 */
#define VIC_N_PET_TYPES   6    /* vicNl_def.h:213 */
#define VIC_MAX_ZWTVMOIST 11   /* user_def.h:100 */

/* error codes */
#define VICGPU_OK               0
#define VICGPU_ERR_ARG         -1
#define VICGPU_ERR_UNSUPPORTED -2
#define VICGPU_ERR_HIP         -3
#define VICGPU_ERR_STATE       -4
#define VICGPU_ERR_NOMEM       -5

/* option enumerations (values follow vicNl_def.h:166-210) */
#define VIC_SNOW_ALBEDO_USACE   0
#define VIC_SNOW_ALBEDO_SUN1999 1
#define VIC_DENS_BRAS   0
#define VIC_DENS_SNTHRM 1
#define VIC_AR_406      0
#define VIC_AR_406_LS   1
#define VIC_AR_406_FULL 2
#define VIC_AR_410      3
#define VIC_AR_COMBO    4
#define VIC_GF_406  0
#define VIC_GF_410  1
#define VIC_GF_FULL 2
#define VIC_NODE_SOLVER_BRENT  0
#define VIC_NODE_SOLVER_NEWTON 1
#define VIC_TEMP_TH_VIC_412 0
#define VIC_TEMP_TH_KIENZLE 1

/* Snapshot of the option_struct / global_param_struct fields the hot path reads
 * (vicNl_def.h:655-790, 854-895). */
typedef struct vicgpu_options {
  int abi_version;          /* must be VICGPU_ABI_VERSION */
  int Nlayer;               /* must be 3 */
  int Nnode;                /* 3 with QUICK_FLUX; <= VIC_MAX_NODES */
  int Nband;                /* options.SNOW_BAND */
  int dt;                   /* global_param.dt, hours */
  int snow_step;            /* options.SNOW_STEP, hours (== dt when dt < 24, get_global_param.c:965) */
  int FULL_ENERGY;
  int FROZEN_SOIL;
  int QUICK_FLUX;
  int NOFLUX;
  int EXP_TRANS;
  int GRND_FLUX_TYPE;       /* VIC_GF_* */
  int TFALLBACK;
  int AERO_RESIST_CANSNOW;  /* VIC_AR_* */
  int SNOW_ALBEDO;
  int SNOW_DENSITY;
  int TEMP_TH_TYPE;
  int GLACIER_ID;           /* veg_class number of the glacier class, -1 = none */
  int GLACIER_DYNAMICS;
  int frozen_compat;        /* 1 = reproduce frozen_soil.c:218-221 (layer arrays indexed by node,
                               SURVEY.md Finding 1.2); 0 = node arrays ("fixed") */
  int nveg_types;           /* veg_lib[0].NVegLibTypes; the table holds nveg_types + 4 rows */
  int CORRPREC;             /* gauge-undercatch correction of precipitation (correct_precip.c, full_energy.c:188-194) */
  int IMPLICIT;             /* options.IMPLICIT: Newton-Raphson soil heat solver (frozen_soil.c:229-301, newt_raph_func_fast.c),
                               the explicit solver as its fallback; rejected with QUICK_FLUX or frozen_compat */
  int BLOWING;              /* options.BLOWING: sublimation from blowing snow (CalcBlowingSnow.c), once per snow sub-step */
  int QUICK_SOLVE;          /* options.QUICK_SOLVE (calc_surf_energy_bal.c:289-314, 400-475): ignored with QUICK_FLUX (as in the
                               reference); the iteration runs with NOFLUX and EXP_TRANS forced off, NOFLUX comes back with a second
                               iteration only, EXP_TRANS never (calc_surf_energy_bal.c:298-308, 403); rejected together with IMPLICIT */
  int NODE_SOLVER;          /* VIC_NODE_SOLVER_*: how the frozen-node heat balance (soil_thermal_eqn.c) is solved -- not a
                               reference option; BRENT replays root_brent.c's iteration, NEWTON converges to the same root */
  double wind_h;            /* global_param.wind_h (m) */
  double reserved_d[3];
} vicgpu_options;

/* NF / NR as get_global_param.c:964-973 derives them */
#define VICGPU_NF(opt) ((opt)->dt / (opt)->snow_step == 1 ? 1 : (opt)->dt / (opt)->snow_step)
#define VICGPU_NR(opt) ((opt)->dt / (opt)->snow_step == 1 ? 0 : (opt)->dt / (opt)->snow_step)

/* ---------------------------------------------------------------- veg library
 * veg_lib_struct (vicNl_def.h:1034-1051): AoS table double[nveg_types+4][VL_NFIELD];
 * the last 4 rows are the reference PET surfaces read_veglib.c:118-136 appends. */
enum {
  VL_OVERSTORY = 0, VL_RARC, VL_RMIN, VL_RAD_ATTEN, VL_TRUNK_RATIO, VL_WIND_ATTEN,
  VL_WIND_H, VL_RGL, VL_VEG_CLASS,
  VL_LAI = 9,            /* 12 monthly values each from here on */
  VL_WDMAX = 21,
  VL_ALBEDO = 33,
  VL_DISPLACEMENT = 45,
  VL_EMISSIVITY = 57,
  VL_ROUGHNESS = 69,
  VL_NFIELD = 81
};

/* ---------------------------------------------------------------- forcing
 * atmos_data_struct (vicNl_def.h:1060-1076).  Units as inside the reference's
 * time loop: T in C, prec mm/step, pressure/vp/vpd in Pa, W/m2, kg/m3, m/s.
 * Layout double[nsteps][VIC_NFORCE][NF+1][ncell]; sub-index NR holds the
 * step average, 0..NF-1 the snow sub-steps.  snowflag uint8[nsteps][NF+1][ncell]. */
enum {
  VIC_F_AIR_TEMP = 0, VIC_F_PREC, VIC_F_PRESSURE, VIC_F_VP, VIC_F_VPD,
  VIC_F_DENSITY, VIC_F_SHORTWAVE, VIC_F_LONGWAVE, VIC_F_WIND, VIC_NFORCE
};

/* The same forcing as the forcing files hold it, hour by hour, before initialize_atmos.c derives atmos[rec] from it:
 * double[nsteps][VIC_NRAW][dt][ncell] (dt = hours per model step), pressure and vapour pressure in kPa
 * (initialize_atmos.c:290-295).  vicgpu_prefetch_forcing_raw derives the table above from it on the device. */
enum {
  VIC_RAW_AIR_TEMP = 0, VIC_RAW_PREC, VIC_RAW_PRESSURE_KPA, VIC_RAW_VP_KPA, VIC_RAW_SHORTWAVE, VIC_RAW_LONGWAVE, VIC_RAW_WIND,
  VIC_NRAW
};

/* dmy_struct (vicNl_def.h:1082-1088): int[nsteps][VIC_NDMY] */
enum { VIC_DMY_MONTH = 0, VIC_DMY_DAY_IN_YEAR, VIC_DMY_HOUR, VIC_DMY_DAY, VIC_DMY_YEAR, VIC_NDMY };

/* ---------------------------------------------------------------- cell parameters
 * soil_con_struct (vicNl_def.h:900-999) fields read by the path.
 * Table double[VICGPU_CP_NROW(Nnode,Nband)][ncell]. */
enum {
  CP_DS = 0, CP_DSMAX, CP_WS, CP_C, CP_B_INFILT, CP_DP, CP_AVG_TEMP, CP_ROUGH, CP_SNOW_ROUGH,
  CP_ELEVATION, CP_LAT, CP_FS_ACTIVE,
  CP_NEW_SNOW_ALB, CP_SNOW_ALB_ACCUM_A, CP_SNOW_ALB_ACCUM_B, CP_SNOW_ALB_THAW_A, CP_SNOW_ALB_THAW_B,
  CP_MIN_RAIN_TEMP, CP_MAX_SNOW_TEMP, CP_PADJ_R, CP_PADJ_S,
  CP_GLAC_SURF_THICK, CP_GLAC_SURF_WE, CP_GLAC_KMIN, CP_GLAC_DK, CP_GLAC_A, CP_GLAC_ALBEDO, CP_GLAC_ROUGH,
  CP_NSCALAR
};
/* per-layer fields (x VIC_NLAYER) */
enum {
  CPL_KSAT = 0, CPL_WCR, CPL_WPWP, CPL_EXPT, CPL_BUBBLE, CPL_DEPTH, CPL_MAX_MOIST, CPL_RESID_MOIST,
  CPL_POROSITY, CPL_QUARTZ, CPL_ORGANIC, CPL_BULK_DENSITY, CPL_SOIL_DENSITY, CPL_BULK_DENS_MIN,
  CPL_SOIL_DENS_MIN, CPL_NFIELD
};
/* per-node fields (x Nnode) */
enum {
  CPN_ZSUM = 0, CPN_DZ, CPN_ALPHA, CPN_BETA, CPN_GAMMA, CPN_MAX_MOIST, CPN_EXPT, CPN_BUBBLE, CPN_NFIELD
};
/* per-band fields (x Nband) */
enum { CPB_AREAFRACT = 0, CPB_TFACTOR, CPB_PFACTOR, CPB_BANDELEV, CPB_ABOVETREELINE, CPB_NFIELD };
/* water-table curves zwtvmoist_zwt / zwtvmoist_moist [Nlayer+2][MAX_ZWTVMOIST] (vicNl_def.h:965-966) */
#define VIC_NZWT_ROWS ((VIC_NLAYER + 2) * VIC_MAX_ZWTVMOIST)

#define VICGPU_CP_LAYER(f, l)            (CP_NSCALAR + (f) * VIC_NLAYER + (l))
#define VICGPU_CP_NODE0                  (CP_NSCALAR + CPL_NFIELD * VIC_NLAYER)
#define VICGPU_CP_NODE(f, n, Nn)         (VICGPU_CP_NODE0 + (f) * (Nn) + (n))
#define VICGPU_CP_BAND0(Nn)              (VICGPU_CP_NODE0 + CPN_NFIELD * (Nn))
#define VICGPU_CP_BAND(f, b, Nn, Nb)     (VICGPU_CP_BAND0(Nn) + (f) * (Nb) + (b))
#define VICGPU_CP_ZWT0(Nn, Nb)           (VICGPU_CP_BAND0(Nn) + CPB_NFIELD * (Nb))
#define VICGPU_CP_ZWT_ZWT(l, i, Nn, Nb)  (VICGPU_CP_ZWT0(Nn, Nb) + (l) * VIC_MAX_ZWTVMOIST + (i))
#define VICGPU_CP_ZWT_MOIST(l, i, Nn, Nb) (VICGPU_CP_ZWT0(Nn, Nb) + VIC_NZWT_ROWS + (l) * VIC_MAX_ZWTVMOIST + (i))
#define VICGPU_CP_NROW(Nn, Nb)           (VICGPU_CP_ZWT0(Nn, Nb) + 2 * VIC_NZWT_ROWS)

/* ---------------------------------------------------------------- HRU parameters
 * veg_con_struct + HRU meta (vicNl_def.h:1017-1029, 1374-1388).
 * int table int[HPI_NROW][nhru], double table double[HPD_NROW][nhru]. */
enum { HPI_CELL = 0, HPI_BAND, HPI_VEG_INDEX, HPI_VEG_CLASS, HPI_IS_GLACIER, HPI_IS_ARTIFICIAL_BARE, HPI_NROW };
/* HPD_SIGMA_SLOPE, HPD_LAG_ONE, HPD_FETCH: veg_con.sigma_slope / lag_one / fetch (float in the reference, read_vegparam.c),
 * read by the blowing-snow model only (options.BLOWING) */
enum { HPD_CV = 0, HPD_ROOT0, HPD_ROOT1, HPD_ROOT2, HPD_SIGMA_SLOPE, HPD_LAG_ONE, HPD_FETCH, HPD_NROW };

/* ---------------------------------------------------------------- HRU state
 * SURVEY.md Appendix A.  double[VICGPU_SD_NROW(Nnode)][nhru], int[SI_NROW][nhru].
 * Rows up to SD_NPROG-1 plus the node block are prognostic (they steer the next
 * step); rows SD_NPROG.. are "sticky" diagnostics of energy_bal_struct that the
 * reference carries from step to step because surface_fluxes.c:301-302 seeds its
 * scratch copies from last step's struct (they only influence outputs). */
enum {
  /* soil layers (layer_data_struct, vicNl_def.h:1094-1107) */
  SD_MOIST0 = 0, SD_MOIST1, SD_MOIST2,
  SD_ICE0, SD_ICE1, SD_ICE2,
  SD_LAYER_T0, SD_LAYER_T1, SD_LAYER_T2,
  /* energy_bal_struct scalars that feed the next step */
  SD_SNOW_FLUX, SD_GRND_FLUX, SD_DELTAH, SD_FUSION, SD_LONGUNDEROUT, SD_TFOLIAGE,
  /* snow_data_struct (vicNl_def.h:1223-1257) */
  SD_SNOW_ALBEDO, SD_SNOW_COLDCONTENT, SD_SNOW_COVERAGE, SD_SNOW_DENSITY, SD_SNOW_DEPTH,
  SD_SNOW_PACK_TEMP, SD_SNOW_PACK_WATER, SD_SNOW_CANOPY, SD_SNOW_SURF_TEMP, SD_SNOW_SURF_WATER,
  SD_SNOW_SWQ, SD_SNOW_TMP_INT_STORAGE, SD_SNOW_STORE_SWQ, SD_SNOW_STORE_COVERAGE, SD_SNOW_SWQ_SLOPE,
  SD_SNOW_MAX_SWQ,
  /* veg_var_struct */
  SD_WDEW,
  /* glac_data_struct (vicNl_def.h:1341-1365) */
  SD_GLAC_SURF_TEMP, SD_GLAC_WATER_STORAGE, SD_GLAC_CUM_MASS_BALANCE,
  SD_NPROG,
  /* sticky diagnostics */
  SD_TCANOPY = SD_NPROG, SD_TSURF, SD_ALBEDO_OVER, SD_ALBEDO_UNDER,
  SD_CANOPY_ADVECTION, SD_CANOPY_LATENT, SD_CANOPY_LATENT_SUB, SD_CANOPY_SENSIBLE, SD_CANOPY_REFREEZE,
  SD_ADVECTED_SENSIBLE, SD_ADVECTION, SD_DELTACC, SD_REFREEZE_ENERGY, SD_MELT_ENERGY, SD_ERROR,
  SD_LATENT, SD_LATENT_SUB, SD_SENSIBLE,
  SD_LONGOVERIN, SD_NETLONGOVER, SD_NETSHORTOVER, SD_SHORTOVERIN,
  SD_NETLONGUNDER,   /* glacier HRUs feed last step's NetLongUnder into compute_pot_evap (surface_fluxes_glac.c:343,380) */
  SD_NSCALAR
};
/* per-node fields (x Nnode): T is prognostic; the other four are what
 * distribute_node_moisture_properties (runoff.c:763) leaves for the next step */
enum { SDN_T = 0, SDN_MOIST, SDN_ICE, SDN_KAPPA, SDN_CS, SDN_NFIELD };
#define VICGPU_SD_NODE(f, n, Nn) (SD_NSCALAR + (f) * (Nn) + (n))
#define VICGPU_SD_NROW(Nn)       (SD_NSCALAR + SDN_NFIELD * (Nn))

enum {
  SI_SNOW_LAST_SNOW = 0,   /* may be INT_MIN = INVALID_INT (vicNl_def.h:151) */
  SI_SNOW_MELTING, SI_SNOW_SNOW, SI_SNOW_STORE_SNOW,
  SI_SNOW_SURF_TEMP_FBCOUNT, SI_SNOW_SURF_TEMP_FBFLAG,
  SI_TSURF_FBCOUNT, SI_TSURF_FBFLAG, SI_TFOLIAGE_FBCOUNT, SI_TFOLIAGE_FBFLAG,
  SI_TCANOPY_FBCOUNT, SI_TCANOPY_FBFLAG,
  SI_GLAC_SURF_TEMP_FBCOUNT, SI_GLAC_SURF_TEMP_FBFLAG,
  SI_FROZEN, SI_NFROST, SI_NTHAW,
  SI_NSCALAR
};
/* per-node ints (x Nnode): T_fbflag, T_fbcount (vicNl_def.h:1152-1153) */
enum { SIN_T_FBFLAG = 0, SIN_T_FBCOUNT, SIN_NFIELD };
#define VICGPU_SI_NODE(f, n, Nn) (SI_NSCALAR + (f) * (Nn) + (n))
#define VICGPU_SI_NROW(Nn)       (SI_NSCALAR + SIN_NFIELD * (Nn))

/* ---------------------------------------------------------------- per-step HRU fluxes
 * What put_data (put_data.c:762-1232) reads from an HRU after dist_prec.
 * double[FX_NROW][nhru], overwritten every step. */
enum {
  FX_RUNOFF = 0, FX_BASEFLOW, FX_ASAT, FX_INFLOW,
  FX_EVAP0, FX_EVAP1, FX_EVAP2,            /* layer[l].evap, mm */
  FX_CANOPYEVAP, FX_THROUGHFALL,
  FX_SNOW_VAPOR_FLUX, FX_SNOW_CANOPY_VAPOR_FLUX, FX_SNOW_BLOWING_FLUX, FX_SNOW_SURFACE_FLUX,
  FX_SNOW_MELT, FX_SNOW_MASS_ERROR, FX_SNOW_QNET,
  FX_POT_EVAP0, FX_POT_EVAP1, FX_POT_EVAP2, FX_POT_EVAP3, FX_POT_EVAP4, FX_POT_EVAP5,
  FX_AERO_RESIST_SURFACE, FX_AERO_RESIST_OVERSTORY,
  FX_ROOTMOIST, FX_WETNESS,
  FX_ZWT, FX_ZWT2, FX_ZWT3,
  /* energy_bal_struct step averages (surface_fluxes.c:842-881) */
  FX_ATMOS_LATENT, FX_ATMOS_LATENT_SUB, FX_ATMOS_SENSIBLE,
  FX_LONG_UNDER_IN, FX_NET_LONG_ATMOS, FX_NET_LONG_UNDER, FX_NET_SHORT_ATMOS, FX_NET_SHORT_GRND,
  FX_NET_SHORT_UNDER, FX_SHORT_UNDER_IN,
  FX_OUT_PREC, FX_OUT_RAIN, FX_OUT_SNOW,   /* this HRU's contribution before the Cv weighting (full_energy.c:429-431) */
  /* glacier */
  FX_GLAC_MASS_BALANCE, FX_GLAC_ICE_MASS_BALANCE, FX_GLAC_ACCUMULATION, FX_GLAC_MELT, FX_GLAC_VAPOR_FLUX,
  FX_GLAC_INFLOW, FX_GLAC_OUTFLOW, FX_GLAC_OUTFLOW_COEF, FX_GLAC_QNET, FX_GLAC_COLD_CONTENT,
  FX_GLACIER_FLUX, FX_DELTACC_GLAC, FX_GLACIER_MELT_ENERGY,
  /* read by put_data only: frost / thaw front depths (energy.fdepth/tdepth, m; NaN = no such front) and the per-layer
   * water table (layer[l].zwt, cm).  Like every row here they are rewritten by the step for the HRUs it computes; for
   * HRUs whose step does not produce them (glacier HRUs have no soil column step) they keep what vicgpu_set_fluxes gave */
  FX_FDEPTH0, FX_FDEPTH1, FX_FDEPTH2, FX_TDEPTH0, FX_TDEPTH1, FX_TDEPTH2, FX_ZWTL0, FX_ZWTL1, FX_ZWTL2,
  FX_NROW
};

/* per-step cell outputs: atmos->out_prec/out_rain/out_snow (full_energy.c:429-431), double[CO_NROW][ncell] */
enum { CO_OUT_PREC = 0, CO_OUT_RAIN, CO_OUT_SNOW, CO_NROW };

/* per-cell running sums over the steps of one vicgpu_step call (Cv-weighted like
 * put_data.c:789 "AreaFactor = Cv * mu * TreeAdjust", TreeAdjust = 1):
 * double[CA_NROW][ncell]; zeroed by vicgpu_reset_accum. */
enum {
  CA_RUNOFF = 0, CA_BASEFLOW, CA_EVAP, CA_SWE_END, CA_SOIL_MOIST_END0, CA_SOIL_MOIST_END1, CA_SOIL_MOIST_END2,
  CA_GLAC_MASS_BALANCE, CA_PREC, CA_NSTEPS, CA_NROW
};

/* per-cell error bits (the reference's ERROR returns, vicNl.c:545-559) */
#define VICGPU_CELLERR_SOLVER   1   /* a solver returned ERROR with TFALLBACK off */
#define VICGPU_CELLERR_AERO     2   /* CalcAerodynamic trunk-space error (CalcAerodynamic.c:214-217) */
#define VICGPU_CELLERR_NODES    4   /* thermal nodes do not reach below the bottom layer (soil_conduction.c:526-529) */
#define VICGPU_CELLERR_NAN      8   /* non-finite prognostic state after the step */

typedef struct vicgpu_ctx vicgpu_ctx;

/* ---- lifetime ------------------------------------------------------------ */
/* replaces ProgramState construction for the path (initialize_global.c:8, get_global_param.c:131) */
int  vicgpu_create(const vicgpu_options *opt, int device, vicgpu_ctx **out);
void vicgpu_destroy(vicgpu_ctx *ctx);
const char *vicgpu_last_error(const vicgpu_ctx *ctx);
int  vicgpu_abi_version(void);

/* ---- static domain --------------------------------------------------------
 * veg_lib: read_veglib.c:44-137 result.  Domain: readSoilData + read_vegparam +
 * read_snowband results (vicNl.c:237-293,170-175).  cell_hru_offset[ncell+1] /
 * cell_hru_list[nhru] is the CSR listing of each cell's HRUs in hruList order
 * (the order full_energy.c:216 iterates and sums out_prec in). */
int vicgpu_set_veglib(vicgpu_ctx *ctx, int nrow, const double *veglib);
int vicgpu_set_domain(vicgpu_ctx *ctx, int ncell, int nhru,
                      const double *cell_params,      /* [CP_NROW][ncell] */
                      const int    *hru_iparams,      /* [HPI_NROW][nhru] */
                      const double *hru_dparams,      /* [HPD_NROW][nhru] */
                      const int    *cell_hru_offset,  /* [ncell+1] */
                      const int    *cell_hru_list);   /* [nhru] */

/* ---- state: initialize_model_state result in / write_model_state content out -- */
int vicgpu_set_state(vicgpu_ctx *ctx, const double *state_d, const int *state_i);
int vicgpu_get_state(vicgpu_ctx *ctx, double *state_d, int *state_i);

/* The same state as the reference's state file holds it (write_model_state.c:95-337, processCellForStateFile): one record
 * per HRU, HRUs in cell-major hruList order (cell_hru_list order), every value as a double, fields in the order the
 * StateIO stream sees them (SR_*; node arrays Nnode long).  vicgpu_get_state_records gathers the records on the device
 * (the write side: the host streams them through its StateIO back-end unchanged); vicgpu_set_state_records is the read
 * side (read_initial_model_state.c): it scatters the persisted fields into the state and flux tables and leaves every
 * other row as it is -- like the reference, whose reader fills the HRUs initialize_model_state has just initialised.
 * Band and vegetation class of every record must match the domain (the reference throws, write_model_state.c:179-188):
 * VICGPU_ERR_ARG otherwise, with nothing scattered. */
enum {
  SR_BAND_INDEX = 0, SR_VEG_CLASS,
  SR_MOIST0, SR_MOIST1, SR_MOIST2, SR_ICE0, SR_ICE1, SR_ICE2,
  SR_WDEW,                       /* not in the stream of an artificial bare-soil HRU (write_model_state.c:240-242) */
  SR_SNOW_CANOPY, SR_SNOW_DENSITY, SR_SNOW_DEPTH, SR_SNOW_PACK_WATER, SR_SNOW_SURF_WATER, SR_SNOW_SWQ,
  SR_GLAC_WATER_STORAGE, SR_GLAC_CUM_MASS_BALANCE,
  SR_ENERGY_T                    /* Nnode values; the fields after it are addressed with VICGPU_SR() */
};
/* position of the fields that follow the first node array */
enum {
  SRT_TFOLIAGE = 0, SRT_GLAC_SURF_TEMP, SRT_SNOW_COLD_CONTENT, SRT_SNOW_PACK_TEMP, SRT_SNOW_SURF_TEMP, SRT_SNOW_ALBEDO,
  SRT_SNOW_LAST_SNOW, SRT_SNOW_MELTING, SRT_TCANOPY_FBCOUNT,
  SRT_T_FBCOUNT                  /* Nnode values */
};
enum {
  SRU_TFOLIAGE_FBCOUNT = 0, SRU_TSURF_FBCOUNT, SRU_GLAC_SURF_TEMP_FBCOUNT, SRU_SNOW_SURF_TEMP_FBCOUNT,
  SRU_GLAC_SURF_TEMP_FBFLAG, SRU_GLAC_VAPOR_FLUX, SRU_SNOW_CANOPY_ALBEDO, SRU_SNOW_SURFACE_FLUX, SRU_SNOW_SURF_TEMP_FBFLAG,
  SRU_SNOW_TMP_INT_STORAGE, SRU_SNOW_VAPOR_FLUX, SRU_NFIELD
};
#define VICGPU_SR_T(f, Nn)   (SR_ENERGY_T + (Nn) + (f))                  /* SRT_* field */
#define VICGPU_SR_U(f, Nn)   (SR_ENERGY_T + (Nn) + SRT_T_FBCOUNT + (Nn) + (f))   /* SRU_* field */
#define VICGPU_SR_LEN(Nn)    VICGPU_SR_U(SRU_NFIELD, Nn)
int vicgpu_get_state_records(vicgpu_ctx *ctx, double *records);         /* [nhru][VICGPU_SR_LEN(Nnode)] */
int vicgpu_set_state_records(vicgpu_ctx *ctx, const double *records);

/* ---- forcing chunk: replaces cell.atmos[rec] (initialize_atmos.c) for steps
 * [0, nsteps) of the chunk; copied to the device asynchronously on the copy stream. */
int vicgpu_push_forcing(vicgpu_ctx *ctx, int nsteps,
                        const double *forcing,          /* [nsteps][VIC_NFORCE][NF+1][ncell] */
                        const unsigned char *snowflag,  /* [nsteps][NF+1][ncell] */
                        const int *dmy);                /* [nsteps][VIC_NDMY] */

/* ---- forcing streaming.  The context holds two forcing chunks: the current one (vicgpu_step indexes it) and a spare.
 *   vicgpu_prefetch_forcing      starts the upload of the NEXT chunk into the spare (copy stream; returns at once)
 *   vicgpu_prefetch_forcing_raw  the same from hourly raw forcing (VIC_RAW_*): the device derives what initialize_atmos.c
 *                                derives per record -- kPa -> Pa (:290-295), the MIN_WIND_SPEED floor (:518-536), air
 *                                density from pressure (:980-1000, plapse != 0: the PLAPSE form), vpd = svp(T) - vp clipped
 *                                at 0 (:1175-1193), the snow_step aggregation with the step mean / sum in sub-index NR, and
 *                                the snowflag (:1275-1303)
 *   vicgpu_swap_forcing          makes the prefetched chunk the current one (steps already queued keep the old one)
 * so a driver overlaps the transfer of chunk k+1 with the steps of chunk k:
 *     push(0);  for k: { prefetch(k+1); step(0, n_k); swap(); }
 * vicgpu_push_forcing == prefetch + swap.  Host buffers: memory from vicgpu_host_alloc (pinned) is read by DMA while the
 * steps run and must stay unchanged until vicgpu_swap_forcing returns; any other memory is first copied into the
 * library's own pinned staging area, so it can be reused as soon as the call returns. */
int vicgpu_prefetch_forcing(vicgpu_ctx *ctx, int nsteps, const double *forcing, const unsigned char *snowflag, const int *dmy);
int vicgpu_prefetch_forcing_raw(vicgpu_ctx *ctx, int nsteps,
                                const double *raw,      /* [nsteps][VIC_NRAW][dt][ncell] */
                                const int *dmy, double min_wind_speed, int plapse);
int vicgpu_swap_forcing(vicgpu_ctx *ctx);
int vicgpu_get_forcing(vicgpu_ctx *ctx, int step, double *forcing /* [VIC_NFORCE][NF+1][ncell] */,
                       unsigned char *snowflag /* [NF+1][ncell] */);       /* read-back of the current chunk (tests) */
void *vicgpu_host_alloc(size_t bytes);                  /* pinned host memory for forcing chunks */
void vicgpu_host_free(void *p);

/* ---- the hot path: for rec in [step0, step0+nsteps): dist_prec for every cell
 * (vicNl.c:506-543).  With QUICK_FLUX the call only enqueues work.  With the
 * finite-difference soil profile (QUICK_FLUX off / FROZEN_SOIL) a step is a
 * data-dependent number of kernel rounds, so the call returns when the last
 * step's kernels have been issued, i.e. it blocks for most of the run time.
 * vicgpu_synchronize waits for completion in both cases.
 * Environment: VICGPU_CHUNKS=n runs n cell chunks as independent pipelines
 * (own streams and host threads; results are identical for any n),
 * VICGPU_TRACE / VICGPU_STATS print per-step round counts and timings. */
int vicgpu_step(vicgpu_ctx *ctx, int step0, int nsteps);
int vicgpu_synchronize(vicgpu_ctx *ctx);

/* ---- outputs of the last executed step / accumulated over calls ---------- */
int vicgpu_get_fluxes(vicgpu_ctx *ctx, double *flux);             /* [FX_NROW][nhru] */
int vicgpu_get_cell_outputs(vicgpu_ctx *ctx, double *cell_out);   /* [CO_NROW][ncell] */
int vicgpu_get_accum(vicgpu_ctx *ctx, double *accum);             /* [CA_NROW][ncell] */
int vicgpu_reset_accum(vicgpu_ctx *ctx);
int vicgpu_get_cell_errors(vicgpu_ctx *ctx, int *flags);          /* [ncell] */

/* ---- plumbing for callers that own device memory / streams --------------- */
int   vicgpu_set_stream(vicgpu_ctx *ctx, void *hip_stream);       /* compute stream; NULL = library-owned */
int   vicgpu_set_write_fluxes(vicgpu_ctx *ctx, int on);           /* 0: skip the per-HRU flux table (accumulators still kept) */
void *vicgpu_device_ptr(vicgpu_ctx *ctx, int which);              /* VICGPU_PTR_* */
enum { VICGPU_PTR_STATE_D = 0, VICGPU_PTR_STATE_I, VICGPU_PTR_FLUX, VICGPU_PTR_FORCING, VICGPU_PTR_ACCUM, VICGPU_PTR_CELL_OUT };
/* ---- glacier mass balance: what accumulateGlacierMassBalance.c:53-66 does when an
 * accumulation interval ends (the date test stays with the driver, like the opening of the
 * window): per cell the quadratic best fit of the accumulated mass balance of its glacier
 * HRUs against band elevation (GlacierMassBalanceResult.c:34-73, GraphingEquation.c:8-125;
 * 1 point: constant, 2 points: line), then -- if reset -- cum_mass_balance = 0 for every
 * glacier HRU (resetAccumulationValues).  eq: double[GMB_NROW][ncell]; cells without a
 * valid point get the default equation (0, 0, 0, fitError -1). */
enum { GMB_B0 = 0, GMB_B1, GMB_B2, GMB_FIT_ERROR, GMB_NROW };
int   vicgpu_glacier_mass_balance_fit(vicgpu_ctx *ctx, double *eq, int reset);

/* ---- test hook: the pure functions of the path, evaluated one by one ----------
 * (SURVEY.md 8(c) fixture plan (i)).  in: double[n][VICGPU_PURE_NIN], out: double[n].
 * Functions that read cell parameters or options use cell 0 of the domain and the
 * context's options.  The reference harness (oracle/ref_build) and the oracle
 * export the same hook as vicref_pure / vicorc_pure. */
enum {
  VICGPU_PURE_SVP = 0,            /* svp.c:7: T */
  VICGPU_PURE_SVP_SLOPE,          /* svp.c:26: T */
  VICGPU_PURE_CALC_RAINONLY,      /* calc_rainonly.c:12: air_temp, prec, MAX_SNOW_TEMP, MIN_RAIN_TEMP */
  VICGPU_PURE_SNOW_ALBEDO,        /* snow_utility.c:229: new_snow, swq, depth, albedo, cold_content, dt, last_snow, MELTING */
  VICGPU_PURE_NEW_SNOW_DENSITY,   /* snow_utility.c:199: air_temp */
  VICGPU_PURE_STABILITY,          /* StabilityCorrection.c:44: Z, d, TSurf, Tair, Wind, Z0 */
  VICGPU_PURE_PENMAN,             /* penman.c:96: tair, elevation, rad, vpd, ra, rc, rarc */
  VICGPU_PURE_CALC_RC,            /* penman.c:44: rs, net_short, RGL, tair, vpd, lai, gsm_inv, ref_crop */
  VICGPU_PURE_ESTIMATE_T1,        /* estimate_T1.c:8: Ts, T1_old, T2, D1, D2, kappa1, kappa2, Cs2 (Cs1 is unused: = Cs2), dp, delta_t */
  VICGPU_PURE_SOIL_CONDUCTIVITY,  /* soil_conduction.c:7: moist, Wu, soil_dens_min, bulk_dens_min, quartz, soil_density, bulk_density, organic */
  VICGPU_PURE_VOL_HEAT_CAPACITY,  /* soil_conduction.c:108: soil_fract, water_fract, ice_fract, organic_fract */
  VICGPU_PURE_MAX_UNFROZEN_WATER, /* soil_conduction.c: T, max_moist, bubble, expt */
  VICGPU_PURE_LINEAR_INTERP,      /* x, lx, ux, ly, uy */
  VICGPU_PURE_VEG_HEIGHT,         /* calc_veg_params.c:26: displacement, L (NaN for L = 0) */
  VICGPU_PURE_NFN,
  /* device-only hook (no reference counterpart, not part of the golden vectors): soil_conductivity with the layer
   * constants the library derives once per domain (cell 0): moist, Wu, layer.  Must equal VICGPU_PURE_SOIL_CONDUCTIVITY
   * fed with that layer's parameters, bit for bit. */
  VICGPU_PURE_SOIL_CONDUCTIVITY_DERIVED = VICGPU_PURE_NFN,
  VICGPU_PURE_NFN_DEVICE
};
#define VICGPU_PURE_NIN 10
int   vicgpu_debug_pure(vicgpu_ctx *ctx, int fn, int n, const double *in, double *out);

/* GPU time (ms) per model step of the last vicgpu_step call, measured with
 * hipEvents on the library's streams: QUICK_FLUX: the step's HRU kernel, one
 * event pair per step; finite-difference pipeline: all kernels of all steps
 * of the call divided by the number of steps.  *nlaunch = steps covered. */
int   vicgpu_last_kernel_ms(vicgpu_ctx *ctx, double *ms_per_launch, int *nlaunch);

#ifdef __cplusplus
}
#endif
#endif /* VICGPU_H_ */
