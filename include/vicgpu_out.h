/*
 * vicgpu_out.h — put_data on the device: the aggregated per-cell output variables.
 *
 * What this replaces.  After full_energy, dist_prec.c:167 calls put_data (put_data.c:7-760) for the cell: it sums
 * the HRUs' storages and fluxes, weighted by area, into the cell's OutputData list (collect_wb_terms
 * put_data.c:762-948, collect_eb_terms :950-1232), derives totals and the water / energy balance errors
 * (:560-633, calc_water_energy_balance_errors.c:7-94) and aggregates in time (:663-685).  The writer
 * (WriteOutputNetCDF.c:387-455) then reads OutputData.aggdata of the selected variables as floats.
 *
 *   vicgpu_put_data_config    once: output interval (out_dt / dt) -> enables the per-step aggregation kernel
 *   vicgpu_put_data_init      the reference's put_data(rec = -nrecs) call before the first step (vicNl.c:524-541):
 *                             storages and balance-error accumulators are initialised from the current state
 *   vicgpu_step               ... then aggregates every step it executes (no extra call)
 *   vicgpu_get_outputs        aggdata of the listed variables, float[sum nelem][ncell], in the order asked for;
 *                             reset = what vicNl.c:599-606 does after write_data_all_cells
 *   vicgpu_get_output_data    the same variables as doubles: which = 0 the un-aggregated values of the last step
 *                             (OutputData.data), which = 1 the aggregates (OutputData.aggdata)
 *
 * Variables are identified by the reference's names ("OUT_RUNOFF"); vicgpu_out_var_id maps a name to this
 * library's index, so a binding translates its own enum (vicNl_def.h:351-564) once by name
 * (OutputData.varname).  Lake variables (LAKES is not supported), the EXCESS_ICE variables and OUT_TSKC (cloud
 * fraction is not among the forcing variables of the path) have no index here: vicgpu_out_var_id returns -1.
 * Element counts: 1, Nlayer, Nnode, SNOW_BAND or MAX_FRONTS (output_list_utils.c:296-351); aggregation END / SUM /
 * AVG (output_list_utils.c:355-480).
 */
#ifndef VICGPU_OUT_H_
#define VICGPU_OUT_H_

#include "vicgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VIC_MAX_FRONTS 3   /* user_def.h MAX_FRONTS */

enum { VOUT_K1 = 0, VOUT_KLAYER, VOUT_KNODE, VOUT_KBAND, VOUT_KFRONT };   /* number of elements of a variable */
enum { VOUT_AGG_END = 0, VOUT_AGG_SUM, VOUT_AGG_AVG };                    /* AGG_TYPE_* (vicNl_def.h) */

/* X(name without the OUT_ prefix, element kind, aggregation) */
#define VICGPU_OUT_VARS(X)                                                                                         \
  /* forcing echoes (put_data.c:229-256) */                                                                        \
  X(AIR_TEMP, VOUT_K1, VOUT_AGG_AVG) X(DENSITY, VOUT_K1, VOUT_AGG_AVG) X(LONGWAVE, VOUT_K1, VOUT_AGG_AVG)          \
  X(PREC, VOUT_K1, VOUT_AGG_SUM) X(PRESSURE, VOUT_K1, VOUT_AGG_AVG) X(QAIR, VOUT_K1, VOUT_AGG_AVG)                 \
  X(RAINF, VOUT_K1, VOUT_AGG_SUM) X(REL_HUMID, VOUT_K1, VOUT_AGG_AVG) X(SHORTWAVE, VOUT_K1, VOUT_AGG_AVG)          \
  X(SNOWF, VOUT_K1, VOUT_AGG_SUM) X(VP, VOUT_K1, VOUT_AGG_AVG) X(VPD, VOUT_K1, VOUT_AGG_AVG)                       \
  X(WIND, VOUT_K1, VOUT_AGG_AVG)                                                                                   \
  /* water balance: storages */                                                                                    \
  X(ASAT, VOUT_K1, VOUT_AGG_END) X(ROOTMOIST, VOUT_K1, VOUT_AGG_END) X(SMFROZFRAC, VOUT_KLAYER, VOUT_AGG_END)      \
  X(SMLIQFRAC, VOUT_KLAYER, VOUT_AGG_END) X(SNOW_CANOPY, VOUT_K1, VOUT_AGG_END) X(SNOW_COVER, VOUT_K1, VOUT_AGG_END) \
  X(SNOW_DEPTH, VOUT_K1, VOUT_AGG_END) X(SOIL_ICE, VOUT_KLAYER, VOUT_AGG_END) X(SOIL_ICE_TOT, VOUT_K1, VOUT_AGG_END) \
  X(SOIL_LIQ, VOUT_KLAYER, VOUT_AGG_END) X(SOIL_LIQ_TOT, VOUT_K1, VOUT_AGG_END) X(SOIL_MOIST, VOUT_KLAYER, VOUT_AGG_END) \
  X(SOIL_MOIST_TOT, VOUT_K1, VOUT_AGG_END) X(SOIL_WET, VOUT_K1, VOUT_AGG_END) X(SURFSTOR, VOUT_K1, VOUT_AGG_END)   \
  X(SURF_FROST_FRAC, VOUT_K1, VOUT_AGG_END) X(SWE, VOUT_K1, VOUT_AGG_END) X(WDEW, VOUT_K1, VOUT_AGG_END)           \
  X(ZWT, VOUT_K1, VOUT_AGG_END) X(ZWT2, VOUT_K1, VOUT_AGG_END) X(ZWT3, VOUT_K1, VOUT_AGG_END)                      \
  X(ZWTL, VOUT_KLAYER, VOUT_AGG_END)                                                                               \
  /* water balance: fluxes */                                                                                      \
  X(BASEFLOW, VOUT_K1, VOUT_AGG_SUM) X(DELINTERCEPT, VOUT_K1, VOUT_AGG_SUM) X(DELSOILMOIST, VOUT_K1, VOUT_AGG_SUM) \
  X(DELSURFSTOR, VOUT_K1, VOUT_AGG_SUM) X(DELSWE, VOUT_K1, VOUT_AGG_SUM) X(EVAP, VOUT_K1, VOUT_AGG_SUM)            \
  X(EVAP_BARE, VOUT_K1, VOUT_AGG_SUM) X(EVAP_CANOP, VOUT_K1, VOUT_AGG_SUM) X(INFLOW, VOUT_K1, VOUT_AGG_SUM)        \
  X(PET_SATSOIL, VOUT_K1, VOUT_AGG_SUM) X(PET_H2OSURF, VOUT_K1, VOUT_AGG_SUM) X(PET_SHORT, VOUT_K1, VOUT_AGG_SUM)  \
  X(PET_TALL, VOUT_K1, VOUT_AGG_SUM) X(PET_NATVEG, VOUT_K1, VOUT_AGG_SUM) X(PET_VEGNOCR, VOUT_K1, VOUT_AGG_SUM)    \
  X(REFREEZE, VOUT_K1, VOUT_AGG_SUM) X(RUNOFF, VOUT_K1, VOUT_AGG_SUM) X(SNOW_MELT, VOUT_K1, VOUT_AGG_SUM)          \
  X(SUB_BLOWING, VOUT_K1, VOUT_AGG_SUM) X(SUB_CANOP, VOUT_K1, VOUT_AGG_SUM) X(SUB_SNOW, VOUT_K1, VOUT_AGG_SUM)     \
  X(SUB_SURFACE, VOUT_K1, VOUT_AGG_SUM) X(TRANSP_VEG, VOUT_K1, VOUT_AGG_SUM) X(WATER_ERROR, VOUT_K1, VOUT_AGG_AVG) \
  /* energy balance: states */                                                                                     \
  X(ALBEDO, VOUT_K1, VOUT_AGG_AVG) X(BARESOILT, VOUT_K1, VOUT_AGG_AVG) X(FDEPTH, VOUT_KFRONT, VOUT_AGG_AVG)        \
  X(RAD_TEMP, VOUT_K1, VOUT_AGG_AVG) X(SALBEDO, VOUT_K1, VOUT_AGG_AVG) X(SNOW_PACK_TEMP, VOUT_K1, VOUT_AGG_AVG)    \
  X(SNOW_SURF_TEMP, VOUT_K1, VOUT_AGG_AVG) X(SNOWT_FBFLAG, VOUT_K1, VOUT_AGG_SUM) X(SOIL_TEMP, VOUT_KLAYER, VOUT_AGG_AVG) \
  X(SOIL_TNODE, VOUT_KNODE, VOUT_AGG_AVG) X(SOILT_FBFLAG, VOUT_KNODE, VOUT_AGG_SUM) X(SURF_TEMP, VOUT_K1, VOUT_AGG_AVG) \
  X(SURFT_FBFLAG, VOUT_K1, VOUT_AGG_SUM) X(TCAN_FBFLAG, VOUT_K1, VOUT_AGG_SUM) X(TDEPTH, VOUT_KFRONT, VOUT_AGG_AVG) \
  X(TFOL_FBFLAG, VOUT_K1, VOUT_AGG_SUM) X(VEGT, VOUT_K1, VOUT_AGG_AVG)                                             \
  /* energy balance: fluxes */                                                                                     \
  X(ADV_SENS, VOUT_K1, VOUT_AGG_AVG) X(ADVECTION, VOUT_K1, VOUT_AGG_AVG) X(DELTACC, VOUT_K1, VOUT_AGG_AVG)         \
  X(DELTAH, VOUT_K1, VOUT_AGG_AVG) X(ENERGY_ERROR, VOUT_K1, VOUT_AGG_AVG) X(FUSION, VOUT_K1, VOUT_AGG_AVG)         \
  X(GRND_FLUX, VOUT_K1, VOUT_AGG_AVG) X(IN_LONG, VOUT_K1, VOUT_AGG_AVG) X(LATENT, VOUT_K1, VOUT_AGG_AVG)           \
  X(LATENT_SUB, VOUT_K1, VOUT_AGG_AVG) X(MELT_ENERGY, VOUT_K1, VOUT_AGG_AVG) X(NET_LONG, VOUT_K1, VOUT_AGG_AVG)    \
  X(NET_SHORT, VOUT_K1, VOUT_AGG_AVG) X(R_NET, VOUT_K1, VOUT_AGG_AVG) X(RFRZ_ENERGY, VOUT_K1, VOUT_AGG_AVG)        \
  X(SENSIBLE, VOUT_K1, VOUT_AGG_AVG) X(SNOW_FLUX, VOUT_K1, VOUT_AGG_AVG)                                           \
  /* aerodynamics */                                                                                               \
  X(AERO_COND, VOUT_K1, VOUT_AGG_AVG) X(AERO_COND1, VOUT_K1, VOUT_AGG_AVG) X(AERO_COND2, VOUT_K1, VOUT_AGG_AVG)    \
  X(AERO_RESIST, VOUT_K1, VOUT_AGG_AVG) X(AERO_RESIST1, VOUT_K1, VOUT_AGG_AVG) X(AERO_RESIST2, VOUT_K1, VOUT_AGG_AVG) \
  X(SURF_COND, VOUT_K1, VOUT_AGG_AVG) /* never assigned by put_data: always 0 */                                   \
  /* snow / elevation bands */                                                                                     \
  X(ADV_SENS_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(ADVECTION_BAND, VOUT_KBAND, VOUT_AGG_AVG)                           \
  X(ALBEDO_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(AREA_BAND, VOUT_KBAND, VOUT_AGG_END)                                  \
  X(DELTACC_BAND, VOUT_KBAND, VOUT_AGG_SUM) X(ELEV_BAND, VOUT_KBAND, VOUT_AGG_END)                                 \
  X(GRND_FLUX_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(IN_LONG_BAND, VOUT_KBAND, VOUT_AGG_AVG)                            \
  X(LATENT_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(LATENT_SUB_BAND, VOUT_KBAND, VOUT_AGG_AVG)                            \
  X(MELT_ENERGY_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(NET_LONG_BAND, VOUT_KBAND, VOUT_AGG_AVG)                         \
  X(NET_SHORT_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(RFRZ_ENERGY_BAND, VOUT_KBAND, VOUT_AGG_AVG)                        \
  X(SENSIBLE_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(SNOW_CANOPY_BAND, VOUT_KBAND, VOUT_AGG_END)                         \
  X(SNOW_COVER_BAND, VOUT_KBAND, VOUT_AGG_END) X(SNOW_DEPTH_BAND, VOUT_KBAND, VOUT_AGG_END)                        \
  X(SNOW_FLUX_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(SNOW_MELT_BAND, VOUT_KBAND, VOUT_AGG_AVG)                          \
  X(SNOW_PACKT_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(SNOW_SURFT_BAND, VOUT_KBAND, VOUT_AGG_AVG)                        \
  X(SWE_BAND, VOUT_KBAND, VOUT_AGG_END)                                                                            \
  /* glacier */                                                                                                    \
  X(GLAC_WAT_STOR, VOUT_K1, VOUT_AGG_END) X(GLAC_AREA, VOUT_K1, VOUT_AGG_END) X(GLAC_MBAL, VOUT_K1, VOUT_AGG_SUM)  \
  X(GLAC_IMBAL, VOUT_K1, VOUT_AGG_SUM) X(GLAC_ACCUM, VOUT_K1, VOUT_AGG_SUM) X(GLAC_MELT, VOUT_K1, VOUT_AGG_SUM)    \
  X(GLAC_SUB, VOUT_K1, VOUT_AGG_SUM) X(GLAC_INFLOW, VOUT_K1, VOUT_AGG_SUM) X(GLAC_OUTFLOW, VOUT_K1, VOUT_AGG_SUM)  \
  X(GLAC_SURF_TEMP, VOUT_K1, VOUT_AGG_END) X(GLAC_TSURF_FBFLAG, VOUT_K1, VOUT_AGG_END)                             \
  X(GLAC_DELTACC, VOUT_K1, VOUT_AGG_AVG) X(GLAC_FLUX, VOUT_K1, VOUT_AGG_AVG) X(GLAC_OUTFLOW_COEF, VOUT_K1, VOUT_AGG_END) \
  X(GLAC_MELT_ENERGY, VOUT_K1, VOUT_AGG_AVG)                                                                       \
  X(GLAC_DELTACC_BAND, VOUT_KBAND, VOUT_AGG_AVG) X(GLAC_FLUX_BAND, VOUT_KBAND, VOUT_AGG_AVG)                       \
  X(GLAC_WAT_STOR_BAND, VOUT_KBAND, VOUT_AGG_END) X(GLAC_AREA_BAND, VOUT_KBAND, VOUT_AGG_END)                      \
  X(GLAC_MBAL_BAND, VOUT_KBAND, VOUT_AGG_SUM) X(GLAC_IMBAL_BAND, VOUT_KBAND, VOUT_AGG_SUM)                         \
  X(GLAC_ACCUM_BAND, VOUT_KBAND, VOUT_AGG_SUM) X(GLAC_MELT_BAND, VOUT_KBAND, VOUT_AGG_SUM)                         \
  X(GLAC_SUB_BAND, VOUT_KBAND, VOUT_AGG_SUM) X(GLAC_INFLOW_BAND, VOUT_KBAND, VOUT_AGG_SUM)                         \
  X(GLAC_OUTFLOW_BAND, VOUT_KBAND, VOUT_AGG_SUM)

#define VICGPU_OUT_ENUM_(name, kind, agg) VOUT_##name,
enum { VICGPU_OUT_VARS(VICGPU_OUT_ENUM_) VOUT_NVAR };
#undef VICGPU_OUT_ENUM_

/* per-cell bookkeeping put_data carries from step to step (vicNl_def.h:1405-1410, 1451-1478, 1524-1539):
 * double[PB_NROW][ncell] */
enum {
  PB_SAVE_TOTAL_SOIL_MOIST = 0, PB_SAVE_SWE, PB_SAVE_WDEW, PB_SAVE_SURFSTOR,      /* save_data */
  PB_WATER_LAST_STORAGE, PB_WATER_CUM_ERROR, PB_WATER_MAX_ERROR,                  /* cellErrors */
  PB_ENERGY_CUM_ERROR, PB_ENERGY_MAX_ERROR,
  PB_FB_TFOLIAGE, PB_FB_TCANOPY, PB_FB_TSNOWSURF, PB_FB_TSURF, PB_FB_TSOIL, PB_FB_TGLACSURF,   /* fallBackStats totals */
  PB_NROW
};

int vicgpu_out_nvar(void);
int vicgpu_out_var_id(const char *name);                 /* "OUT_RUNOFF" -> VOUT_RUNOFF, -1 when not provided */
const char *vicgpu_out_var_name(int id);                 /* "OUT_RUNOFF" */
int vicgpu_out_var_kind(int id);                         /* VOUT_K* */
int vicgpu_out_var_agg(int id);                          /* VOUT_AGG_* */
int vicgpu_out_var_nelem(const vicgpu_options *opt, int id);

int vicgpu_put_data_config(vicgpu_ctx *ctx, int out_step_ratio);
int vicgpu_put_data_init(vicgpu_ctx *ctx);
int vicgpu_get_outputs(vicgpu_ctx *ctx, int nvar, const int *var_ids, float *out, int reset);
int vicgpu_get_output_data(vicgpu_ctx *ctx, int nvar, const int *var_ids, int which, double *out);
int vicgpu_get_balance(vicgpu_ctx *ctx, double *pb);     /* [PB_NROW][ncell] */
int vicgpu_set_fluxes(vicgpu_ctx *ctx, const double *flux);   /* [FX_NROW][nhru]: the per-HRU values put_data reads that the
                                                                 reference carries from initialize_model_state (see FX_*) */

#ifdef __cplusplus
}
#endif
#endif /* VICGPU_OUT_H_ */
