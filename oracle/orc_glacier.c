/*
 * orc_glacier.c — TEST INFRASTRUCTURE (CPU oracle): glacier HRU path (surface_fluxes_glac and below).
 */
#include "orc.h"

int orc_surface_fluxes_glac(const orc_model *m, orc_hru *h, const orc_soil *sc, orc_atmos *atmos, const orc_dmy *dmy,
                            double BareAlbedo, double ice0, double moist0, orc_vc *aero_resist, orc_vc *displacement,
                            orc_vc *ref_height, orc_vc *roughness, orc_vc *wind_speed, double *out_prec, double *out_rain,
                            double *out_snow) {
  (void)m; (void)h; (void)sc; (void)atmos; (void)dmy; (void)BareAlbedo; (void)ice0; (void)moist0; (void)aero_resist;
  (void)displacement; (void)ref_height; (void)roughness; (void)wind_speed; (void)out_prec; (void)out_rain; (void)out_snow;
  return -1;
}
