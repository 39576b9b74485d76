/*
 * vicref_shim.cpp — TEST INFRASTRUCTURE (never linked into the product).
 *
 * A thin extern "C" harness around the *real* reference implementation
 * (/root/reference, compiled where it lies by oracle/ref_build/build_ref.sh
 * into oracle/_ref/libvicref*.so).  It converts the flat tables of
 * include/vicgpu.h into the reference's own structs (cell_info_struct,
 * soil_con_struct, HRU, atmos_data_struct: vicNl_def.h:900-1539), calls the
 * reference's initialize_model_state (initialize_model_state.c:8) and
 * full_energy (full_energy.c:8) exactly as vicNl.c:420-427 / dist_prec.c:159 do,
 * and converts the result back.  No reference source text lives here: only calls
 * into it and field-by-field copies.
 *
 * Used by tests/ (to pin oracle/vic_oracle.c against the reference and to generate
 * tests/golden/ fixtures) and by bench.py's cpu_baseline leg (kind "reference").
 */
#include "vicNl.h"
#include "global.h"
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "vicgpu.h"
#include "vicgpu_out.h"
#include "vicgpu_binding.h"      /* integration/: the reference-side binding, compiled and exercised here */
#include "GlacierMassBalanceResult.h"

struct vicref_handle {
  ProgramState state;
  vicgpu_options opt;
  int ncell, nhru, NR, NF;
  std::vector<cell_info_struct> cells;
  std::vector<int> hru_cell;          /* global hru id -> cell */
  std::vector<int> hru_pos;           /* global hru id -> position in the cell's hruList */
  std::vector<int> cell_off, cell_list;
  std::vector<double> cp;             /* copy of the cell parameter table */
  int cp_nrow;
  std::vector<OutputData*> out;       /* the reference's own output list (create_output_list) per cell, for put_data */
  OutputData *out_template = NULL;
};

static inline double cpv(const vicref_handle *h, int row, int c) { return h->cp[(size_t)row * h->ncell + c]; }

/* accumulateGlacierMassBalance.c:5 (C++ linkage, not declared in a header) */
void resetAccumulationValues(std::vector<HRU>* hruList);

/* ---- the reference's own state-file stream (processCellForStateFile, write_model_state.c:105-337) into / out of memory:
 * a StateIO back-end that records every value the function hands it, with the variable id, instead of formatting it.
 * Pins the record order of vicgpu_get_state_records / vicorc_state_records. */
class VicrefBufferIO : public StateIO {
public:
  std::vector<double> vals;
  std::vector<int> ids;
  size_t at = 0;
  VicrefBufferIO(IOType t, const ProgramState *st) : StateIO("(memory)", t, st) {}
  void initializeOutput() {}
  template <typename T> int put(const T *d, int n, int id) { for (int i = 0; i < n; i++) { vals.push_back((double)d[i]); ids.push_back(id); } return n; }
  template <typename T> int get(T *d, int n, int id) {
    for (int i = 0; i < n; i++) {
      if (at >= vals.size() || ids[at] != id) throw VICException("VicrefBufferIO: stream out of step");
      d[i] = (T)vals[at++];
    }
    return n;
  }
  int write(const int *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return put(d, n, id); }
  int write(const double *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return put(d, n, id); }
  int write(const float *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return put(d, n, id); }
  int write(const char *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return put(d, n, id); }
  int write(const bool *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return put(d, n, id); }
  int read(int *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return get(d, n, id); }
  int read(double *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return get(d, n, id); }
  int read(float *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return get(d, n, id); }
  int read(bool *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return get(d, n, id); }
  int read(char *d, int n, const StateVariables::StateMetaDataVariableIndices id) { return get(d, n, id); }
  StateHeader readHeader() { return StateHeader(0, 0, 0, 0, 0); }
  int seekToCell(int, int *, int *) { return 0; }
  void flush() {}
  void rewindFile() {}
};


/* ---- initialize_atmos (initialize_atmos.c:7-1349) driven from memory.  read_forcing_data.c calls read_atmos_data for each
 * forcing file; the reference's read_atmos_data.c parses a NetCDF / ASCII / binary file and is not part of this build (it
 * needs the netCDF C library).  The harness supplies the file-reading end itself: the "file" is a table the test filled,
 * [type][record at the forcing time step], copied into forcing_data[] exactly as read_atmos_data leaves the values of a
 * file (file units: kPa for PRESSURE and VP; initialize_atmos converts).  Everything downstream -- local-time arrays, MTCLIM,
 * sub-step aggregation, vpd, density, snowflag -- is the reference's own code. */
struct VicrefForcingSource { const double *data[N_FORCING_TYPES]; int nrec; };
static thread_local const VicrefForcingSource *vicref_forcing_source = NULL;

void read_atmos_data(FILE *, int, int file_num, int, double **forcing_data, soil_con_struct *, const ProgramState *state) {
  const VicrefForcingSource *src = vicref_forcing_source;
  if (!src || file_num != 0) throw VICException("vicref: read_atmos_data without an in-memory forcing source");
  for (int t = 0; t < N_FORCING_TYPES; t++)
    if (state->param_set.TYPE[t].SUPPLIED && src->data[t])
      for (int k = 0; k < src->nrec; k++) forcing_data[t][k] = src->data[t][k];
}

extern "C" {

void *vicref_create(const vicgpu_options *opt) {
  if (!opt || opt->abi_version != VICGPU_ABI_VERSION || opt->Nlayer != 3) return NULL;
  vicref_handle *h = new vicref_handle();
  h->opt = *opt;
  ProgramState &s = h->state;
  s.initialize_global();
  s.options.Nlayer = opt->Nlayer;
  s.options.Nnode = opt->Nnode;
  s.options.SNOW_BAND = opt->Nband;
  s.options.SNOW_STEP = opt->snow_step;
  s.options.FULL_ENERGY = opt->FULL_ENERGY;
  s.options.FROZEN_SOIL = opt->FROZEN_SOIL;
  s.options.QUICK_FLUX = opt->QUICK_FLUX;
  s.options.CORRPREC = opt->CORRPREC;
  s.options.BLOWING = opt->BLOWING;
  s.options.NOFLUX = opt->NOFLUX;
  s.options.IMPLICIT = opt->IMPLICIT;
  s.options.QUICK_SOLVE = opt->QUICK_SOLVE;
  s.options.EXP_TRANS = opt->EXP_TRANS;
  s.options.GRND_FLUX_TYPE = opt->GRND_FLUX_TYPE;
  s.options.TFALLBACK = opt->TFALLBACK;
  s.options.AERO_RESIST_CANSNOW = opt->AERO_RESIST_CANSNOW;
  s.options.SNOW_ALBEDO = opt->SNOW_ALBEDO;
  s.options.SNOW_DENSITY = opt->SNOW_DENSITY;
  s.options.TEMP_TH_TYPE = opt->TEMP_TH_TYPE;
  s.options.GLACIER_ID = opt->GLACIER_ID;
  s.options.GLACIER_DYNAMICS = opt->GLACIER_DYNAMICS != 0;
  s.options.CONTINUEONERROR = TRUE;
  s.global_param.dt = opt->dt;
  s.global_param.out_dt = opt->dt;
  s.global_param.wind_h = opt->wind_h;
  s.global_param.measure_h = 2.0;
  s.global_param.resolution = 0.0625f;
  s.global_param.nrecs = 24 / opt->dt * 366 * 100;
  s.global_param.startyear = 2000; s.global_param.startmonth = 1; s.global_param.startday = 1; s.global_param.starthour = 0;
  s.global_param.glacierAccumStartYear = 0; s.global_param.glacierAccumStartMonth = 0; s.global_param.glacierAccumStartDay = 0;
  s.global_param.glacierAccumInterval = 0;
  s.glacier_accum_started = false;
  h->NF = VICGPU_NF(opt);
  h->NR = VICGPU_NR(opt);
  s.NF = h->NF; s.NR = h->NR;
  s.dt_sec = opt->dt * 3600; s.out_dt_sec = s.dt_sec; s.out_step_ratio = 1;
  s.num_veg_types = opt->nveg_types;
  h->ncell = h->nhru = 0;
  return h;
}

void vicref_destroy(void *hv) {
  vicref_handle *h = (vicref_handle *)hv;
  if (!h) return;
  delete h;   /* (small leaks of calloc'ed band arrays / atmos are accepted in this test harness) */
}

int vicref_set_veglib(void *hv, int nrow, const double *t) {
  vicref_handle *h = (vicref_handle *)hv;
  if (nrow != h->opt.nveg_types + 4) return -1;
  veg_lib_struct *vl = (veg_lib_struct *)calloc(nrow, sizeof(veg_lib_struct));
  for (int i = 0; i < nrow; i++) {
    const double *r = t + (size_t)i * VL_NFIELD;
    vl[i].overstory = (char)(r[VL_OVERSTORY] != 0);
    vl[i].rarc = r[VL_RARC]; vl[i].rmin = r[VL_RMIN]; vl[i].rad_atten = r[VL_RAD_ATTEN];
    vl[i].trunk_ratio = r[VL_TRUNK_RATIO]; vl[i].wind_atten = r[VL_WIND_ATTEN]; vl[i].wind_h = r[VL_WIND_H];
    vl[i].RGL = (float)r[VL_RGL]; vl[i].veg_class = (int)r[VL_VEG_CLASS];
    vl[i].NVegLibTypes = h->opt.nveg_types;
    for (int m = 0; m < 12; m++) {
      vl[i].LAI[m] = r[VL_LAI + m]; vl[i].Wdmax[m] = r[VL_WDMAX + m]; vl[i].albedo[m] = r[VL_ALBEDO + m];
      vl[i].displacement[m] = r[VL_DISPLACEMENT + m]; vl[i].emissivity[m] = r[VL_EMISSIVITY + m];
      vl[i].roughness[m] = r[VL_ROUGHNESS + m];
    }
  }
  h->state.veg_lib = vl;
  return 0;
}

static void fill_soil_con(vicref_handle *h, int c, soil_con_struct *sc) {
  const int Nn = h->opt.Nnode, Nb = h->opt.Nband;
  memset(sc, 0, sizeof(*sc));
  sc->Ds = cpv(h, CP_DS, c); sc->Dsmax = cpv(h, CP_DSMAX, c); sc->Ws = cpv(h, CP_WS, c); sc->c = cpv(h, CP_C, c);
  sc->b_infilt = cpv(h, CP_B_INFILT, c); sc->dp = cpv(h, CP_DP, c); sc->avg_temp = cpv(h, CP_AVG_TEMP, c);
  sc->rough = cpv(h, CP_ROUGH, c); sc->snow_rough = cpv(h, CP_SNOW_ROUGH, c);
  sc->elevation = (float)cpv(h, CP_ELEVATION, c); sc->lat = (float)cpv(h, CP_LAT, c);
  sc->FS_ACTIVE = (int)cpv(h, CP_FS_ACTIVE, c);
  sc->NEW_SNOW_ALB = cpv(h, CP_NEW_SNOW_ALB, c);
  sc->SNOW_ALB_ACCUM_A = cpv(h, CP_SNOW_ALB_ACCUM_A, c); sc->SNOW_ALB_ACCUM_B = cpv(h, CP_SNOW_ALB_ACCUM_B, c);
  sc->SNOW_ALB_THAW_A = cpv(h, CP_SNOW_ALB_THAW_A, c); sc->SNOW_ALB_THAW_B = cpv(h, CP_SNOW_ALB_THAW_B, c);
  sc->MIN_RAIN_TEMP = cpv(h, CP_MIN_RAIN_TEMP, c); sc->MAX_SNOW_TEMP = cpv(h, CP_MAX_SNOW_TEMP, c);
  sc->PADJ_R = cpv(h, CP_PADJ_R, c); sc->PADJ_S = cpv(h, CP_PADJ_S, c);
  sc->GLAC_SURF_THICK = cpv(h, CP_GLAC_SURF_THICK, c); sc->GLAC_SURF_WE = cpv(h, CP_GLAC_SURF_WE, c);
  sc->GLAC_KMIN = cpv(h, CP_GLAC_KMIN, c); sc->GLAC_DK = cpv(h, CP_GLAC_DK, c); sc->GLAC_A = cpv(h, CP_GLAC_A, c);
  sc->GLAC_ALBEDO = cpv(h, CP_GLAC_ALBEDO, c); sc->GLAC_ROUGH = cpv(h, CP_GLAC_ROUGH, c);
  for (int l = 0; l < 3; l++) {
    sc->Ksat[l] = cpv(h, VICGPU_CP_LAYER(CPL_KSAT, l), c);
    sc->Wcr[l] = cpv(h, VICGPU_CP_LAYER(CPL_WCR, l), c);
    sc->Wpwp[l] = cpv(h, VICGPU_CP_LAYER(CPL_WPWP, l), c);
    sc->expt[l] = cpv(h, VICGPU_CP_LAYER(CPL_EXPT, l), c);
    sc->bubble[l] = cpv(h, VICGPU_CP_LAYER(CPL_BUBBLE, l), c);
    sc->depth[l] = cpv(h, VICGPU_CP_LAYER(CPL_DEPTH, l), c);
    sc->max_moist[l] = cpv(h, VICGPU_CP_LAYER(CPL_MAX_MOIST, l), c);
    sc->resid_moist[l] = cpv(h, VICGPU_CP_LAYER(CPL_RESID_MOIST, l), c);
    sc->porosity[l] = cpv(h, VICGPU_CP_LAYER(CPL_POROSITY, l), c);
    sc->quartz[l] = cpv(h, VICGPU_CP_LAYER(CPL_QUARTZ, l), c);
    sc->organic[l] = cpv(h, VICGPU_CP_LAYER(CPL_ORGANIC, l), c);
    sc->bulk_density[l] = cpv(h, VICGPU_CP_LAYER(CPL_BULK_DENSITY, l), c);
    sc->soil_density[l] = cpv(h, VICGPU_CP_LAYER(CPL_SOIL_DENSITY, l), c);
    sc->bulk_dens_min[l] = cpv(h, VICGPU_CP_LAYER(CPL_BULK_DENS_MIN, l), c);
    sc->soil_dens_min[l] = cpv(h, VICGPU_CP_LAYER(CPL_SOIL_DENS_MIN, l), c);
    sc->init_moist[l] = 0;  /* set by vicref_init_state */
  }
  for (int n = 0; n < Nn; n++) {
    sc->Zsum_node[n] = cpv(h, VICGPU_CP_NODE(CPN_ZSUM, n, Nn), c);
    sc->dz_node[n] = cpv(h, VICGPU_CP_NODE(CPN_DZ, n, Nn), c);
    sc->alpha[n] = cpv(h, VICGPU_CP_NODE(CPN_ALPHA, n, Nn), c);
    sc->beta[n] = cpv(h, VICGPU_CP_NODE(CPN_BETA, n, Nn), c);
    sc->gamma[n] = cpv(h, VICGPU_CP_NODE(CPN_GAMMA, n, Nn), c);
    sc->max_moist_node[n] = cpv(h, VICGPU_CP_NODE(CPN_MAX_MOIST, n, Nn), c);
    sc->expt_node[n] = cpv(h, VICGPU_CP_NODE(CPN_EXPT, n, Nn), c);
    sc->bubble_node[n] = cpv(h, VICGPU_CP_NODE(CPN_BUBBLE, n, Nn), c);
  }
  sc->BandElev = (float *)calloc(Nb, sizeof(float));
  sc->AreaFract = (double *)calloc(Nb, sizeof(double));
  sc->Pfactor = (double *)calloc(Nb, sizeof(double));
  sc->Tfactor = (double *)calloc(Nb, sizeof(double));
  sc->AboveTreeLine = (char *)calloc(Nb, sizeof(char));
  for (int b = 0; b < Nb; b++) {
    sc->AreaFract[b] = cpv(h, VICGPU_CP_BAND(CPB_AREAFRACT, b, Nn, Nb), c);
    sc->Tfactor[b] = cpv(h, VICGPU_CP_BAND(CPB_TFACTOR, b, Nn, Nb), c);
    sc->Pfactor[b] = cpv(h, VICGPU_CP_BAND(CPB_PFACTOR, b, Nn, Nb), c);
    sc->BandElev[b] = (float)cpv(h, VICGPU_CP_BAND(CPB_BANDELEV, b, Nn, Nb), c);
    sc->AboveTreeLine[b] = (char)cpv(h, VICGPU_CP_BAND(CPB_ABOVETREELINE, b, Nn, Nb), c);
  }
  for (int l = 0; l < VIC_NLAYER + 2; l++)
    for (int i = 0; i < VIC_MAX_ZWTVMOIST; i++) {
      sc->zwtvmoist_zwt[l][i] = cpv(h, VICGPU_CP_ZWT_ZWT(l, i, Nn, Nb), c);
      sc->zwtvmoist_moist[l][i] = cpv(h, VICGPU_CP_ZWT_MOIST(l, i, Nn, Nb), c);
    }
  sc->frost_fract[0] = 1.0;
  sc->gridcel = c;
}

static atmos_data_struct *make_atmos(int NR) {
  atmos_data_struct *a = (atmos_data_struct *)calloc(1, sizeof(atmos_data_struct));
  a->air_temp = (double *)calloc(NR + 1, sizeof(double)); a->channel_in = (double *)calloc(NR + 1, sizeof(double));
  a->density = (double *)calloc(NR + 1, sizeof(double)); a->longwave = (double *)calloc(NR + 1, sizeof(double));
  a->prec = (double *)calloc(NR + 1, sizeof(double)); a->pressure = (double *)calloc(NR + 1, sizeof(double));
  a->shortwave = (double *)calloc(NR + 1, sizeof(double)); a->snowflag = (char *)calloc(NR + 1, sizeof(char));
  a->tskc = (double *)calloc(NR + 1, sizeof(double)); a->vp = (double *)calloc(NR + 1, sizeof(double));
  a->vpd = (double *)calloc(NR + 1, sizeof(double)); a->wind = (double *)calloc(NR + 1, sizeof(double));
  return a;
}

int vicref_set_domain(void *hv, int ncell, int nhru, const double *cell_params, const int *hpi, const double *hpd,
                      const int *cell_off, const int *cell_list) {
  vicref_handle *h = (vicref_handle *)hv;
  h->ncell = ncell; h->nhru = nhru;
  h->cp_nrow = VICGPU_CP_NROW(h->opt.Nnode, h->opt.Nband);
  h->cp.assign(cell_params, cell_params + (size_t)h->cp_nrow * ncell);
  h->cell_off.assign(cell_off, cell_off + ncell + 1);
  h->cell_list.assign(cell_list, cell_list + nhru);
  h->hru_cell.assign(nhru, -1); h->hru_pos.assign(nhru, -1);
  h->cells.clear(); h->cells.resize(ncell);
  for (int c = 0; c < ncell; c++) {
    cell_info_struct &cell = h->cells[c];
    fill_soil_con(h, c, &cell.soil_con);
    memset(&cell.lake_con, 0, sizeof(cell.lake_con));
    cell.lake_con.lake_idx = -1;
    cell.atmos = make_atmos(h->NR);
    int n = cell_off[c + 1] - cell_off[c];
    cell.prcp.hruList.resize(n);
    for (int k = 0; k < n; k++) {
      int g = cell_list[cell_off[c] + k];
      if (hpi[(size_t)HPI_CELL * nhru + g] != c) return -2;
      h->hru_cell[g] = c; h->hru_pos[g] = k;
      HRU &hru = cell.prcp.hruList[k];
      /* the HRU struct has no full constructor: zero the POD members */
      memset(&hru.cell, 0, sizeof(hru.cell));
      memset(&hru.energy, 0, sizeof(hru.energy));
      memset(&hru.snow, 0, sizeof(hru.snow));
      memset(&hru.veg_var, 0, sizeof(hru.veg_var));
      hru.glacier = glac_data_struct();
      hru.veg_con.Cv = hpd[(size_t)HPD_CV * nhru + g];
      for (int l = 0; l < 3; l++) hru.veg_con.root[l] = (float)hpd[(size_t)(HPD_ROOT0 + l) * nhru + g];
      hru.veg_con.vegIndex = hpi[(size_t)HPI_VEG_INDEX * nhru + g];
      hru.veg_con.vegClass = hpi[(size_t)HPI_VEG_CLASS * nhru + g];
      hru.veg_con.sigma_slope = (float)hpd[(size_t)HPD_SIGMA_SLOPE * nhru + g]; hru.veg_con.lag_one = (float)hpd[(size_t)HPD_LAG_ONE * nhru + g];
      hru.veg_con.fetch = (float)hpd[(size_t)HPD_FETCH * nhru + g];
      hru.veg_con.LAKE = 0;
      hru.veg_con.zone_depth = NULL; hru.veg_con.zone_fract = NULL;
      hru.init_STILL_STORM = 0; hru.init_DRY_TIME = 0;
      hru.mu = 1.0;
      hru.isGlacier = hpi[(size_t)HPI_IS_GLACIER * nhru + g] != 0;
      hru.isArtificialBareSoil = hpi[(size_t)HPI_IS_ARTIFICIAL_BARE * nhru + g] != 0;
      hru.bandIndex = hpi[(size_t)HPI_BAND * nhru + g];
      cell.Cv_sum += hru.veg_con.Cv;
    }
  }
  for (int g = 0; g < nhru; g++) if (h->hru_cell[g] < 0) return -3;
  return 0;
}

static void load_atmos(vicref_handle *h, int c, const double *forcing, const unsigned char *snowflag) {
  /* forcing [VIC_NFORCE][NF+1][ncell] for ONE step */
  const int ns = h->NR + 1; const int nc = h->ncell;
  atmos_data_struct *a = h->cells[c].atmos;
  for (int s = 0; s < ns; s++) {
#define FV(v) forcing[((size_t)(v) * ns + s) * nc + c]
    a->air_temp[s] = FV(VIC_F_AIR_TEMP); a->prec[s] = FV(VIC_F_PREC); a->pressure[s] = FV(VIC_F_PRESSURE);
    a->vp[s] = FV(VIC_F_VP); a->vpd[s] = FV(VIC_F_VPD); a->density[s] = FV(VIC_F_DENSITY);
    a->shortwave[s] = FV(VIC_F_SHORTWAVE); a->longwave[s] = FV(VIC_F_LONGWAVE); a->wind[s] = FV(VIC_F_WIND);
#undef FV
    a->snowflag[s] = snowflag ? (char)snowflag[(size_t)s * nc + c] : 0;
    a->channel_in[s] = 0; a->tskc[s] = 0;
  }
}

/* initialize_model_state for every cell (vicNl.c:420-427 -> initializeCell); init_moist [3][ncell] */
int vicref_init_state(void *hv, const double *forcing0, const int *dmy0, const double *init_moist) {
  vicref_handle *h = (vicref_handle *)hv;
  dmy_struct d; d.month = dmy0[VIC_DMY_MONTH]; d.day_in_year = dmy0[VIC_DMY_DAY_IN_YEAR]; d.hour = dmy0[VIC_DMY_HOUR];
  d.day = dmy0[VIC_DMY_DAY]; d.year = dmy0[VIC_DMY_YEAR];
  filep_struct filep; memset(&filep, 0, sizeof(filep));
  for (int c = 0; c < h->ncell; c++) {
    cell_info_struct &cell = h->cells[c];
    for (int l = 0; l < 3; l++) cell.soil_con.init_moist[l] = init_moist[(size_t)l * h->ncell + c];
    load_atmos(h, c, forcing0, NULL);
    int err = initialize_model_state(&cell, d, filep, 1, "", &h->state);
    if (err == ERROR) return -1;
  }
  return 0;
}

/* export the (possibly init-modified) cell parameter table: node geometry and node constants are
 * written by initialize_model_state -> set_node_parameters (soil_conduction.c:142-303) */
int vicref_get_cell_params(void *hv, double *out) {
  vicref_handle *h = (vicref_handle *)hv;
  const int Nn = h->opt.Nnode; const int nc = h->ncell;
  memcpy(out, h->cp.data(), sizeof(double) * h->cp.size());
  for (int c = 0; c < nc; c++) {
    const soil_con_struct &sc = h->cells[c].soil_con;
    for (int n = 0; n < Nn; n++) {
      out[(size_t)VICGPU_CP_NODE(CPN_ZSUM, n, Nn) * nc + c] = sc.Zsum_node[n];
      out[(size_t)VICGPU_CP_NODE(CPN_DZ, n, Nn) * nc + c] = sc.dz_node[n];
      out[(size_t)VICGPU_CP_NODE(CPN_ALPHA, n, Nn) * nc + c] = sc.alpha[n];
      out[(size_t)VICGPU_CP_NODE(CPN_BETA, n, Nn) * nc + c] = sc.beta[n];
      out[(size_t)VICGPU_CP_NODE(CPN_GAMMA, n, Nn) * nc + c] = sc.gamma[n];
      out[(size_t)VICGPU_CP_NODE(CPN_MAX_MOIST, n, Nn) * nc + c] = sc.max_moist_node[n];
      out[(size_t)VICGPU_CP_NODE(CPN_EXPT, n, Nn) * nc + c] = sc.expt_node[n];
      out[(size_t)VICGPU_CP_NODE(CPN_BUBBLE, n, Nn) * nc + c] = sc.bubble_node[n];
    }
  }
  return 0;
}

#define SDP(row) sd[(size_t)(row) * nh + g]
#define SIP(row) si[(size_t)(row) * nh + g]

int vicref_get_state(void *hv, double *sd, int *si) {
  vicref_handle *h = (vicref_handle *)hv;
  vicgpu_binding_state_to_tables(h->cells, h->hru_cell.data(), h->hru_pos.data(), h->nhru, h->opt.Nnode, sd, si);   /* integration/vicgpu_binding.cpp */
  return 0;
}

int vicref_set_state(void *hv, const double *sd, const int *si) {
  vicref_handle *h = (vicref_handle *)hv;
  vicgpu_binding_tables_to_state(h->cells, h->hru_cell.data(), h->hru_pos.data(), h->nhru, h->opt.Nnode, sd, si);
  return 0;
}

static void export_flux(vicref_handle *h, double *fx, double *cell_out, const int *cell_err) {
  const size_t nh = h->nhru;
  for (int g = 0; g < h->nhru; g++) {
    const HRU &u = h->cells[h->hru_cell[g]].prcp.hruList[h->hru_pos[g]];
    const hru_data_struct &cw = u.cell[WET];
    const energy_bal_struct &e = u.energy; const snow_data_struct &s = u.snow;
#define FXP(row) fx[(size_t)(row) * nh + g]
    FXP(FX_RUNOFF) = cw.runoff; FXP(FX_BASEFLOW) = cw.baseflow; FXP(FX_ASAT) = cw.asat; FXP(FX_INFLOW) = cw.inflow;
    for (int l = 0; l < 3; l++) FXP(FX_EVAP0 + l) = cw.layer[l].evap;
    FXP(FX_CANOPYEVAP) = u.veg_var[WET].canopyevap; FXP(FX_THROUGHFALL) = u.veg_var[WET].throughfall;
    FXP(FX_SNOW_VAPOR_FLUX) = s.vapor_flux; FXP(FX_SNOW_CANOPY_VAPOR_FLUX) = s.canopy_vapor_flux;
    FXP(FX_SNOW_BLOWING_FLUX) = s.blowing_flux; FXP(FX_SNOW_SURFACE_FLUX) = s.surface_flux;
    FXP(FX_SNOW_MELT) = s.melt; FXP(FX_SNOW_MASS_ERROR) = s.mass_error; FXP(FX_SNOW_QNET) = s.Qnet;
    for (int l = 0; l < 3; l++) { FXP(FX_FDEPTH0 + l) = e.fdepth[l]; FXP(FX_TDEPTH0 + l) = e.tdepth[l]; FXP(FX_ZWTL0 + l) = cw.layer[l].zwt; }
    for (int p = 0; p < 6; p++) FXP(FX_POT_EVAP0 + p) = cw.pot_evap[p];
    FXP(FX_AERO_RESIST_SURFACE) = cw.aero_resist.surface; FXP(FX_AERO_RESIST_OVERSTORY) = cw.aero_resist.overstory;
    FXP(FX_ROOTMOIST) = cw.rootmoist; FXP(FX_WETNESS) = cw.wetness;
    FXP(FX_ZWT) = cw.zwt; FXP(FX_ZWT2) = cw.zwt2; FXP(FX_ZWT3) = cw.zwt3;
    FXP(FX_ATMOS_LATENT) = e.AtmosLatent; FXP(FX_ATMOS_LATENT_SUB) = e.AtmosLatentSub; FXP(FX_ATMOS_SENSIBLE) = e.AtmosSensible;
    FXP(FX_LONG_UNDER_IN) = e.LongUnderIn; FXP(FX_NET_LONG_ATMOS) = e.NetLongAtmos; FXP(FX_NET_LONG_UNDER) = e.NetLongUnder;
    FXP(FX_NET_SHORT_ATMOS) = e.NetShortAtmos; FXP(FX_NET_SHORT_GRND) = e.NetShortGrnd; FXP(FX_NET_SHORT_UNDER) = e.NetShortUnder;
    FXP(FX_SHORT_UNDER_IN) = e.ShortUnderIn;
    FXP(FX_OUT_PREC) = NAN; FXP(FX_OUT_RAIN) = NAN; FXP(FX_OUT_SNOW) = NAN;  /* not observable per HRU in the reference */
    FXP(FX_GLAC_MASS_BALANCE) = u.glacier.mass_balance; FXP(FX_GLAC_ICE_MASS_BALANCE) = u.glacier.ice_mass_balance;
    FXP(FX_GLAC_ACCUMULATION) = u.glacier.accumulation; FXP(FX_GLAC_MELT) = u.glacier.melt;
    FXP(FX_GLAC_VAPOR_FLUX) = u.glacier.vapor_flux; FXP(FX_GLAC_INFLOW) = u.glacier.inflow;
    FXP(FX_GLAC_OUTFLOW) = u.glacier.outflow; FXP(FX_GLAC_OUTFLOW_COEF) = u.glacier.outflow_coef;
    FXP(FX_GLAC_QNET) = u.glacier.Qnet; FXP(FX_GLAC_COLD_CONTENT) = u.glacier.cold_content;
    FXP(FX_GLACIER_FLUX) = e.glacier_flux; FXP(FX_DELTACC_GLAC) = e.deltaCC_glac; FXP(FX_GLACIER_MELT_ENERGY) = e.glacier_melt_energy;
#undef FXP
  }
  if (cell_out) for (int c = 0; c < h->ncell; c++) {
    const atmos_data_struct *a = h->cells[c].atmos;
    cell_out[(size_t)CO_OUT_PREC * h->ncell + c] = a->out_prec;
    cell_out[(size_t)CO_OUT_RAIN * h->ncell + c] = a->out_rain;
    cell_out[(size_t)CO_OUT_SNOW * h->ncell + c] = a->out_snow;
  }
  (void)cell_err;
}

/* One model step for every cell: the body of the OpenMP loop vicNl.c:514-563 without put_data.
 * forcing [VIC_NFORCE][NF+1][ncell], snowflag [NF+1][ncell], dmy [VIC_NDMY].
 * flux / cell_out / cell_err may be NULL. Returns the number of cells that returned ERROR. */
int vicref_step(void *hv, const double *forcing, const unsigned char *snowflag, const int *dmyv,
                double *flux, double *cell_out, int *cell_err, int nthreads) {
  vicref_handle *h = (vicref_handle *)hv;
  dmy_struct d; d.month = dmyv[VIC_DMY_MONTH]; d.day_in_year = dmyv[VIC_DMY_DAY_IN_YEAR]; d.hour = dmyv[VIC_DMY_HOUR];
  d.day = dmyv[VIC_DMY_DAY]; d.year = dmyv[VIC_DMY_YEAR];
  int nerr = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+:nerr)
  for (int c = 0; c < h->ncell; c++) {
    cell_info_struct &cell = h->cells[c];
    load_atmos(h, c, forcing, snowflag);
    int err = full_energy(FALSE, 0, cell.atmos, &cell.prcp, &d, &cell.lake_con, &cell.soil_con, &cell.writeDebug, &h->state);
    if (cell_err) cell_err[c] = (err == ERROR) ? VICGPU_CELLERR_SOLVER : 0;
    if (err == ERROR) nerr++;
    /* accumulateGlacierMassBalance (vicNl.c:562-563) is a host-side += of glacier.mass_balance; the
       accumulation-window logic is driver state, so the harness applies the += unconditionally once
       cum_mass_balance has been made valid by the caller */
    for (std::vector<HRU>::iterator it = cell.prcp.hruList.begin(); it != cell.prcp.hruList.end(); ++it)
      if (it->isGlacier && IS_VALID(it->glacier.cum_mass_balance) && IS_VALID(it->glacier.mass_balance))
        it->glacier.cum_mass_balance += it->glacier.mass_balance;
  }
  if (flux) export_flux(h, flux, cell_out, cell_err);
  return nerr;
}

int vicref_get_fluxes(void *hv, double *flux) { export_flux((vicref_handle *)hv, flux, NULL, NULL); return 0; }

/* write: streams every cell (cell order) and returns the number of values; vals / ids may be NULL to ask for the count.
 * cell_start[ncell+1] receives where each cell's values begin. */
int vicref_state_stream_write(void *hv, double *vals, int *ids, int cap, int *cell_start) {
  vicref_handle *h = (vicref_handle *)hv;
  VicrefBufferIO io(StateIO::Writer, &h->state);
  for (int c = 0; c < h->ncell; c++) {
    if (cell_start) cell_start[c] = (int)io.vals.size();
    processCellForStateFile(&h->cells[c], &io, &h->state);
  }
  if (cell_start) cell_start[h->ncell] = (int)io.vals.size();
  if (vals && ids) {
    if ((int)io.vals.size() > cap) return -1;
    for (size_t i = 0; i < io.vals.size(); i++) { vals[i] = io.vals[i]; ids[i] = io.ids[i]; }
  }
  return (int)io.vals.size();
}
/* read: feeds a stream (as written above) back through the same function in Reader mode */
int vicref_state_stream_read(void *hv, const double *vals, const int *ids, int n) {
  vicref_handle *h = (vicref_handle *)hv;
  VicrefBufferIO io(StateIO::Reader, &h->state);
  io.vals.assign(vals, vals + n); io.ids.assign(ids, ids + n);
  try {
    for (int c = 0; c < h->ncell; c++) processCellForStateFile(&h->cells[c], &io, &h->state);
  } catch (VICException &e) { return -1; }
  return io.at == (size_t)n ? 0 : -2;
}
/* the ids the tests need to find their way through the stream */
int vicref_state_var_id(const char *name) {
  using namespace StateVariables;
  struct { const char *n; int v; } t[] = {{"HRU_BAND_INDEX", HRU_BAND_INDEX}, {"HRU_VEG_INDEX", HRU_VEG_INDEX}, {"SOIL_DZ_NODE", SOIL_DZ_NODE},
    {"SOIL_ZSUM_NODE", SOIL_ZSUM_NODE}, {"GLAC_MASS_BALANCE_EQN_TERMS", GLAC_MASS_BALANCE_EQN_TERMS}, {"HRU_VEG_VAR_WDEW", HRU_VEG_VAR_WDEW}};
  for (size_t i = 0; i < sizeof(t) / sizeof(t[0]); i++) if (strcmp(name, t[i].n) == 0) return t[i].v;
  return -1;
}

/* ---- the reference-side binding (integration/vicgpu_binding.cpp) on the harness's cells.
 * vicref_binding_tables: what the binding packs from the reference structs (its own HRU numbering), for the round-trip test:
 * the tables the harness was built from must come back.  Buffers sized by the caller from ncell / nhru. */
int vicref_binding_tables(void *hv, double *veglib, double *cell_params, int *hpi, double *hpd, int *cell_off, int *cell_list,
                          double *sd, int *si, vicgpu_options *opt_out) {
  vicref_handle *h = (vicref_handle *)hv;
  VicGpuTables t;
  vicgpu_binding_options(&h->state, opt_out);
  vicgpu_binding_number_hrus(h->cells, t);
  vicgpu_binding_pack_veglib(&h->state, t);
  vicgpu_binding_pack_domain(&h->state, h->cells, t);
  if (t.nhru != h->nhru || t.ncell != h->ncell) return -1;
  t.sd.assign((size_t)VICGPU_SD_NROW(t.Nnode) * t.nhru, 0.0); t.si.assign((size_t)VICGPU_SI_NROW(t.Nnode) * t.nhru, 0);
  vicgpu_binding_state_to_tables(h->cells, t.hru_cell.data(), t.hru_pos.data(), t.nhru, t.Nnode, t.sd.data(), t.si.data());
  memcpy(veglib, t.veglib.data(), sizeof(double) * t.veglib.size());
  memcpy(cell_params, t.cell_params.data(), sizeof(double) * t.cell_params.size());
  memcpy(hpi, t.hpi.data(), sizeof(int) * t.hpi.size()); memcpy(hpd, t.hpd.data(), sizeof(double) * t.hpd.size());
  memcpy(cell_off, t.cell_off.data(), sizeof(int) * t.cell_off.size()); memcpy(cell_list, t.cell_list.data(), sizeof(int) * t.cell_list.size());
  memcpy(sd, t.sd.data(), sizeof(double) * t.sd.size()); memcpy(si, t.si.data(), sizeof(int) * t.si.size());
  return 0;
}

/* nsteps records through VicGpuBinding (libvicgpu.so must be loaded in the process: the vicgpu_* symbols bind lazily): the
 * cells get an atmos[] array of nsteps records like the one initialize_atmos builds, the binding replaces the cell loop of
 * vicNl.c:506-593, and the device state ends up back in the cells' HRU structs.  frozen_compat / node_solver: the two
 * vicgpu_options fields that are not reference options.  nout > 0: put_data runs on the device too and the aggregates of the
 * named variables come back as the writer's float table.  Returns 0, or 100 + the binding's error, flags[ncell] = ERROR flags. */
int vicref_run_through_binding(void *hv, int nsteps, const double *forcing, const unsigned char *snowflag, const int *dmyv, int device,
                               int *flags, int out_step_ratio, int nout, const char *const *out_names, float *outputs) {
  vicref_handle *h = (vicref_handle *)hv;
  const size_t ns = h->NR + 1, nc = h->ncell;
  std::vector<atmos_data_struct *> saved(h->ncell);
  for (int c = 0; c < h->ncell; c++) {
    saved[c] = h->cells[c].atmos;
    atmos_data_struct *arr = (atmos_data_struct *)calloc(nsteps, sizeof(atmos_data_struct));
    for (int r = 0; r < nsteps; r++) { atmos_data_struct *one = make_atmos(h->NR); arr[r] = *one; free(one); }
    h->cells[c].atmos = arr;
  }
  std::vector<dmy_struct> dmy(nsteps);
  for (int r = 0; r < nsteps; r++) {
    memset(&dmy[r], 0, sizeof(dmy_struct));
    dmy[r].month = dmyv[r * VIC_NDMY + VIC_DMY_MONTH]; dmy[r].day_in_year = dmyv[r * VIC_NDMY + VIC_DMY_DAY_IN_YEAR];
    dmy[r].hour = dmyv[r * VIC_NDMY + VIC_DMY_HOUR]; dmy[r].day = dmyv[r * VIC_NDMY + VIC_DMY_DAY]; dmy[r].year = dmyv[r * VIC_NDMY + VIC_DMY_YEAR];
    for (int c = 0; c < h->ncell; c++) {
      atmos_data_struct *keep = h->cells[c].atmos;
      h->cells[c].atmos = &keep[r];
      load_atmos(h, c, forcing + (size_t)r * VIC_NFORCE * ns * nc, snowflag + (size_t)r * ns * nc);
      h->cells[c].atmos = keep;
    }
  }
  int rc = 0;
  {
    /* the two fields of vicgpu_options that are not options of the reference come from the harness's own options */
    VicGpuBinding b(&h->state, h->cells, device, h->opt.frozen_compat, h->opt.NODE_SOLVER);
    if (!b.ok()) rc = 100;
    if (!rc && nout > 0) { int r = b.enable_put_data(out_step_ratio); if (r) rc = 100 - r; }
    if (!rc) { int r = b.run(0, nsteps, dmy.data()); if (r) rc = 100 - r; }
    if (!rc && nout > 0) {          /* the aggregates of the output interval that ends with the last record */
      std::vector<std::string> names(out_names, out_names + nout);
      std::vector<float> o;
      int r = b.outputs(names, o, true);
      if (r < 0) rc = 100 - r; else memcpy(outputs, o.data(), sizeof(float) * o.size());
    }
    if (!rc) { int r = b.finish(flags); if (r) rc = 100 - r; }
  }
  for (int c = 0; c < h->ncell; c++) {
    atmos_data_struct *arr = h->cells[c].atmos;
    for (int r = 0; r < nsteps; r++) {
      free(arr[r].air_temp); free(arr[r].channel_in); free(arr[r].density); free(arr[r].longwave); free(arr[r].prec); free(arr[r].pressure);
      free(arr[r].shortwave); free(arr[r].snowflag); free(arr[r].tskc); free(arr[r].vp); free(arr[r].vpd); free(arr[r].wind);
    }
    free(arr);
    h->cells[c].atmos = saved[c];
  }
  return rc;
}

/* ---- the reference's own put_data (put_data.c:7) on the harness's cells, for pinning oracle/orc_putdata.c.
 * rec < 0: the initialisation call of vicNl.c:524-541; otherwise the harness keeps ONE atmos record per cell, so the
 * call is made with rec = 0 (put_data only indexes cell->atmos with it and compares it with 0 / nrecs-1 for messages).
 * forcing / cell_out: the step's forcing and out_prec/out_rain/out_snow (vicref_step's cell_out). */
int vicref_put_data(void *hv, int rec, const double *forcing, const double *cell_out, int out_step_ratio) {
  vicref_handle *h = (vicref_handle *)hv;
  h->state.out_step_ratio = out_step_ratio;
  if (!h->out_template) h->out_template = create_output_list(&h->state);
  if (h->out.empty()) {
    for (int c = 0; c < h->ncell; c++) copy_output_data(h->out, h->out_template, &h->state);   /* one list per cell (vicNl.c) */
  }
  dmy_struct d; memset(&d, 0, sizeof(d));
  for (int c = 0; c < h->ncell; c++) {
    cell_info_struct &cell = h->cells[c];
    if (rec >= 0) {
      load_atmos(h, c, forcing, NULL);
      cell.atmos->out_prec = cell_out[(size_t)CO_OUT_PREC * h->ncell + c];
      cell.atmos->out_rain = cell_out[(size_t)CO_OUT_RAIN * h->ncell + c];
      cell.atmos->out_snow = cell_out[(size_t)CO_OUT_SNOW * h->ncell + c];
    }
    int err = put_data(&cell, NULL, h->out[c], &d, rec < 0 ? -1 : 0, &h->state);
    if (err == ERROR) return -1;
  }
  return 0;
}

/* number of variables of the reference's list; name / nelem / aggtype of variable v */
int vicref_out_nvar(void) { return N_OUTVAR_TYPES; }
int vicref_out_info(void *hv, int v, char *name, int *nelem, int *aggtype) {
  vicref_handle *h = (vicref_handle *)hv;
  if (!h->out_template) h->out_template = create_output_list(&h->state);
  if (v < 0 || v >= N_OUTVAR_TYPES) return -1;
  strcpy(name, h->out_template[v].varname.c_str());
  /* three variables get no varname in output_list_utils.c (they cannot be selected in a global file); the harness names
   * them after their enum so that the tests can address them */
  if (v == OUT_AREA_BAND) strcpy(name, "OUT_AREA_BAND");
  if (v == OUT_ELEV_BAND) strcpy(name, "OUT_ELEV_BAND");
  if (v == OUT_GLAC_MELT_ENERGY) strcpy(name, "OUT_GLAC_MELT_ENERGY");
  *nelem = h->out_template[v].nelem;
  *aggtype = h->out_template[v].aggtype;   /* AGG_TYPE_* */
  return 0;
}
/* which: 0 = data, 1 = aggdata; out [nelem][ncell]; returns nelem */
int vicref_get_output(void *hv, int v, int which, double *out) {
  vicref_handle *h = (vicref_handle *)hv;
  if (h->out.empty() || v < 0 || v >= N_OUTVAR_TYPES) return -1;
  const int ne = h->out_template[v].nelem;
  for (int c = 0; c < h->ncell; c++)
    for (int i = 0; i < ne; i++) out[(size_t)i * h->ncell + c] = which ? h->out[c][v].aggdata[i] : h->out[c][v].data[i];
  return ne;
}
int vicref_reset_agg(void *hv) {       /* vicNl.c:599-606 */
  vicref_handle *h = (vicref_handle *)hv;
  for (size_t c = 0; c < h->out.size(); c++)
    for (int v = 0; v < N_OUTVAR_TYPES; v++)
      for (int i = 0; i < h->out_template[v].nelem; i++) h->out[c][v].aggdata[i] = 0;
  return 0;
}
/* save_data, cellErrors and fallBackStats of every cell: [PB_NROW][ncell] (include/vicgpu_out.h) */
int vicref_get_balance(void *hv, double *pb) {
  vicref_handle *h = (vicref_handle *)hv;
  const size_t nc = h->ncell;
  for (int c = 0; c < h->ncell; c++) {
    const cell_info_struct &cell = h->cells[c];
    pb[PB_SAVE_TOTAL_SOIL_MOIST * nc + c] = cell.save_data.total_soil_moist; pb[PB_SAVE_SWE * nc + c] = cell.save_data.swe;
    pb[PB_SAVE_WDEW * nc + c] = cell.save_data.wdew; pb[PB_SAVE_SURFSTOR * nc + c] = cell.save_data.surfstor;
    pb[PB_WATER_LAST_STORAGE * nc + c] = cell.cellErrors.water_last_storage; pb[PB_WATER_CUM_ERROR * nc + c] = cell.cellErrors.water_cum_error;
    pb[PB_WATER_MAX_ERROR * nc + c] = cell.cellErrors.water_max_error; pb[PB_ENERGY_CUM_ERROR * nc + c] = cell.cellErrors.energy_cum_error;
    pb[PB_ENERGY_MAX_ERROR * nc + c] = cell.cellErrors.energy_max_error;
    pb[PB_FB_TFOLIAGE * nc + c] = cell.fallBackStats.Tfoliage_fbcount_total; pb[PB_FB_TCANOPY * nc + c] = cell.fallBackStats.Tcanopy_fbcount_total;
    pb[PB_FB_TSNOWSURF * nc + c] = cell.fallBackStats.Tsnowsurf_fbcount_total; pb[PB_FB_TSURF * nc + c] = cell.fallBackStats.Tsurf_fbcount_total;
    pb[PB_FB_TSOIL * nc + c] = cell.fallBackStats.Tsoil_fbcount_total; pb[PB_FB_TGLACSURF * nc + c] = cell.fallBackStats.Tglacsurf_fbcount_total;
  }
  return 0;
}

/* Timed multi-step run for bench.py's cpu_baseline leg: forcing [nsteps][VIC_NFORCE][NF+1][ncell].
 * Returns wall seconds. */
double vicref_run(void *hv, int nsteps, const double *forcing, const unsigned char *snowflag, const int *dmyv, int nthreads) {
  vicref_handle *h = (vicref_handle *)hv;
  const size_t fstride = (size_t)VIC_NFORCE * (h->NR + 1) * h->ncell;
  const size_t sstride = (size_t)(h->NR + 1) * h->ncell;
  auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < nsteps; s++)
    vicref_step(hv, forcing + s * fstride, snowflag ? snowflag + s * sstride : NULL, dmyv + (size_t)s * VIC_NDMY, NULL, NULL, NULL, nthreads);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

/* the reference's own GlacierMassBalanceResult + resetAccumulationValues (accumulateGlacierMassBalance.c:53-66) */
int vicref_glacier_fit(void *hv, double *eq, int reset) {
  vicref_handle *h = (vicref_handle *)hv;
  if (!h || !eq) return -1;
  dmy_struct d;
  memset(&d, 0, sizeof(d));
  for (int c = 0; c < h->ncell; c++) {
    cell_info_struct &cell = h->cells[c];
    GlacierMassBalanceResult result(cell.prcp.hruList, &cell.soil_con, d);
    eq[(size_t)GMB_B0 * h->ncell + c] = result.equation.b0;
    eq[(size_t)GMB_B1 * h->ncell + c] = result.equation.b1;
    eq[(size_t)GMB_B2 * h->ncell + c] = result.equation.b2;
    eq[(size_t)GMB_FIT_ERROR * h->ncell + c] = result.equation.fitError;
    if (reset) resetAccumulationValues(&cell.prcp.hruList);
  }
  return 0;
}

/* the pure functions of the path one by one (include/vicgpu.h VICGPU_PURE_*): the reference's own functions */
int vicref_pure(void *hv, int fn, int n, const double *in, double *out) {
  vicref_handle *h = (vicref_handle *)hv;
  if (!h || n < 0 || !in || !out) return -1;
  const ProgramState *st = &h->state;
  const soil_con_struct *sc = h->cells.empty() ? NULL : &h->cells[0].soil_con;
  for (int i = 0; i < n; i++) {
    const double *a = in + (size_t)i * VICGPU_PURE_NIN;
    double r;
    switch (fn) {
      case VICGPU_PURE_SVP: r = svp(a[0]); break;
      case VICGPU_PURE_SVP_SLOPE: r = svp_slope(a[0]); break;
      case VICGPU_PURE_CALC_RAINONLY: r = calc_rainonly(a[0], a[1], a[2], a[3], 1.0, st); break;
      case VICGPU_PURE_SNOW_ALBEDO:
        if (!sc) return -1;
        r = snow_albedo(a[0], a[1], a[2], a[3], a[4], a[5], (int)a[6], a[7] != 0.0, sc, st); break;
      case VICGPU_PURE_NEW_SNOW_DENSITY: r = new_snow_density(a[0], st); break;
      case VICGPU_PURE_STABILITY: r = StabilityCorrection(a[0], a[1], a[2], a[3], a[4], a[5]); break;
      case VICGPU_PURE_PENMAN: r = penman(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
      case VICGPU_PURE_CALC_RC: r = calc_rc(a[0], a[1], (float)a[2], a[3], a[4], a[5], a[6], a[7] != 0.0 ? TRUE : FALSE); break;
      case VICGPU_PURE_ESTIMATE_T1: r = estimate_T1(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[7], a[8], a[9]); break;
      case VICGPU_PURE_SOIL_CONDUCTIVITY: r = soil_conductivity(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
      case VICGPU_PURE_VOL_HEAT_CAPACITY: r = volumetric_heat_capacity(a[0], a[1], a[2], a[3]); break;
      case VICGPU_PURE_MAX_UNFROZEN_WATER: r = maximum_unfrozen_water(a[0], a[1], a[2], a[3]); break;
      case VICGPU_PURE_LINEAR_INTERP: r = linear_interp(a[0], a[1], a[2], a[3], a[4]); break;
      case VICGPU_PURE_VEG_HEIGHT: r = calc_veg_height(a[0], a[1]); break;
      default: return -1;
    }
    out[i] = r;
  }
  return 0;
}

/* atmos[rec] of every cell from sub-daily forcing supplied at `force_dt`-hour steps (force_dt = 1, or = snow_step): the
 * reference's initialize_atmos on in-memory records.  file[v][k][cell], v = VIC_RAW_* (file units), k < nsteps * dt / force_dt.
 * Out: forcing [nsteps][VIC_NFORCE][NR+1][ncell] and snowflag [nsteps][NR+1][ncell] in the layout of vicgpu_push_forcing.
 * nsteps * dt must be whole days starting at hour 0 (the cell's time zone is its own longitude: no local-time shift). */
int vicref_derive_forcing(void *hv, int nsteps, int force_dt, const double *file, double min_wind, int plapse, double *forcing,
                          unsigned char *snowflag) {
  vicref_handle *h = (vicref_handle *)hv;
  const int dt = h->opt.dt, NR = h->NR, NF = h->NF, nc = h->ncell;
  if (nsteps <= 0 || (nsteps * dt) % 24 != 0 || (force_dt != 1 && force_dt != h->opt.snow_step)) return -1;
  if (force_dt == 1 && h->opt.snow_step != 1) return -1;        /* forcing_data[] holds nrecs * NF values (read_forcing_data.c:31) */
  const int nfile = nsteps * dt / force_dt;
  ProgramState st = h->state;
  st.global_param.nrecs = nsteps;
  st.global_param.starthour = 0; st.global_param.startyear = 2001; st.global_param.startmonth = 1; st.global_param.startday = 1;
  st.global_param.forceskip[0] = 0; st.global_param.forceskip[1] = 0;
  st.options.MIN_WIND_SPEED = min_wind;
  st.options.PLAPSE = plapse ? TRUE : FALSE;
  st.options.OUTPUT_FORCE = FALSE;
  st.options.COMPUTE_TREELINE = FALSE;
  st.options.ALMA_INPUT = FALSE;
  for (int t = 0; t < N_FORCING_TYPES; t++) st.param_set.TYPE[t].SUPPLIED = 0;
  const int types[VIC_NRAW] = {AIR_TEMP, PREC, PRESSURE, VP, SHORTWAVE, LONGWAVE, WIND};
  const int raws[VIC_NRAW] = {VIC_RAW_AIR_TEMP, VIC_RAW_PREC, VIC_RAW_PRESSURE_KPA, VIC_RAW_VP_KPA, VIC_RAW_SHORTWAVE, VIC_RAW_LONGWAVE, VIC_RAW_WIND};
  for (int i = 0; i < VIC_NRAW; i++) st.param_set.TYPE[types[i]].SUPPLIED = 1;
  st.param_set.FORCE_DT[0] = force_dt; st.param_set.FORCE_DT[1] = INVALID_INT;
  std::vector<dmy_struct> dmy(nsteps);
  for (int r = 0; r < nsteps; r++) {
    memset(&dmy[r], 0, sizeof(dmy_struct));
    dmy[r].hour = (r * dt) % 24; dmy[r].day = 1 + (r * dt) / 24; dmy[r].day_in_year = dmy[r].day; dmy[r].month = 1; dmy[r].year = 2001;
  }
  std::vector<double> col((size_t)VIC_NRAW * nfile);
  for (int c = 0; c < nc; c++) {
    soil_con_struct sc;
    fill_soil_con(h, c, &sc);
    /* what MTCLIM reads besides (its estimates of shortwave / vapour pressure are not used when both are supplied; tskc is
     * not on the path) */
    sc.lng = -120.f; sc.time_zone_lng = -120.f; sc.slope = 0; sc.aspect = 0; sc.ehoriz = 0; sc.whoriz = 0; sc.annual_prec = 800.;
    sc.cell_area = 3.6e7;
    VicrefForcingSource src;
    for (int t = 0; t < N_FORCING_TYPES; t++) src.data[t] = NULL;
    src.nrec = nfile;
    for (int i = 0; i < VIC_NRAW; i++) {
      for (int k = 0; k < nfile; k++) col[(size_t)i * nfile + k] = file[((size_t)raws[i] * nfile + k) * nc + c];
      src.data[types[i]] = &col[(size_t)i * nfile];
    }
    atmos_data_struct *atmos = (atmos_data_struct *)calloc(nsteps, sizeof(atmos_data_struct));
    for (int r = 0; r < nsteps; r++) { atmos_data_struct *a = make_atmos(NR); atmos[r] = *a; free(a); }
    vicref_forcing_source = &src;
    FILE *infile[2] = {NULL, NULL};
    int ncids[2] = {0, 0};
    int rc = 0;
    try { initialize_atmos(atmos, dmy.data(), infile, ncids, &sc, &st); } catch (...) { rc = -2; }
    vicref_forcing_source = NULL;
    for (int r = 0; r < nsteps && rc == 0; r++) {
      const atmos_data_struct &a = atmos[r];
      for (int j = 0; j <= NR; j++) {
        double *f = forcing + ((size_t)r * VIC_NFORCE * (NR + 1)) * nc;
#define PUTF(v, x) f[((size_t)(v) * (NR + 1) + j) * nc + c] = (x)
        PUTF(VIC_F_AIR_TEMP, a.air_temp[j]); PUTF(VIC_F_PREC, a.prec[j]); PUTF(VIC_F_PRESSURE, a.pressure[j]); PUTF(VIC_F_VP, a.vp[j]);
        PUTF(VIC_F_VPD, a.vpd[j]); PUTF(VIC_F_DENSITY, a.density[j]); PUTF(VIC_F_SHORTWAVE, a.shortwave[j]);
        PUTF(VIC_F_LONGWAVE, a.longwave[j]); PUTF(VIC_F_WIND, a.wind[j]);
#undef PUTF
        snowflag[((size_t)r * (NR + 1) + j) * nc + c] = a.snowflag[j] ? 1 : 0;
      }
    }
    for (int r = 0; r < nsteps; r++) {
      atmos_data_struct &a = atmos[r];
      free(a.air_temp); free(a.channel_in); free(a.density); free(a.longwave); free(a.prec); free(a.pressure); free(a.shortwave);
      free(a.snowflag); free(a.tskc); free(a.vp); free(a.vpd); free(a.wind);
    }
    free(atmos);
    free(sc.BandElev); free(sc.AreaFract); free(sc.Pfactor); free(sc.Tfactor); free(sc.AboveTreeLine);
    if (rc) return rc;
  }
  return 0;
}

int vicref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

} /* extern "C" */
