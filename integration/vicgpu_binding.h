// vicgpu_binding.h -- the reference-side binding of libvicgpu.so: what a maintainer adds to pacificclimate/VIC so that
// runModel() (vicNl.c:506-610) runs its cell loop on the GPU.  Compiles against the reference's own headers (vicNl.h);
// this repository builds it only as part of oracle/ref_build (where /root/reference is present), links it into the
// reference harness and runs it end to end in tests/test_binding.py: reference structs -> tables -> libvicgpu.so ->
// reference structs, against the reference's own full_energy on a second copy of the same cells.
#ifndef VICGPU_BINDING_H_
#define VICGPU_BINDING_H_
#include <string>
#include <vector>
#include "vicNl.h"
#include "vicgpu.h"
#include "vicgpu_out.h"

struct VicGpuTables {
  int ncell = 0, nhru = 0, Nnode = 0, Nband = 0, nveg_rows = 0;
  std::vector<int> hru_cell, hru_pos;          // HRU id -> (cell index, position in the cell's hruList)
  std::vector<int> cell_off, cell_list;        // CSR: every cell's HRUs in hruList order
  std::vector<double> veglib, cell_params, hpd, sd;
  std::vector<int> hpi, si;
};

// options / global parameters -> vicgpu_options
void vicgpu_binding_options(const ProgramState *state, vicgpu_options *opt);
// HRU numbering: position-major (all cells' first HRU, then all second ones, ...), so that a 64-lane wavefront holds the
// same tile / band of 64 neighbouring cells wherever the domain is regular
void vicgpu_binding_number_hrus(const std::vector<cell_info_struct> &cells, VicGpuTables &t);
void vicgpu_binding_pack_veglib(const ProgramState *state, VicGpuTables &t);                                 // veg_lib_struct[]
void vicgpu_binding_pack_domain(const ProgramState *state, const std::vector<cell_info_struct> &cells, VicGpuTables &t);   // soil_con, veg_con, bands
// HRU state <-> SD_* / SI_* rows, for any numbering
void vicgpu_binding_state_to_tables(const std::vector<cell_info_struct> &cells, const int *hru_cell, const int *hru_pos, int nhru, int Nn,
                                    double *sd, int *si);
void vicgpu_binding_tables_to_state(std::vector<cell_info_struct> &cells, const int *hru_cell, const int *hru_pos, int nhru, int Nn,
                                    const double *sd, const int *si);
// cell.atmos[rec] of every cell -> one step of the forcing chunk [VIC_NFORCE][NF+1][ncell] + snowflag [NF+1][ncell]
void vicgpu_binding_pack_forcing(const std::vector<cell_info_struct> &cells, int rec, int NR, double *forcing, unsigned char *snowflag);

// The replacement of the OpenMP cell loop: construct once after initializeCell() of every cell, call run() per chunk of
// records, finish() before write_model_state / at the end.
class VicGpuBinding {
public:
  // frozen_compat: 1 = FROZEN_SOIL exactly as the reference ships it (frozen_soil.c:218-221), 0 = the node arrays;
  // node_solver: VIC_NODE_SOLVER_BRENT (the reference's iteration replayed) or _NEWTON (converged, faster)
  VicGpuBinding(const ProgramState *state, std::vector<cell_info_struct> &cells, int device, int frozen_compat = 1,
                int node_solver = VIC_NODE_SOLVER_BRENT);
  ~VicGpuBinding();
  bool ok() const { return ctx != NULL; }
  const char *error() const;
  // put_data on the device (dist_prec.c:167): call once before the first run(); out_step_ratio = out_dt / dt.  The per-HRU
  // values initialize_model_state left outside the state tables (frost fronts, ...) go with it.
  int enable_put_data(int out_step_ratio);
  // records [rec0, rec0 + nrec) of cell.atmos[] with their dates; returns 0 or a VICGPU_ERR_*
  int run(int rec0, int nrec, const dmy_struct *dmy);
  // OutputData.aggdata of the named variables ("OUT_RUNOFF", ...) as write_data_all_cells wants them: float
  // [sum nelem][ncell], cells in the order of the vector; reset = vicNl.c:599-606.  Returns the number of rows or < 0.
  int outputs(const std::vector<std::string> &names, std::vector<float> &out, bool reset);
  // device state -> the cells' HRU structs; per-cell ERROR flags (vicNl.c:545-559) into flags[ncell] if given
  int finish(int *flags);
  VicGpuTables tables;
private:
  const ProgramState *state;
  std::vector<cell_info_struct> &cells;
  vicgpu_ctx *ctx;
};
#endif
