// Handlers for a -fsanitize=array-bounds build of the device code (debug only; see tools/exp/README in DESIGN.md "sanitizers").
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
struct UbSrcLoc { const char* file; unsigned line, col; };
struct UbOobData { UbSrcLoc loc; void* array_type; void* index_type; };
__device__ unsigned ub_count;
extern "C" __device__ __attribute__((noinline)) void __ubsan_handle_out_of_bounds(UbOobData* d, unsigned long idx) {
  if (atomicAdd(&ub_count, 1u) < 16u) printf("UBSAN array index out of bounds: %s line %u col %u index %ld\n", d->loc.file, d->loc.line, d->loc.col, (long)idx);
}
