// FETCH_SIZE calibration (MI355X_MICROARCH.md: on gfx950 the counter reports half the bytes of a 16-byte-per-lane streaming
// read).  The parked-context kernels read 8 bytes per lane from wave slabs [wave][word][lane] (VIC_CTX_PAIR = 0) or 16 bytes
// per lane from [wave][word / 2][lane][2] (the shipped layout); the item blocks are read as 880 contiguous bytes per lane by
// dwordx4 loads.  Each kernel below reads a known number of bytes once, from a buffer far larger than the caches:
//   mode 0: slab, 8 B / lane / access      mode 1: slab, 16 B / lane / access      mode 2: per-lane contiguous blocks, 16 B / access
// Usage (GPU box): hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o /tmp/fetch_calib
//                  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- /tmp/fetch_calib
// tools/calib/fetch_calib.py turns the CSV into bytes-per-counted-byte factors.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int WORDS = 96;                 // words per lane and wave, like the parked context
__global__ __launch_bounds__(64) void read_slab8(const double* __restrict__ p, double* out) {
  const double* q = p + (size_t)blockIdx.x * WORDS * 64 + threadIdx.x;
  double s = 0;
#pragma unroll
  for (int w = 0; w < WORDS; w++) s += q[(size_t)w * 64];
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(64) void read_slab16(const double2* __restrict__ p, double* out) {
  const double2* q = p + (size_t)blockIdx.x * (WORDS / 2) * 64 + threadIdx.x;
  double s = 0;
#pragma unroll
  for (int w = 0; w < WORDS / 2; w++) { const double2 v = q[(size_t)w * 64]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(64) void read_block16(const double2* __restrict__ p, double* out) {
  const double2* q = p + ((size_t)blockIdx.x * 64 + threadIdx.x) * (WORDS / 2);
  double s = 0;
#pragma unroll
  for (int w = 0; w < WORDS / 2; w++) { const double2 v = q[w]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
int main() {
  const size_t nwave = 1u << 16;                                   // 64k waves x 96 words x 64 lanes x 8 B = 3.2 GB
  const size_t bytes = nwave * WORDS * 64 * sizeof(double);
  double *buf, *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(read_slab8, dim3(nwave), dim3(64), 0, 0, buf, out);
    hipLaunchKernelGGL(read_slab16, dim3(nwave), dim3(64), 0, 0, (const double2*)buf, out);
    hipLaunchKernelGGL(read_block16, dim3(nwave), dim3(64), 0, 0, (const double2*)buf, out);
  }
  hipDeviceSynchronize();
  printf("bytes_per_launch %zu\n", bytes);
  return 0;
}
