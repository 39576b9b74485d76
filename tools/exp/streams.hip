// experiment: do kernels on different HIP streams overlap on this box?   hipcc --offload-arch=gfx950 -O2 streams.hip -o streams
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
__global__ void spin(long long cycles, int* out) {
  long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
int main(int argc, char** argv) {
  int nstream = argc > 1 ? atoi(argv[1]) : 2;
  int nblk = argc > 2 ? atoi(argv[2]) : 64;
  int threaded = argc > 3 ? atoi(argv[3]) : 0;
  int* d; hipMalloc(&d, 4);
  std::vector<hipStream_t> st(nstream);
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[0], 1000, d);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(spin, dim3(nblk), dim3(64), 0, st[0], 100000000LL, d);
  hipDeviceSynchronize();
  double one = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  t0 = std::chrono::steady_clock::now();
  if (!threaded) {
    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(nblk), dim3(64), 0, s, 100000000LL, d);
  } else {
    std::vector<std::thread> th;
    for (auto& s : st) th.emplace_back([&s, nblk, d]() { hipSetDevice(0); hipLaunchKernelGGL(spin, dim3(nblk), dim3(64), 0, s, 100000000LL, d); hipStreamSynchronize(s); });
    for (auto& t : th) t.join();
  }
  hipDeviceSynchronize();
  double all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("streams=%d blocks=%d threaded=%d: one kernel %.2f ms, %d kernels on %d streams %.2f ms\n", nstream, nblk, threaded, one, nstream, nstream, all);
  return 0;
}
