// hip/hip_runtime.h stand-in for the SANITIZER build of vic_amd/csrc/vicgpu_api.hip (tools/hostemu/build.sh).
//
// Test infrastructure only.  GPU AddressSanitizer is not available on the MI355X pool, so to put the device code under
// ASan / UBSan the translation unit is compiled as plain C++ for the host with this header first on the include path:
// kernels become ordinary functions, a launch runs the grid block after block with one fiber per work-item, and
// the handful of wave-level operations the kernels use (__ballot, __any, __shfl, readlane, readfirstlane,
// __syncthreads) are rendezvous points between the fibers of a wave.  "Device" memory is malloc'ed, so out-of-bounds
// accesses to the state tables and to kernel-local arrays are both visible to the sanitizer; HOSTEMU_POISON=1 fills
// fresh device allocations with NaN patterns so reads of never-written words show up in the outputs.
//
// The product (vic_amd/libvicgpu.so) is never built from this header and vic_amd.api only loads the result when
// VICGPU_LIB names it explicitly; it is orders of magnitude slower than the oracle and exists to find bugs.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <ucontext.h>
#include <vector>

#define VIC_HOSTEMU 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef struct hostemu_stream* hipStream_t;
struct hostemu_event { std::chrono::steady_clock::time_point t; };
typedef hostemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
#define HIP_SYMBOL(x) (&(x))

namespace hostemu {

// One wave = up to 64 fibers (ucontext) scheduled round-robin on the launching thread.  A cross-lane operation parks the
// lane; when every lane still in the kernel is parked at one, the operation completes and the lanes run on.  A lane that
// leaves the kernel simply stops taking part, which is what an exited lane does to the hardware's exec mask.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTEMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

struct Wave {
  enum { STACK = 2 << 20 };
  ucontext_t sched, ctx[64];
  char* stack[64] = {};
  bool done[64];
  int nlane = 0, cur = -1;
  unsigned long long present = 0, snap_mask = 0;
  uint64_t slot[64], snap[64];
  const std::function<void()>* body = nullptr;
  unsigned block = 0, block_y = 0, first_thread = 0;
  const void* sched_bottom = nullptr;
  size_t sched_size = 0;

  void park(bool leaving) {                                   // lane -> scheduler
#ifdef HOSTEMU_ASAN
    void* fs = nullptr;
    __sanitizer_start_switch_fiber(leaving ? nullptr : &fs, sched_bottom, sched_size);
#endif
    const int me = cur;
    swapcontext(&ctx[me], &sched);
#ifdef HOSTEMU_ASAN
    __sanitizer_finish_switch_fiber(fs, nullptr, nullptr);
#endif
  }
  unsigned long long gather(int l, uint64_t v, uint64_t* out) {
    slot[l] = v; present |= 1ull << l;
    park(false);
    if (out) memcpy(out, snap, sizeof(snap));
    return snap_mask;
  }
  static void entry(unsigned lo, unsigned hi) {
    Wave* w = reinterpret_cast<Wave*>(((uintptr_t)hi << 32) | lo);
#ifdef HOSTEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &w->sched_bottom, &w->sched_size);
#endif
    (*w->body)();
    w->done[w->cur] = true;
    w->park(true);
  }
  void run() {
    for (int l = 0; l < nlane; l++) {
      if (!stack[l]) stack[l] = (char*)malloc(STACK);          // kept for the life of the process
#ifdef HOSTEMU_ASAN
      __asan_unpoison_memory_region(stack[l], STACK);          // frames of the previous fiber on this stack never unwound
#endif
      done[l] = false;
      getcontext(&ctx[l]);
      ctx[l].uc_stack.ss_sp = stack[l]; ctx[l].uc_stack.ss_size = STACK; ctx[l].uc_link = nullptr;
      const uintptr_t self = (uintptr_t)this;
      makecontext(&ctx[l], (void (*)())entry, 2, (unsigned)(self & 0xffffffffu), (unsigned)(self >> 32));
    }
    int left = nlane;
    while (left > 0) {
      for (int l = 0; l < nlane; l++) {
        if (done[l]) continue;
        cur = l;
        threadIdx = dim3(first_thread + (unsigned)l); blockIdx = dim3(block, block_y);
        hostemu_set_lane(this, l);
#ifdef HOSTEMU_ASAN
        void* fs = nullptr;
        __sanitizer_start_switch_fiber(&fs, stack[l], STACK);
#endif
        swapcontext(&sched, &ctx[l]);
#ifdef HOSTEMU_ASAN
        __sanitizer_finish_switch_fiber(fs, nullptr, nullptr);
#endif
        if (done[l]) left--;
      }
      // every lane still in the kernel is now parked at a cross-lane operation: complete it
      memcpy(snap, slot, sizeof(snap));
      snap_mask = present;
      present = 0;
    }
  }
  static void hostemu_set_lane(Wave* w, int l);
};
inline thread_local Wave* wave;
inline thread_local int lane;
inline void Wave::hostemu_set_lane(Wave* w, int l) { wave = w; lane = l; }

template <typename T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

inline unsigned long long ballot(bool p) {
  uint64_t all[64];
  const unsigned long long mask = wave->gather(lane, p ? 1 : 0, all);
  unsigned long long r = 0;
  for (int i = 0; i < 64; i++) if (((mask >> i) & 1) && all[i]) r |= 1ull << i;
  return r;
}
template <typename T> inline T shfl(T v, int src) {
  uint64_t all[64];
  const unsigned long long mask = wave->gather(lane, to_bits(v), all);
  src &= 63;
  if (!((mask >> src) & 1)) return v;           // reading an inactive lane is undefined on the hardware; keep our own value
  return from_bits<T>(all[src]);
}
template <typename T> inline T readfirstlane(T v) {
  uint64_t all[64];
  const unsigned long long mask = wave->gather(lane, to_bits(v), all);
  return from_bits<T>(all[__builtin_ctzll(mask)]);
}

// Blocks run one after another, and within a block wave after wave (no kernel here synchronises across waves).
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  static std::mutex one_launch;                  // kernels of concurrent host threads (chunks) run one after another
  std::lock_guard<std::mutex> lk(one_launch);
  gridDim = grid; blockDim = block;
  const int nthread = (int)block.x, nwave = (nthread + 63) / 64;
  static thread_local Wave* w = new Wave;
  for (unsigned by = 0; by < grid.y; by++)
    for (unsigned b = 0; b < grid.x; b++)
      for (int i = 0; i < nwave; i++) {
        w->nlane = std::min(64, nthread - 64 * i);
        w->body = &body; w->block = b; w->block_y = by; w->first_thread = 64u * i; w->present = 0;
        w->run();
      }
}

inline void* device_alloc(size_t n) {
  void* p = malloc(n ? n : 1);
  static const bool poison = getenv("HOSTEMU_POISON") && atoi(getenv("HOSTEMU_POISON"));
  if (p && poison) memset(p, 0xFF, n);           // doubles: NaN; ints: -1
  return p;
}

}  // namespace hostemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hostemu::launch(grid, block, [=]() { kernel(__VA_ARGS__); })

inline int __lane_id() { return hostemu::lane; }
inline unsigned long long __ballot(bool p) { return hostemu::ballot(p); }
inline bool __any(bool p) { return hostemu::ballot(p) != 0; }
template <typename T> inline T __shfl(T v, int src) { return hostemu::shfl(v, src); }
#define __builtin_amdgcn_readlane(v, l) hostemu::shfl((v), (l))
#define __builtin_amdgcn_readfirstlane(v) hostemu::readfirstlane((v))
inline void __syncthreads() {
  if (blockDim.x > 64) { fprintf(stderr, "hostemu: __syncthreads in a multi-wave block is not emulated\n"); abort(); }
  hostemu::wave->gather(hostemu::lane, 0, nullptr);
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline long long __double_as_longlong(double v) { return hostemu::from_bits<long long>(hostemu::to_bits(v)); }
inline double __longlong_as_double(long long v) { return hostemu::from_bits<double>(hostemu::to_bits(v)); }
template <typename T, typename U> inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }

// ---- runtime API subset: everything is synchronous, streams and events are tokens ----
inline const char* hipGetErrorString(hipError_t) { return "hostemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }
template <typename K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 1; return hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)hostemu::device_alloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t width, size_t height, hipStream_t = nullptr) {
  for (size_t r = 0; r < height; r++) memset((char*)d + r * pitch, v, width);
  return hipSuccess;
}
inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n) { memcpy(d, sym, n); return hipSuccess; }
inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n) { memcpy(sym, s, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostemu_event; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hostemu_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
