// tests/emu/hip/hip_runtime.h - TEST INFRASTRUCTURE ONLY.
//
// A host stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel and host sources of battgp_amd/csrc
// be compiled for the build container's CPU ("the CPU build": logic tests of the kernels' indexing and
// algebra while no GPU is at hand, and a target for sanitizers, which the GPU pool does not offer).  Threads
// of a workgroup run as cooperative fibers; __syncthreads and the cross-lane operations (readlane, shuffles,
// v_mfma_f64_16x16x4) are rendezvous points with the documented gfx950 lane layouts.  By default every stream
// operation runs at once, in host order.  With HIPEMU_SCHED = lazy | eager | random[:seed] the streams are real
// queues: work is deferred and executed in an order that honours ONLY what the stream / event graph orders (in-stream
// order, hipStreamWaitEvent edges, the host synchronisation calls) and is otherwise adversarial - a missing edge
// between two streams shows up as a wrong (not bit-identical) result.  It says nothing about speed or occupancy.
//
// The product (battgp_amd/_lib.py) never loads the library built from this header: it is linked only into
// tests/emu/_build/libbattgp_emu.so, which tests/test_emu_kernels.py injects into the binding by hand.
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <functional>
#include <memory>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define BGP_WAVES_PER_EU(n)  // the kernel sources' register-budget attribute means nothing on the host ...
#define BGP_GLOBAL_AS         // ... and neither does the global address space
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 {
  double x, y;
} __attribute__((aligned(16)));
static inline double2 make_double2(double x, double y) {
  double2 r;
  r.x = x;
  r.y = y;
  return r;
}

// ---- error / stream / event API (synchronous, in order) ---------------------------------------------------
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct ihipStream_t {
  int id;
  void* queue;  // hipemu's queue of this stream (deferred modes)
};
typedef ihipStream_t* hipStream_t;
struct ihipEvent_t {
  double t_ms;
  void* rec_queue;   // the queue and position of the last hipEventRecord (deferred modes)
  uint64_t rec_seq;
};
typedef ihipEvent_t* hipEvent_t;
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipHostMallocDefault 0

namespace hipemu {
double now_ms();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
// stream model (hipemu.cpp): immediate execution, or queues drained by an adversarial scheduler (HIPEMU_SCHED)
void stream_create(hipStream_t s);
void stream_destroy(hipStream_t s);
void enqueue(hipStream_t s, std::function<void()> fn);
void event_record(hipEvent_t e, hipStream_t s);
void event_forget(hipEvent_t e);
void stream_wait_event(hipStream_t s, hipEvent_t e);
void stream_sync(hipStream_t s);
void event_sync(hipEvent_t e);
bool event_done(hipEvent_t e);
void device_sync();
bool deferred();
struct Ctx {
  dim3 tid, bid, bdim, gdim;
};
extern thread_local Ctx* g_ctx;  // the running fiber's coordinates
// rendezvous primitives (all live lanes of the calling lane's wavefront / all live threads of its workgroup)
uint64_t wave_xchg(uint64_t mine, int srclane, int fallback_self);
int wave_first_lane();
void wave_mfma_f64_16x16x4(double a, double b, double* d4, int neg_a);
void block_barrier();
int block_or(int pred);
}  // namespace hipemu

#define threadIdx (::hipemu::g_ctx->tid)
#define blockIdx (::hipemu::g_ctx->bid)
#define blockDim (::hipemu::g_ctx->bdim)
#define gridDim (::hipemu::g_ctx->gdim)

static inline const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "hipSuccess";
    case hipErrorInvalidValue: return "hipErrorInvalidValue";
    case hipErrorOutOfMemory: return "hipErrorOutOfMemory";
    case hipErrorNotReady: return "hipErrorNotReady";
    default: return "hipError";
  }
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() {
  ::hipemu::device_sync();
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *free_b = (size_t)48 << 30;
  *total_b = (size_t)64 << 30;
  return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t bytes) {
  void* q = nullptr;
  // a bounded "device": sizes the GPU tests use to provoke an allocation failure (or that would swamp the build
  // container) fail the same way here
  const char* lim = getenv("HIPEMU_MEM_GB");
  if (bytes > ((size_t)(lim ? atoi(lim) : 6) << 30)) return hipErrorOutOfMemory;
  const char* lim_mb = getenv("HIPEMU_MEM_MB");  // the same bound in MiB (read per call: a test can move it between two calls)
  if (lim_mb && bytes > ((size_t)atoi(lim_mb) << 20)) return hipErrorOutOfMemory;
  if (posix_memalign(&q, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
  // fresh device memory is NOT zero: 0xA5 bytes (a finite -1e-128 in fp64) by default; HIPEMU_POISON=ff makes every
  // never-written double a NaN, so that anything that lets a don't-care entry reach a result shows up at once
  static const int poison = getenv("HIPEMU_POISON") ? (int)strtol(getenv("HIPEMU_POISON"), nullptr, 16) : 0xA5;
  memset(q, poison, bytes < ((size_t)1 << 28) ? bytes : ((size_t)1 << 28));
  *p = q;
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
  ::hipemu::device_sync();  // like the real call: nothing in flight may still use the memory
  free(p);
  return hipSuccess;
}
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind kind, hipStream_t st = nullptr) {
  if (!::hipemu::deferred() || st == nullptr) {
    memmove(d, s, n);
  } else if (kind == hipMemcpyHostToDevice) {
    // pageable host memory is staged before the call returns: the source may change or go away afterwards
    auto staged = std::make_shared<std::vector<char>>(static_cast<const char*>(s), static_cast<const char*>(s) + n);
    ::hipemu::enqueue(st, [d, staged]() { memcpy(d, staged->data(), staged->size()); });
  } else {
    ::hipemu::enqueue(st, [d, s, n]() { memmove(d, s, n); });
  }
  return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                          hipStream_t st = nullptr) {
  ::hipemu::enqueue(st, [=]() {
    for (size_t r = 0; r < height; ++r) memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  });
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) {
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr) {
  ::hipemu::enqueue(st, [d, v, n]() { memset(d, v, n); });
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new ihipStream_t{0, nullptr};
  ::hipemu::stream_create(*s);
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t s) {
  ::hipemu::stream_destroy(s);
  delete s;
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t s) {
  ::hipemu::stream_sync(s);
  return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  ::hipemu::stream_wait_event(s, e);
  return hipSuccess;
}
static inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new ihipEvent_t{0.0, nullptr, 0};
  return hipSuccess;
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) {
  ::hipemu::event_forget(e);
  delete e;
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) {
  ::hipemu::event_record(e, s);
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) {
  ::hipemu::event_sync(e);
  return hipSuccess;
}
static inline hipError_t hipEventQuery(hipEvent_t e) { return ::hipemu::event_done(e) ? hipSuccess : hipErrorNotReady; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!::hipemu::event_done(a) || !::hipemu::event_done(b)) return hipErrorNotReady;  // like the real call
  *ms = (float)(b->t_ms - a->t_ms);
  return hipSuccess;
}

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                                        \
  do {                                                                                                   \
    const dim3 hipemu_g_ = dim3(grid), hipemu_b_ = dim3(block);                                          \
    ::hipemu::enqueue((stream), [=]() { ::hipemu::launch(hipemu_g_, hipemu_b_, [=]() { kern(__VA_ARGS__); }); }); \
  } while (0)

// ---- device intrinsics --------------------------------------------------------------------------------------
#define __syncthreads() ::hipemu::block_barrier()
#define __syncthreads_or(p) ::hipemu::block_or(p)
#define __threadfence_block() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
// v_rsq_f64 is a ~2^-26 seed on the GPU; every caller refines it, so the exact value is a valid stand-in
#define __builtin_amdgcn_rsq(x) (1.0 / __builtin_sqrt((double)(x)))
#define __builtin_amdgcn_readlane(v, l) ((int)::hipemu::wave_xchg((uint64_t)(uint32_t)(v), (l), 0))
#define __builtin_amdgcn_readfirstlane(v) ((int)::hipemu::wave_xchg((uint64_t)(uint32_t)(v), ::hipemu::wave_first_lane(), 0))

static inline int __double2loint(double v) {
  uint64_t b;
  memcpy(&b, &v, 8);
  return (int)(uint32_t)b;
}
static inline int __double2hiint(double v) {
  uint64_t b;
  memcpy(&b, &v, 8);
  return (int)(uint32_t)(b >> 32);
}
static inline double __hiloint2double(int hi, int lo) {
  const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double v;
  memcpy(&v, &b, 8);
  return v;
}
static inline double __shfl_down(double v, unsigned off) {
  uint64_t b;
  memcpy(&b, &v, 8);
  const int lane = (int)((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63);
  const int src = lane + (int)off;
  b = ::hipemu::wave_xchg(b, src < 64 ? src : lane, 1);
  memcpy(&v, &b, 8);
  return v;
}
// D[m][n] += sum_k A[m][k] B[k][n];  lane l holds a = A[l & 15][l >> 4], b = B[l >> 4][l & 15] and, in register r,
// D[(l >> 4) + 4 r][l & 15]  (the gfx950 f64 16x16x4 layout the kernels are written against); `neg` negates A
template <class V4>
static inline V4 hipemu_mfma_f64(double a, double b, V4 c, int neg) {
  double d[4] = {c[0], c[1], c[2], c[3]};
  ::hipemu::wave_mfma_f64_16x16x4(a, b, d, neg & 1);
  V4 r;
  r[0] = d[0];
  r[1] = d[1];
  r[2] = d[2];
  r[3] = d[3];
  return r;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, cbsz, abid, blgp) hipemu_mfma_f64((a), (b), (c), (blgp))

static inline int atomicCAS(int* p, int expected, int desired) {
  __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expected;
}
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline void unsafeAtomicAdd(double* p, double v) {
  uint64_t old_b, new_b;
  memcpy(&old_b, p, 8);
  for (;;) {
    double o;
    memcpy(&o, &old_b, 8);
    const double n = o + v;
    memcpy(&new_b, &n, 8);
    if (__atomic_compare_exchange_n(reinterpret_cast<uint64_t*>(p), &old_b, new_b, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return;
  }
}
static inline uint64_t wall_clock64() {
  return (uint64_t)(::hipemu::now_ms() * 1e5);  // 100 MHz like the GPU's constant-rate counter
}
static inline uint64_t clock64() { return __builtin_ia32_rdtsc(); }
