// TEST INFRASTRUCTURE ONLY -- never part of the product path.
//
// A tiny SIMT emulator that lets the *unmodified* HIP device sources under
// proxsuite_amd/csrc/ be compiled with g++ and executed on the CPU, so that kernel
// logic (indexing, barriers, reductions, control flow) can be validated against
// the oracle in the authoring container, which has no GPU.  One fiber per
// thread of a workgroup; fibers switch only at collective points
// (__syncthreads, __shfl*, __ballot), so a missing barrier shows up as a wrong
// result here instead of "working by lockstep luck" on hardware, and a collective
// reached by only part of a wave/block is reported as an error (deadlock).
//
// The product library (libproxqp_hip.so) is built by hipcc from the same sources
// and is the only thing proxsuite_amd ever loads; this header is injected with
// `g++ -include tests/emu/hip_emu.hpp` by tests/emu/build.py only.
#ifndef PQP_HIP_EMU_HPP
#define PQP_HIP_EMU_HPP

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define PQP_EMULATED 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local

struct dim3
{
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1)
    : x(x_)
    , y(y_)
    , z(z_)
  {
  }
};
struct uint3_emu
{
  unsigned x, y, z;
};

namespace hipemu {
struct Fiber;
struct Block
{
  unsigned nthreads = 0;
  unsigned bar_count = 0;
  unsigned long bar_gen = 0;
  unsigned alive = 0;
  std::vector<Fiber*> fibers;
  unsigned cur = 0;
  char* dyn_smem = nullptr;
  // per-wave collective scratch
  struct Wave
  {
    unsigned count = 0;
    unsigned long gen = 0;
    unsigned alive = 0;
    std::uint64_t slot64[64];
    std::uint64_t ballot = 0;
    unsigned long rd_count = 0;
  };
  std::vector<Wave> waves;
};
extern thread_local Block* g_block;
extern thread_local uint3_emu g_threadIdx, g_blockIdx;
extern thread_local dim3 g_blockDim, g_gridDim;

void yield_to_next();
void block_barrier();
void wave_barrier();
[[noreturn]] void fatal(const char* msg);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
std::uint64_t wave_exchange(std::uint64_t v, int src_lane, bool valid_src);
std::uint64_t wave_ballot(bool pred);
} // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
static const int warpSize = 64;

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::g_block->dyn_smem);

inline void
__syncthreads()
{
  hipemu::block_barrier();
}

template<typename T>
inline T
__shfl(T v, int src, int width = 64)
{
  static_assert(sizeof(T) <= 8, "shfl payload");
  std::uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  int lane = int(threadIdx.x & 63u);
  int base = lane & ~(width - 1);
  int s = base + (src & (width - 1));
  raw = hipemu::wave_exchange(raw, s, true);
  T out;
  std::memcpy(&out, &raw, sizeof(T));
  return out;
}
template<typename T>
inline T
__shfl_xor(T v, int mask, int width = 64)
{
  int lane = int(threadIdx.x & 63u);
  return __shfl(v, (lane ^ mask), width);
}
template<typename T>
inline T
__shfl_down(T v, unsigned delta, int width = 64)
{
  int lane = int(threadIdx.x & 63u);
  int pos = lane & (width - 1);
  int src = (pos + int(delta) < width) ? lane + int(delta) : lane;
  return __shfl(v, src, 64);
}
template<typename T>
inline T
__shfl_up(T v, unsigned delta, int width = 64)
{
  int lane = int(threadIdx.x & 63u);
  int pos = lane & (width - 1);
  int src = (pos - int(delta) >= 0) ? lane - int(delta) : lane;
  return __shfl(v, src, 64);
}
inline unsigned long long
__ballot(int pred)
{
  return hipemu::wave_ballot(pred != 0);
}
inline int
__popcll(unsigned long long v)
{
  return __builtin_popcountll(v);
}
inline int
__ffsll(unsigned long long v)
{
  return __builtin_ffsll((long long)v);
}
inline long long
clock64()
{
  static thread_local long long c = 0;
  return ++c;
}
inline long long
wall_clock64()
{
  // a real clock (nanoseconds), so that the Info timings of the emulated kernels are ordered and non-zero
  return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(
           std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double
__fma_rn(double a, double b, double c)
{
  return std::fma(a, b, c);
}
inline int
atomicAdd(int* p, int v)
{
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned
atomicAdd(unsigned* p, unsigned v)
{
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned long long
atomicAdd(unsigned long long* p, unsigned long long v)
{
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

// ---- minimal HIP runtime surface used by the host side of the C-ABI ----------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum
{
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorNoDevice = 100,
  hipErrorPeerAccessAlreadyEnabled = 704
};
enum hipMemcpyKind
{
  hipMemcpyHostToHost = 0,
  hipMemcpyHostToDevice = 1,
  hipMemcpyDeviceToHost = 2,
  hipMemcpyDeviceToDevice = 3,
  hipMemcpyDefault = 4
};
struct hipDeviceProp_t
{
  char name[256];
  int multiProcessorCount;
  size_t sharedMemPerBlock;
  size_t totalGlobalMem;
  char gcnArchName[256];
};
struct hipemuEvent
{
  double t;
};
inline hipError_t
hipMalloc(void** p, size_t n)
{
  *p = std::calloc(n ? n : 1, 1);
  return *p ? hipSuccess : hipErrorInvalidValue;
}
template<typename T>
inline hipError_t
hipMalloc(T** p, size_t n)
{
  return hipMalloc(reinterpret_cast<void**>(p), n);
}
inline hipError_t
hipFree(void* p)
{
  std::free(p);
  return hipSuccess;
}
inline hipError_t
hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t
hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t
hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                 hipMemcpyKind, hipStream_t = nullptr)
{
  for (size_t r = 0; r < height; ++r)
    std::memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return hipSuccess;
}
inline hipError_t
hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k)
{
  return hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, k);
}
inline hipError_t
hipMemset(void* d, int v, size_t n)
{
  std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t
hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr)
{
  std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t
hipDeviceSynchronize()
{
  return hipSuccess;
}
inline hipError_t
hipStreamSynchronize(hipStream_t)
{
  return hipSuccess;
}
inline hipError_t
hipStreamCreate(hipStream_t* s)
{
  *s = nullptr;
  return hipSuccess;
}
inline hipError_t
hipStreamDestroy(hipStream_t)
{
  return hipSuccess;
}
enum
{
  hipStreamNonBlocking = 1,
  hipHostMallocMapped = 2
};
inline hipError_t
hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
  *s = nullptr;
  return hipSuccess;
}
// pinned, device-mapped host memory: the emulated device IS the host
inline hipError_t
hipHostMalloc(void** p, size_t n, unsigned = 0)
{
  *p = std::calloc(n ? n : 1, 1);
  return *p ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t
hipHostGetDevicePointer(void** dp, void* hp, unsigned)
{
  *dp = hp;
  return hipSuccess;
}
inline hipError_t
hipHostFree(void* p)
{
  std::free(p);
  return hipSuccess;
}
// (one emulated device: no peer to map)
inline hipError_t
hipDeviceCanAccessPeer(int* can, int, int)
{
  *can = 0;
  return hipSuccess;
}
inline hipError_t
hipDeviceEnablePeerAccess(int, unsigned)
{
  return hipSuccess;
}
inline hipError_t
hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr)
{
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t
hipGetLastError()
{
  return hipSuccess;
}
inline const char*
hipGetErrorString(hipError_t)
{
  return "hipemu";
}
inline hipError_t
hipGetDeviceCount(int* n)
{
  // HIPEMU_DEVICES=<k>: k emulated devices (all of them the host), for the multi-device host logic
  const char* e = std::getenv("HIPEMU_DEVICES");
  const int k = e ? std::atoi(e) : 1;
  *n = k > 0 ? k : 1;
  return hipSuccess;
}
inline hipError_t
hipSetDevice(int)
{
  return hipSuccess;
}
enum
{
  hipDeviceAttributeMultiprocessorCount = 63,
  hipDeviceAttributeWallClockRate = 10017
};
inline hipError_t
hipDeviceGetAttribute(int* v, int attr, int)
{
  if (attr == hipDeviceAttributeWallClockRate)
    *v = 1000000; // kHz: wall_clock64() of the emulator counts nanoseconds
  else
    *v = 1; // one "CU": every launch of more than three workgroups takes the throughput build of the C2 kernel
  return hipSuccess;
}
inline hipError_t
hipGetDevice(int* d)
{
  *d = 0;
  return hipSuccess;
}
inline hipError_t
hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
  std::memset(p, 0, sizeof(*p));
  std::strcpy(p->name, "hipemu (CPU fibers)");
  std::strcpy(p->gcnArchName, "emu");
  p->multiProcessorCount = 256;
  p->sharedMemPerBlock = 160 * 1024;
  p->totalGlobalMem = size_t(64) << 30;
  return hipSuccess;
}
double hipemu_now_ms();
inline hipError_t
hipEventCreate(hipEvent_t* e)
{
  *e = new hipemuEvent{ 0 };
  return hipSuccess;
}
inline hipError_t
hipEventDestroy(hipEvent_t e)
{
  delete e;
  return hipSuccess;
}
inline hipError_t
hipEventRecord(hipEvent_t e, hipStream_t = nullptr)
{
  e->t = hipemu_now_ms();
  return hipSuccess;
}
inline hipError_t
hipEventSynchronize(hipEvent_t)
{
  return hipSuccess;
}
inline hipError_t
hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
  *ms = float(b->t - a->t);
  return hipSuccess;
}
template<typename F>
inline hipError_t
hipFuncSetAttribute(F, int, int)
{
  return hipSuccess;
}
enum
{
  hipFuncAttributeMaxDynamicSharedMemorySize = 8
};

#define HIPEMU_UNPAREN(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                \
  hipemu::launch((grid), (block), (shmem), [=]() { HIPEMU_UNPAREN kernel(__VA_ARGS__); })

#endif
