// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel and
// C-ABI sources of steganosaurus_amd/csrc be compiled with g++ and executed on
// the CPU, one workgroup at a time, with one ucontext fiber per HIP thread
// (__syncthreads() = yield to the next fiber).  It exists so that index math
// and barrier placement can be debugged (and run under ASan/UBSan) in a
// container without a GPU.  It is NOT a fallback: the product library
// (steganosaurus_amd/libturtlefft_hip.so) is built by hipcc for gfx950 only and
// the Python package refuses to load anything else; the emulated build is a
// separate file (tests/emu/libtfft_emu.so) that only tests/test_emulated.py opens.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define TFFT_WAVES_PER_EU(n)

struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx { unsigned x, y, z; };
extern emu_idx threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };

void emu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void emu_syncthreads();
void emu_wave_sync();
#define __syncthreads() emu_syncthreads()
// wave-level sync points of the kernels: a barrier among the 64 fibers of the wave
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// wave collectives (call sites must be wave uniform, as on the hardware path they are used in)
unsigned long long emu_ballot(int pred);
unsigned emu_lane();
unsigned emu_readfirstlane(unsigned v);
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((unsigned)(v))
#define __ballot(p) emu_ballot((p) ? 1 : 0)
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline unsigned emu_mbcnt_lo(unsigned m, unsigned init) { const unsigned l = emu_lane(); return init + (unsigned)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
static inline unsigned emu_mbcnt_hi(unsigned m, unsigned init) { const unsigned l = emu_lane(); return init + (l < 32 ? 0u : (unsigned)__builtin_popcount(m & ((1u << (l - 32)) - 1u))); }
static inline unsigned emu_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3))); }
#define __builtin_amdgcn_alignbyte(hi, lo, sh) emu_alignbyte((hi), (lo), (sh))
#define __builtin_amdgcn_mbcnt_lo(m, i) emu_mbcnt_lo((m), (i))
#define __builtin_amdgcn_mbcnt_hi(m, i) emu_mbcnt_hi((m), (i))
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); if (x == floor(x)) *s = 0.0; if (x - floor(x) == 0.5) *c = 0.0; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = v; return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p |= v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }

static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "gfx950:emulated-on-cpu"); p->multiProcessorCount = 4; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
