// tests/emu/dev/hip/hip_runtime.h — TEST INFRASTRUCTURE (tests/emu/README.md, "the emulated device").
//
// A stand-in for <hip/hip_runtime.h> under which the product's ONE device translation unit (zkgl_device.hip) and the host files that talk
// to the HIP runtime compile as plain host C++.  The kernels run on `emu_rt.cpp`'s scheduler: a workgroup is a set of fibers (one per
// work-item, 64 consecutive ones = a wavefront), run on one OS thread, one workgroup after another.
//   * __syncthreads        = every live work-item of the workgroup arrives before any leaves
//   * wavefront operations = the live lanes of the wavefront that arrive at the SAME call site exchange values there:
//                            ballot, readfirstlane, readlane, __shfl / __shfl_up / __shfl_xor, update_dpp, wave_barrier
//   * __shared__           = static storage of the (one at a time) workgroup;  extern __shared__ = the launch's dynamic LDS buffer
//   * buffer loads/stores  = base + voffset + soffset;  atomics = plain read-modify-write (one OS thread)
//   * hipMalloc & co       = host memory (poisoned, not zeroed);  streams and events = immediate
// The product never sees this file: it is on the include path of tests/emu/dev/build.sh only.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <cmath>

#define ZKGL_EMULATED_DEVICE 1

// ---- language ----
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#ifdef EMU_TSAN
#define __shared__ __attribute__((section("emu_lds"))) static   // race-detector build: one process = one grid at a time; the scheduler knows where LDS is
#else
#define __shared__ static thread_local   // fibers of a workgroup share the OS thread
#endif
#define HIP_SYMBOL(x) (&(x))
#define __noinline__ __attribute__((noinline))
#define address_space(n)            /* __attribute__((address_space(4))) -> an empty attribute */
#define amdgpu_waves_per_eu(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- vector types ----
struct ulonglong2 { unsigned long long x, y; };
struct ulonglong4 { unsigned long long x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace emu {
enum WaveOp : int { OP_BALLOT, OP_READFIRST, OP_READLANE, OP_SHFL, OP_SHFL_UP, OP_SHFL_XOR, OP_DPP, OP_WAVE_BARRIER };
struct Fiber;
extern thread_local Fiber* cur;
extern thread_local dim3 t_idx, b_idx, b_dim, g_dim;
extern thread_local void* dyn_lds;
extern thread_local unsigned lane_in_wave;
// the calling lane blocks until the live lanes of its wavefront meet at this call site; returns its result
// (noduplicate + convergent: the host compiler may neither clone a call into the two arms of a branch nor make it depend on a new condition —
//  a call site is the identity of the operation, as it is for the wavefront)
uint64_t wave_op(int op, uint64_t in, uint64_t aux, uint64_t aux2 = 0) __attribute__((noinline, noduplicate, convergent));
void sync_threads() __attribute__((noinline, noduplicate, convergent));
struct Body { virtual void run() const = 0; };
template <class F> struct BodyOf : Body { const F& f; explicit BodyOf(const F& f_) : f(f_) {} void run() const override { f(); } };
void launch_body(dim3 grid, dim3 block, size_t lds_bytes, const Body& b);
template <class F> inline void launch(dim3 grid, dim3 block, size_t lds_bytes, void* /*stream*/, const F& f) { launch_body(grid, block, lds_bytes, BodyOf<F>(f)); }
template <class F> inline void launch(dim3 grid, dim3 block, const F& f) { launch_body(grid, block, 0, BodyOf<F>(f)); }
void* alloc(size_t n);
void duplicate_lane(bool dup);
void release(void* p);
}  // namespace emu

#ifdef EMU_TSAN
#define EMU_DUPLICATE_LANE(dup) emu::duplicate_lane(dup)
#else
#define EMU_DUPLICATE_LANE(dup) ((void)0)
#endif
#define threadIdx (emu::t_idx)
#define blockIdx (emu::b_idx)
#define blockDim (emu::b_dim)
#define gridDim (emu::g_dim)
#define warpSize 64

// ---- wavefront / workgroup intrinsics ----
static inline void __syncthreads() { emu::sync_threads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)emu::wave_op(emu::OP_WAVE_BARRIER, 0, 0))
#define __builtin_amdgcn_ballot_w64(p) ((uint64_t)emu::wave_op(emu::OP_BALLOT, (p) ? 1 : 0, 0))
#define __ballot(p) ((unsigned long long)emu::wave_op(emu::OP_BALLOT, (p) ? 1 : 0, 0))
#define __builtin_amdgcn_readfirstlane(x) ((int)(uint32_t)emu::wave_op(emu::OP_READFIRST, (uint32_t)(x), 0))
#define __builtin_amdgcn_readlane(x, l) ((int)(uint32_t)emu::wave_op(emu::OP_READLANE, (uint32_t)(x), (uint32_t)(l)))
#define __builtin_amdgcn_mbcnt_lo(m, a) ((unsigned)(a) + (unsigned)__builtin_popcount((unsigned)(m) & (unsigned)((emu::lane_in_wave >= 32 ? 0xffffffffull : ((1ull << emu::lane_in_wave) - 1)))))
#define __builtin_amdgcn_mbcnt_hi(m, a) ((unsigned)(a) + (unsigned)__builtin_popcount((unsigned)(m) & (unsigned)(emu::lane_in_wave > 32 ? ((1ull << (emu::lane_in_wave - 32)) - 1) : 0)))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
    ((int)(uint32_t)emu::wave_op(emu::OP_DPP, (uint32_t)(src), (uint64_t)(ctrl) | ((uint64_t)(row_mask) << 16) | ((uint64_t)(bank_mask) << 20) | ((uint64_t)((bound_ctrl) ? 1 : 0) << 24), (uint32_t)(old)))
#define __builtin_amdgcn_s_memrealtime() ((uint64_t)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10))
#define __builtin_amdgcn_s_memtime() ((uint64_t)__builtin_readcyclecounter())
template <class T> static __forceinline__ T __shfl(T v, int src, int /*width*/ = 64) { return (T)emu::wave_op(emu::OP_SHFL, (uint64_t)v, (uint32_t)src); }
template <class T> static __forceinline__ T __shfl_up(T v, unsigned d, int /*width*/ = 64) { return (T)emu::wave_op(emu::OP_SHFL_UP, (uint64_t)v, d); }
template <class T> static __forceinline__ T __shfl_xor(T v, int m, int /*width*/ = 64) { return (T)emu::wave_op(emu::OP_SHFL_XOR, (uint64_t)v, (uint32_t)m); }
static inline unsigned __brev(unsigned x) { return __builtin_bitreverse32(x); }
static inline unsigned long long __brevll(unsigned long long x) { return __builtin_bitreverse64(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
using std::min;
using std::max;
template <class A, class B> static inline auto min(A a, B b) -> typename std::common_type<A, B>::type { typedef typename std::common_type<A, B>::type T; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B> static inline auto max(A a, B b) -> typename std::common_type<A, B>::type { typedef typename std::common_type<A, B>::type T; return (T)a > (T)b ? (T)a : (T)b; }

// ---- atomics (relaxed builtins: one OS thread runs the grid, but the race-detector build must see them as atomics) ----
template <class T, class U> static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicSub(T* p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicCAS(T* p, U cmp, U v) { T e = (T)cmp; __atomic_compare_exchange_n(p, &e, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return e; }
template <class T, class U> static inline T atomicMin(T* p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v < o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T, class U> static inline T atomicMax(T* p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while ((T)v > o && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}

// ---- buffer addressing: V# = a base address (the kernels use no bounds) ----
struct emu_rsrc { char* base; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) (emu_rsrc{(char*)(p)})
typedef uint32_t emu_u32x2 __attribute__((ext_vector_type(2)));
// how many lane-level buffer operations of each kind the kernels issued (8-byte loads, 8-byte stores, 1-byte loads, 1-byte stores): what a store layout
// costs in issued bytes, counted where the kernel issues them (zk_emu_buffer_ops; tests/test_zz_round6_narrow_store.py compares it with zk_stats)
namespace emu { extern unsigned long long buffer_ops[4]; }
#ifdef EMU_TSAN   // (the race-detector build does not count: a plain counter shared by the fibers would itself be reported)
#define EMU_COUNT(i) ((void)0)
#else
#define EMU_COUNT(i) (++emu::buffer_ops[i])
#endif
static inline emu_u32x2 emu_buffer_load_b64(emu_rsrc r, uint32_t voff, uint32_t soff) { EMU_COUNT(0); emu_u32x2 v; memcpy(&v, r.base + (size_t)voff + (size_t)soff, 8); return v; }
#ifdef EMU_TSAN
// Race-detector build: a store of the value the cell already holds is not a write.  The witness kernels clamp the lanes beyond the batch to the
// last valid lane (they redo its work and store the same values to the same cells: deterministic, and a write-write race by the letter);
// a store of a DIFFERENT value to a cell another work-item wrote is reported as before.
__attribute__((no_sanitize("thread"), noinline)) static bool emu_same_value(const char* p, emu_u32x2 v) { emu_u32x2 o; __builtin_memcpy(&o, p, 8); return o.x == v.x && o.y == v.y; }
static inline void emu_buffer_store_b64(emu_u32x2 v, emu_rsrc r, uint32_t voff, uint32_t soff) {
    EMU_COUNT(1);
    char* p = r.base + (size_t)voff + (size_t)soff;
    if (!emu_same_value(p, v)) memcpy(p, &v, 8);
}
#else
static inline void emu_buffer_store_b64(emu_u32x2 v, emu_rsrc r, uint32_t voff, uint32_t soff) { EMU_COUNT(1); memcpy(r.base + (size_t)voff + (size_t)soff, &v, 8); }
#endif
// one-byte forms (narrow store: buffer_load_ubyte / buffer_store_byte)
static inline uint8_t emu_buffer_load_b8(emu_rsrc r, uint32_t voff, uint32_t soff) { EMU_COUNT(2); return (uint8_t)r.base[(size_t)voff + (size_t)soff]; }
#ifdef EMU_TSAN
__attribute__((no_sanitize("thread"), noinline)) static bool emu_same_byte(const char* p, uint8_t v) { return (uint8_t)*p == v; }
static inline void emu_buffer_store_b8(uint8_t v, emu_rsrc r, uint32_t voff, uint32_t soff) { EMU_COUNT(3); char* p = r.base + (size_t)voff + (size_t)soff; if (!emu_same_byte(p, v)) *p = (char)v; }
#else
static inline void emu_buffer_store_b8(uint8_t v, emu_rsrc r, uint32_t voff, uint32_t soff) { EMU_COUNT(3); r.base[(size_t)voff + (size_t)soff] = (char)v; }
#endif
#define __builtin_amdgcn_raw_buffer_load_b8(rsrc, voff, soff, aux) emu_buffer_load_b8((rsrc), (uint32_t)(voff), (uint32_t)(soff))
#define __builtin_amdgcn_raw_buffer_store_b8(v, rsrc, voff, soff, aux) emu_buffer_store_b8((uint8_t)(v), (rsrc), (uint32_t)(voff), (uint32_t)(soff))
#define __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, aux) emu_buffer_load_b64((rsrc), (uint32_t)(voff), (uint32_t)(soff))
#define __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff, soff, aux) emu_buffer_store_b64((v), (rsrc), (uint32_t)(voff), (uint32_t)(soff))

// ---- runtime API ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct emu_stream* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 1 };
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory (emulated device)" : "invalid value (emulated device)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// one emulated device; EMU_DEVICES=n makes the process see n of them (all the same host memory) so that the one-process-per-GPU launch of bench.py --gpus n
// binds rank r to "device" r and takes its RCCL path over the stand-in collective (tests/emu/dev/rccl/rccl.h)
static inline int emu_device_count() { const char* e = getenv("EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n >= 1 && n <= 64 ? n : 1; }
inline int& emu_current_device() { static int d = 0; return d; }   // (inline, not static: ONE current device for every translation unit of the library)
static inline hipError_t hipGetDeviceCount(int* n) { *n = emu_device_count(); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = emu_current_device(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_device_count()) return hipErrorInvalidValue; emu_current_device() = d; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = emu::alloc(n); return *p || !n ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
template <class T> static inline hipError_t hipMallocAsync(T** p, size_t n, hipStream_t s) { return hipMallocAsync((void**)p, n, s); }
static inline hipError_t hipFree(void* p) { emu::release(p); return hipSuccess; }
static inline hipError_t hipFreeAsync(void* p, hipStream_t) { emu::release(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(const void* sym, const void* src, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { memcpy((char*)sym + off, src, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)emu::alloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { emu::release(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{std::chrono::steady_clock::now()}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
