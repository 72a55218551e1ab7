// tests/emu/emu_shim.hpp — what the LANE HARNESS puts under the product's device source so that it compiles as host C++ (tests/emu/README.md).
// TEST INFRASTRUCTURE: nothing in era-zkevm_circuits_amd/ includes or links this; the C ABI has no way to reach it.
//
// A lane runs as its own one-lane wavefront: every cross-lane operation of the witness interpreter is a per-wavefront OPTIMISATION that keeps
// each lane's results (ballots decide uniform shortcuts, readfirstlane moves wave-uniform program words to the scalar unit, the multiplicity
// aggregation merges equal addresses) — with one lane, ballot(p) = p at bit 0, readfirstlane / readlane(x) = x, the lane index is 0.  The strand
// form is a workgroup of one-lane wavefronts on real threads: __syncthreads is a barrier between them.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static                       /* LDS of the one workgroup that runs at a time: plain lanes run in sequence, the strands of a tile are its threads */
#define __constant__
#define address_space(n)                          /* __attribute__((address_space(4))) -> __attribute__(()) */
#define amdgpu_waves_per_eu(...)                  /* -> an empty attribute */

namespace emu {
struct Dim3 { unsigned x, y, z; };
extern thread_local Dim3 tid, bid, bdim, gdim;
void block_barrier();                             // strand form: all strands of the tile (set up by the driver)
struct Rsrc { char* base; };
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
inline Rsrc make_rsrc(const void* p) { return Rsrc{(char*)const_cast<void*>(p)}; }
inline u32x2_t load_b64(Rsrc r, uint32_t voff, uint32_t soff) { u32x2_t v; std::memcpy(&v, r.base + (size_t)voff + soff, 8); return v; }
inline void store_b64(u32x2_t v, Rsrc r, uint32_t voff, uint32_t soff) { std::memcpy(r.base + (size_t)voff + soff, &v, 8); }
inline uint8_t load_b8(Rsrc r, uint32_t voff, uint32_t soff) { return (uint8_t)r.base[(size_t)voff + soff]; }                 // narrow store: one-byte slots
inline void store_b8(uint8_t v, Rsrc r, uint32_t voff, uint32_t soff) { r.base[(size_t)voff + soff] = (char)v; }
}  // namespace emu
#define threadIdx emu::tid
#define blockIdx emu::bid
#define blockDim emu::bdim
#define gridDim emu::gdim

typedef emu::Rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) emu::make_rsrc(p)
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) emu::load_b64(r, voff, soff)
#define __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, aux) emu::store_b64(v, r, voff, soff)
#define __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, aux) emu::load_b8(r, voff, soff)
#define __builtin_amdgcn_raw_buffer_store_b8(v, r, voff, soff, aux) emu::store_b8((uint8_t)(v), r, voff, soff)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_ballot_w64(p) ((uint64_t)((p) ? 1 : 0))
#define __ballot(p) ((uint64_t)((p) ? 1 : 0))
#define __builtin_amdgcn_mbcnt_lo(m, v) (v)
#define __builtin_amdgcn_mbcnt_hi(m, v) (v)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __syncthreads() emu::block_barrier()
#define __threadfence() std::atomic_thread_fence(std::memory_order_seq_cst)
#define __clz(x) __builtin_clz(x)
#define __popcll(x) __builtin_popcountll(x)
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
