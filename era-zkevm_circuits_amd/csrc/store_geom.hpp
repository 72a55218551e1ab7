// store_geom.hpp — the lane tiling of a value store, shared by the host (cs.cpp) and every kernel that addresses one.
//
//   store[((lane >> T) * n_slots + slot) << T | (lane & (2^T - 1))]          T = tile_log2
//
// T = 6 (the default everywhere): a wavefront owns one contiguous tile of n_slots * 512 B.
// T = 7..12 (ZKGL_STORE_TILE_LOG2, loop scopes): 2^(T-6) wavefronts share a tile and a value of the tile is 2^(T+3) B contiguous.  The
// bare store pattern of the loop kernel streams 5-8 % faster with T = 12 and stops depending on where the pages of the allocation sit
// (tools/layout_probe.hip, profiles/r3_layout_probe.jsonl); the real kernel does not gain (profiles/r3_loop_probe.md §1: 40.4-40.8 ms
// against 39.3-40.3 ms), so the wide tiling is an A/B switch, covered by tests/test_gpu_store_tiling.py, not the default.
//
// A store travels through the launch interface as (pointer, geometry word): the slot count with T in the top byte.  A bare slot count
// (top byte 0) means T = 6, so interfaces that only ever see 64-lane-tiled memory pass their counts unchanged.
//
// NARROW store (loop scopes; cs.cpp build_narrow_layout; a batch takes it when ZKGL_NARROW_STORE=1 is set at zk_cs_set_batch).  CS::bound_values proves from the constraints alone that 29 % of main_vm's
// values are bytes / booleans in every satisfying witness; held in 8-byte slots they are 26 % of the bytes the witness kernel writes (and of
// what it re-fetches).  In a narrow store a value has a CLASS: 8 bytes per lane, or 1 byte per lane; a tile is a sequence of UNITS (one byte
// per lane of the tile = 2^T bytes) and a value is named by its ADDRESS WORD
//     aw = first unit of the value | class << 28          (AW_BYTE: one unit, one byte per lane;  otherwise eight units, 8 B per lane)
// so byte address of (aw, lane) = tile base + ((aw & AW_MASK) << T) + (lane & (2^T - 1)) * width.  The geometry word of a narrow store carries
// NARROW and, as its "slot count", the tile size in 8-byte slots (units / 8, rounded up): allocation and tile addressing are those of an
// ordinary store of that many slots.  Only the kernels of the fused step read a narrow store (witness loop kernel, k_check_prog, links,
// stream links, ZK_OP_LOOP_LAST); every other reader sees the ordinary store k_widen_store expands it into on demand (CS::ensure_p2_filled).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zkgeom {

constexpr uint32_t TILE_SHIFT = 56;
constexpr uint32_t WAVE_TILE_LOG2 = 6, WIDE_TILE_LOG2 = 12;
constexpr uint64_t NARROW = 1ull << 55;                       // geometry word: the store is a narrow store, its words are address words
constexpr uint32_t AW_BYTE = 1u << 28, AW_MASK = AW_BYTE - 1;  // address word: class bit, first unit

constexpr uint64_t pack(uint64_t n_slots, uint32_t tile_log2) { return n_slots | ((uint64_t)tile_log2 << TILE_SHIFT); }
constexpr uint64_t slots(uint64_t geom) { return geom & (NARROW - 1); }
constexpr bool narrow(uint64_t geom) { return (geom & NARROW) != 0; }
constexpr uint32_t tile_log2(uint64_t geom) { return (geom >> TILE_SHIFT) ? (uint32_t)(geom >> TILE_SHIFT) : WAVE_TILE_LOG2; }
// element offset of (slot, lane)
constexpr size_t offset(uint64_t geom, uint64_t slot, uint64_t lane) {
    const uint32_t t = tile_log2(geom);
    return (size_t)((((lane >> t) * slots(geom) + slot) << t) + (lane & ((1ull << t) - 1)));
}
// byte offset of (address word, lane) in a narrow store
constexpr size_t narrow_byte_offset(uint64_t geom, uint32_t aw, uint64_t lane) {
    const uint32_t t = tile_log2(geom);
    return (size_t)((((lane >> t) * slots(geom)) << (t + 3)) + ((uint64_t)(aw & AW_MASK) << t) + (lane & ((1ull << t) - 1)) * ((aw & AW_BYTE) ? 1 : 8));
}
// lanes a store of `lanes` lanes is allocated for (whole tiles)
constexpr uint64_t padded_lanes(uint64_t geom, uint64_t lanes) {
    const uint32_t t = tile_log2(geom);
    return ((lanes + (1ull << t) - 1) >> t) << t;
}

}  // namespace zkgeom
