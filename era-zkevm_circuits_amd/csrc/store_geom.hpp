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
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zkgeom {

constexpr uint32_t TILE_SHIFT = 56;
constexpr uint32_t WAVE_TILE_LOG2 = 6, WIDE_TILE_LOG2 = 12;

constexpr uint64_t pack(uint64_t n_slots, uint32_t tile_log2) { return n_slots | ((uint64_t)tile_log2 << TILE_SHIFT); }
constexpr uint64_t slots(uint64_t geom) { return geom & ((1ull << TILE_SHIFT) - 1); }
constexpr uint32_t tile_log2(uint64_t geom) { return (geom >> TILE_SHIFT) ? (uint32_t)(geom >> TILE_SHIFT) : WAVE_TILE_LOG2; }
// element offset of (slot, lane)
constexpr size_t offset(uint64_t geom, uint64_t slot, uint64_t lane) {
    const uint32_t t = tile_log2(geom);
    return (size_t)((((lane >> t) * slots(geom) + slot) << t) + (lane & ((1ull << t) - 1)));
}
// lanes a store of `lanes` lanes is allocated for (whole tiles)
constexpr uint64_t padded_lanes(uint64_t geom, uint64_t lanes) {
    const uint32_t t = tile_log2(geom);
    return ((lanes + (1ull << t) - 1) >> t) << t;
}

}  // namespace zkgeom
