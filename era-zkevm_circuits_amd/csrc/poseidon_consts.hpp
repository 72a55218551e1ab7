// poseidon_consts.hpp — host-side derivation of the 360 Poseidon-Goldilocks round constants
// (12 x 30) that boojum's Poseidon2 shares with its Poseidon ([EXT]).
#pragma once
#include <cstdint>
namespace zkgl {
// derived once, thread-safe; procedure: ChaCha8Rng::seed_from_u64(0) then one
// `gen_range(0..p)` (rand 0.8 widening-multiply sampling) per constant.
const uint64_t* poseidon_round_constants();
}
