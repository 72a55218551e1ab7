// poseidon_consts.cpp — see poseidon_consts.hpp.
#include "poseidon_consts.hpp"
#include <array>
#include <mutex>

namespace zkgl {
namespace {

class ChaCha8 {  // rand_chacha::ChaCha8Rng word stream (64-bit block counter, stream id 0)
  public:
    explicit ChaCha8(uint64_t seed) {
        // rand_core::SeedableRng::seed_from_u64 — PCG32 expansion into the 256-bit key
        uint64_t st = seed;
        for (auto& k : key_) {
            st = st * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
            uint32_t rot = (uint32_t)(st >> 59);
            k = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
    }
    uint32_t next_u32() {
        if (pos_ == 16) refill();
        return buf_[pos_++];
    }
    uint64_t next_u64() {
        uint64_t lo = next_u32();
        uint64_t hi = next_u32();
        return lo | (hi << 32);
    }

  private:
    static uint32_t rl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static void quarter(std::array<uint32_t, 16>& s, int a, int b, int c, int d) {
        s[a] += s[b]; s[d] = rl(s[d] ^ s[a], 16);
        s[c] += s[d]; s[b] = rl(s[b] ^ s[c], 12);
        s[a] += s[b]; s[d] = rl(s[d] ^ s[a], 8);
        s[c] += s[d]; s[b] = rl(s[b] ^ s[c], 7);
    }
    void refill() {
        std::array<uint32_t, 16> in = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; ++i) in[4 + i] = key_[i];
        in[12] = (uint32_t)ctr_; in[13] = (uint32_t)(ctr_ >> 32); in[14] = 0; in[15] = 0;
        std::array<uint32_t, 16> s = in;
        for (int dr = 0; dr < 4; ++dr) {  // 8 rounds = 4 double rounds
            quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15);
            quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) buf_[i] = s[i] + in[i];
        ++ctr_;
        pos_ = 0;
    }
    std::array<uint32_t, 8> key_{};
    std::array<uint32_t, 16> buf_{};
    uint64_t ctr_ = 0;
    int pos_ = 16;
};

std::array<uint64_t, 360> g_rc;
std::once_flag g_once;

}  // namespace

const uint64_t* poseidon_round_constants() {
    std::call_once(g_once, [] {
        constexpr uint64_t P = 0xFFFFFFFF00000001ull;
        ChaCha8 rng(0);
        int n = 0;
        while (n < 360) {
            unsigned __int128 m = (unsigned __int128)rng.next_u64() * P;
            if ((uint64_t)m <= P - 1) g_rc[n++] = (uint64_t)(m >> 64);
        }
    });
    return g_rc.data();
}

}  // namespace zkgl
