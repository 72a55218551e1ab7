// circuits/eip4844.cpp — host-side mirror of /root/reference/src/eip_4844/mod.rs:107-260 (eip_4844_entry_point) and the
// limb converters :45-102, recorded against the zkgl CS.
//
//   challenge  z = last 16 bytes (big-endian) of keccak256(linear_hash ‖ versioned_hash)               (mod.rs:156-175)
//   opening    y = sum_i chunk_i * z^(n-1-i) mod r_BLS12-381 by Horner, chunk_i = 31 little-endian bytes  (mod.rs:186-204)
//   linear hash  = keccak256(all chunk bytes) must equal the supplied linear_hash_output                  (mod.rs:207-211)
//   output hash  = keccak256(versioned_hash ‖ z_be16 ‖ y_be32)                                          (mod.rs:217-237)
//   public input = commitment to the closed-form input with observable_output = {linear_hash, output_hash}.
//
// The reference unrolls 4096 Horner cycles and then runs one 934-block keccak over the same byte variables.  Here ONE
// loop scope of n_blocks iterations does both: iteration t absorbs Keccak block t and performs `cpi` =
// ceil(n_chunks / n_blocks) Horner steps (chunks cpi*t .. cpi*t+cpi-1; the steps past the last chunk are masked by flags
// derived from the carried counter t).  The blob bytes enter twice per iteration stream — as 136-byte block words and as
// 31-byte chunk words — and a STREAM link (include/zkgl_ir.h) ties the two views byte by byte.  `n_chunks` is a parameter
// (the reference fixes 4096 = ELEMENTS_PER_4844_BLOCK, src/eip_4844/input.rs:26).
//
// INPUT STREAMS
//   outer (64 words): versioned_hash[32] | linear_hash_output[32]
//   loop  (217 + 136 + 31*cpi words): keccak state[200] | opening limbs[16] | t | block bytes[136] | chunk bytes[cpi][31]
#include "keccak_gadget.hpp"
#include "nonnative.hpp"

namespace zkgl {

void keccak_configure_with(CS& cs, uint32_t lookup_repetitions);

namespace {
// BLS12-381 scalar field modulus r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 16-bit limbs LE
const std::array<uint32_t, 16> BLS_FR = {0x0001, 0x0000, 0xffff, 0xffff, 0x5bfe, 0xfffe, 0xa402, 0x53bd,
                                         0xd805, 0x09a1, 0xd808, 0x3339, 0x7d48, 0x299d, 0xa753, 0x73ed};
constexpr int RATE = 136, CHUNK = 31;

// keccak256 of a message shorter than one block, in the current scope
std::array<zk_var, 32> keccak256_one_block(G& g, K& k, const std::vector<zk_var>& msg) {
    if (msg.size() >= RATE) throw ZkError(ZK_ERR_INVALID, "keccak256_one_block: message too long");
    std::array<Lane, 25> s;
    for (auto& lane : s)
        for (auto& b : lane) b = g.zero();
    for (int j = 0; j < RATE; ++j) {
        zk_var v;
        if (j < (int)msg.size()) v = msg[j];
        else {
            uint64_t pad = (j == (int)msg.size() ? 0x01 : 0x00) | (j == RATE - 1 ? 0x80 : 0x00);
            if (!pad) continue;  // x ^ 0 = x
            v = g.constant(pad);
        }
        s[j / 8][j % 8] = k.xor8(s[j / 8][j % 8], v);  // the lookup range-checks the message byte
    }
    k.permutation(s);
    std::array<zk_var, 32> d;
    for (int j = 0; j < 32; ++j) d[j] = s[j / 8][j % 8];
    return d;
}
}  // namespace

void eip_4844_configure(CS& cs) {  // lookup parameters of the reference test: width 3 x 20 repetitions (src/eip_4844/mod.rs:495-501)
    keccak_configure_with(cs, 20);
}

void eip_4844_entry_point(CS& cs, uint32_t n_chunks) {
    if (n_chunks == 0) throw ZkError(ZK_ERR_INVALID, "eip_4844: n_chunks must be positive");
    const uint32_t n_bytes = CHUNK * n_chunks, n_blocks = n_bytes / RATE + 1, rem_bytes = n_bytes % RATE;
    const uint32_t cpi = (n_chunks + n_blocks - 1) / n_blocks;       // Horner steps per iteration
    const uint32_t full = n_chunks / cpi, rem = n_chunks % cpi;        // iterations < full: all active; iteration full: c < rem
    const uint32_t t_last = (n_chunks - 1) / cpi, c_last = (n_chunks - 1) % cpi;  // position of the last chunk (no multiplication)
    G g(cs);
    K ko(g);
    // ---- allocation (mod.rs:134-135) ----
    std::vector<zk_var> versioned_hash(32), linear_hash_output(32);
    for (auto& b : versioned_hash) b = g.next_input();
    for (auto& b : linear_hash_output) b = g.next_input();
    for (int i = 0; i < 32; i += 2) {
        g.range_check_u8_pair(versioned_hash[i], versioned_hash[i + 1]);
        g.range_check_u8_pair(linear_hash_output[i], linear_hash_output[i + 1]);
    }
    // ---- evaluation point (mod.rs:156-175, convert_truncated_keccak_digest_to_field_element :45-72) ----
    std::vector<zk_var> msg(linear_hash_output);
    msg.insert(msg.end(), versioned_hash.begin(), versioned_hash.end());
    auto challenge_hash = keccak256_one_block(g, ko, msg);
    std::array<zk_var, 16> truncated;
    for (int i = 0; i < 16; ++i) truncated[i] = challenge_hash[16 + i];
    std::array<zk_var, 8> z_outer;  // limb i = bytes (14-2i, 15-2i) big-endian
    for (int i = 0; i < 8; ++i)
        z_outer[i] = g.linear_combination({{truncated[15 - 2 * i], 1}, {truncated[14 - 2 * i], 1ull << 8}});
    zk_var outer_zero = g.zero();

    // =========================== loop: Keccak block t + cpi Horner steps (mod.rs:186-205, 207) ===========================
    cs.native_seed_kind = 7;  // the sponge and the Horner recurrence have a native kernel (kernels_fsm_seed.hpp)
    cs.native_seed_param = n_chunks;
    cs.loop_begin(n_blocks);
    K k(g);
    NNField fr(g, BLS_FR);
    std::vector<zk_var> in, out;
    auto carry_in = [&]() {
        zk_var v = g.next_input();
        cs.link(ZK_LINK_FIRST, v, outer_zero);  // keccak state, opening value and counter all start at zero
        in.push_back(v);
        return v;
    };
    std::array<Lane, 25> st;
    for (auto& lane : st)
        for (auto& b : lane) b = carry_in();
    NNElement opening;
    for (auto& l : opening.limbs) l = carry_in();
    opening.bits = 17;  // the last chunk is added without a reduction
    zk_var t = carry_in();
    std::array<zk_var, 8> z;
    for (int i = 0; i < 8; ++i) z[i] = cs.loop_import(z_outer[i]);

    std::vector<zk_var> block_in(RATE), chunk_in(CHUNK * cpi);
    for (auto& b : block_in) b = g.next_input();
    for (auto& b : chunk_in) b = g.next_input();
    cs.stream_link(block_in.data(), RATE, chunk_in.data(), CHUNK * cpi, n_bytes);

    // flags from the iteration counter
    Boolean lt_full = g.overflowing_sub_with_borrow_in(UInt32{t}, UInt32{g.constant(full)}, g.bool_const(false)).second;
    Boolean eq_full = g.equals(t, g.constant(full));
    Boolean partial_active = g.b_or(lt_full, eq_full);
    Boolean is_t_last = g.equals(t, g.constant(t_last));
    Boolean is_last_block = g.equals(t, g.constant(n_blocks - 1));

    // Horner steps
    for (uint32_t c = 0; c < cpi; ++c) {
        const zk_var* b = chunk_in.data() + CHUNK * c;
        for (int i = 0; i + 1 < CHUNK; i += 2) g.range_check_u8_pair(b[i], b[i + 1]);  // BlobChunk::allocate
        g.range_check_u8_pair(b[CHUNK - 1], b[CHUNK - 1]);
        NNElement fe;  // convert_blob_chunk_to_field_element (mod.rs:75-102): little-endian
        for (int i = 0; i < 15; ++i) fe.limbs[i] = g.linear_combination({{b[2 * i], 1}, {b[2 * i + 1], 1ull << 8}});
        fe.limbs[15] = b[30];
        NNElement added = fr.add_lazy(opening, fe);
        NNElement multiplied = fr.mul_reduce(added, z.data(), 8);
        Boolean active = c < rem ? partial_active : lt_full;
        for (int i = 0; i < 16; ++i) {
            zk_var next = multiplied.limbs[i];
            if (c == c_last) next = g.select(is_t_last, added.limbs[i], next);  // `if cycle != limit - 1` (mod.rs:200-202)
            opening.limbs[i] = g.select(active, next, opening.limbs[i]);
        }
    }

    // Keccak block: data bytes, padding in the last block
    std::array<zk_var, RATE> padded_block;
    for (int j = 0; j < RATE; ++j) {
        zk_var v = block_in[j];
        if ((uint32_t)j >= rem_bytes) {
            uint64_t pad = ((uint32_t)j == rem_bytes ? 0x01 : 0x00) | (j == RATE - 1 ? 0x80 : 0x00);
            v = g.select(is_last_block, g.constant(pad), v);
        }
        padded_block[j] = v;
    }
    for (int j = RATE; j < 200; j += 2) g.range_check_u8_pair(st[j / 8][j % 8], st[(j + 1) / 8][(j + 1) % 8]);
    k.absorb_and_permute(st, padded_block.data());

    for (auto& lane : st)
        for (auto b : lane) out.push_back(b);
    for (auto l : opening.limbs) out.push_back(l);
    out.push_back(g.add(t, g.one()));
    for (size_t i = 0; i < in.size(); ++i) cs.link(ZK_LINK_CARRY, in[i], out[i]);
    cs.loop_end();

    // =========================== epilogue (mod.rs:207-259) ===========================
    K ke(g);
    NNField fro(g, BLS_FR);
    std::vector<zk_var> keccak256_hash(32);
    for (int j = 0; j < 32; ++j) {
        keccak256_hash[j] = cs.loop_last(out[j]);
        g.enforce_equal(linear_hash_output[j], keccak256_hash[j]);
    }
    NNElement y;
    for (int i = 0; i < 16; ++i) y.limbs[i] = cs.loop_last(out[200 + i]);
    y.bits = 17;
    y = fro.normalize(y);
    std::vector<zk_var> out_msg(versioned_hash);
    out_msg.insert(out_msg.end(), truncated.begin(), truncated.end());
    for (int i = 15; i >= 0; --i) {  // big-endian serialisation of the opening value
        auto [lo, hi] = fro.range_check_u16(y.limbs[i]);
        out_msg.push_back(hi);
        out_msg.push_back(lo);
    }
    auto output_hash = keccak256_one_block(g, ke, out_msg);

    // ClosedFormInputCompactForm::from_full_form with start_flag = completion_flag = true, observable_input = (),
    // hidden FSM parts = () (src/fsm_input_output/mod.rs:178-253)
    Boolean b_true = g.bool_const(true);
    Num zero_num = g.num_const(0);
    std::vector<zk_var> obs_out(keccak256_hash);
    obs_out.insert(obs_out.end(), output_hash.begin(), output_hash.end());
    auto c_obs_in = g.commit_encoding({});
    auto c_obs_out = g.commit_encoding(obs_out);
    auto c_fsm_in = g.commit_encoding({});
    auto c_fsm_out = g.commit_encoding({});
    std::vector<zk_var> compact = {b_true.v, b_true.v};
    for (int i = 0; i < 4; ++i) compact.push_back(c_obs_in[i].v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(b_true, c_obs_out[i], zero_num).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(b_true, zero_num, c_fsm_in[i]).v);
    for (int i = 0; i < 4; ++i) compact.push_back(g.select(b_true, zero_num, c_fsm_out[i]).v);
    auto input_commitment = g.commit_encoding(compact);
    for (auto& el : input_commitment) cs.place_gate(ZK_GATE_PUBLIC_INPUT, &el.v, 1, nullptr, 0);
}

}  // namespace zkgl
