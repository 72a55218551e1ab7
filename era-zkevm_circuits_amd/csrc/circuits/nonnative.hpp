// circuits/nonnative.hpp — non-native prime field over 16-bit limbs (counterpart of boojum's
// `NonNativeFieldOverU16<F, T, 17>` [EXT] as used by /root/reference/src/eip_4844/mod.rs:40-42, 186-204, 217-227).
//
// boojum's decomposition is absent from the tree; this one is the engine's own:
//   * an element is 16 limbs (base 2^16, little-endian); `bits` bounds every limb (16 = normalised limbs,
//     17+ after lazy additions);
//   * mul_reduce(a, b): witness q, r = divmod(A*B, M) by the ZK_OP_NN_MULMOD big-integer op; the integer identity
//     A*B = q*M + r is enforced column by column in base 2^16 with signed carries:
//         sum_{i+j=k} a_i b_j - sum_{i+j=k} q_i m_j - r_k + c_{k-1} = 2^16 c_k ,   c_{-1} = c_last = 0
//     (every magnitude stays below 2^50, so equality in the Goldilocks field is equality over the integers);
//     q, r limbs are range-checked to 16 bits, the offset carries to 32 bits.  r is only forced below 2^256;
//   * normalize(a): r = a mod M with the extra constraint r <= M - 1 (16-limb borrow chain), i.e. canonical.
#pragma once
#include "../gadgets.hpp"

namespace zkgl {

struct NNElement {
    std::array<zk_var, 16> limbs;
    uint32_t bits = 16;  // every limb < 2^bits
};

class NNField {
  public:
    NNField(G& g, const std::array<uint32_t, 16>& modulus_limbs) : g(g), m(modulus_limbs) {}
    G& g;
    std::array<uint32_t, 16> m;

    // x = lo + 2^8 hi with both bytes range-checked; returns {lo, hi}
    std::pair<zk_var, zk_var> range_check_u16(zk_var x) {
        zk_var first = g.cs.alloc_vars(2);
        zk_var parts[2] = {first, first + 1};
        g.cs.emit_op(ZK_OP_SPLIT, 2, 8, &x, 1, parts, 2, nullptr, 0);
        g.enforce_equal(g.linear_combination({{parts[0], 1}, {parts[1], 1ull << 8}}), x);
        g.range_check_u8_pair(parts[0], parts[1]);
        return {parts[0], parts[1]};
    }

    NNElement add_lazy(const NNElement& a, const NNElement& b) {
        NNElement r;
        for (int i = 0; i < 16; ++i) r.limbs[i] = g.add(a.limbs[i], b.limbs[i]);
        r.bits = std::max(a.bits, b.bits) + 1;
        return r;
    }

    // (A * B) mod M; `b` holds nb <= 16 meaningful limbs of 16 bits
    NNElement mul_reduce(const NNElement& a, const zk_var* b, uint32_t nb) {
        if (a.bits > 20) throw ZkError(ZK_ERR_INVALID, "NNField::mul_reduce: operand limbs too wide");
        const uint32_t na = 16, nq = na + nb - 15;
        std::vector<zk_var> ins(a.limbs.begin(), a.limbs.end());
        ins.insert(ins.end(), b, b + nb);
        std::vector<zk_var> outs(nq + 16);
        zk_var first = g.cs.alloc_vars(nq + 16);
        for (uint32_t i = 0; i < nq + 16; ++i) outs[i] = first + i;
        uint64_t imm[16];
        for (int i = 0; i < 16; ++i) imm[i] = m[i];
        g.cs.emit_op(ZK_OP_NN_MULMOD, na, nb, ins.data(), na + nb, outs.data(), nq + 16, imm, 16);
        for (auto v : outs) (void)range_check_u16(v);
        const zk_var* q = outs.data();
        const zk_var* r = outs.data() + nq;

        const uint64_t OFF = 1ull << 26;  // carries are stored as c + OFF in [0, 2^32)
        const uint32_t ncols = std::max(na + nb - 1, nq + 16 - 1);
        zk_var carry = ZK_VAR_NONE;  // offset carry of the previous column
        for (uint32_t k = 0; k < ncols; ++k) {
            zk_var acc = ZK_VAR_NONE;  // sum_{i+j=k} a_i b_j
            for (uint32_t i = 0; i < na; ++i) {
                if (k < i || k - i >= nb) continue;
                acc = acc == ZK_VAR_NONE ? g.mul(a.limbs[i], b[k - i]) : g.fma(1, a.limbs[i], b[k - i], 1, acc);
            }
            // e_k = acc - sum q_i m_j - r_k + (carry - OFF) + OFF * 2^16  ==  2^16 * (c_k + OFF)
            std::vector<std::pair<zk_var, uint64_t>> terms;
            if (acc != ZK_VAR_NONE) terms.push_back({acc, 1});
            for (uint32_t i = 0; i < nq; ++i)
                if (k >= i && k - i < 16 && m[k - i]) terms.push_back({q[i], GL_P - m[k - i]});
            if (k < 16) terms.push_back({r[k], GL_P - 1});
            uint64_t constant = OFF << 16;
            if (carry != ZK_VAR_NONE) { terms.push_back({carry, 1}); constant -= OFF; }
            terms.push_back({g.one(), constant});
            zk_var e = g.linear_combination(terms);
            if (k + 1 == ncols) {  // c_last = 0
                g.enforce_equal(e, g.constant(OFF << 16));
                break;
            }
            zk_var f2 = g.cs.alloc_vars(2);
            zk_var parts[2] = {f2, f2 + 1};  // low 16 bits (zero for an honest witness), offset carry
            g.cs.emit_op(ZK_OP_SPLIT, 2, 16, &e, 1, parts, 2, nullptr, 0);
            g.enforce_equal(g.fma(1ull << 16, parts[1], g.one(), 0, parts[1]), e);
            g.range_check_u32(parts[1]);
            carry = parts[1];
        }
        NNElement res;
        for (int i = 0; i < 16; ++i) res.limbs[i] = r[i];
        res.bits = 16;
        return res;
    }

    // canonical representative: r = a mod M, 0 <= r <= M - 1
    NNElement normalize(const NNElement& a) {
        zk_var one = g.one();
        NNElement r = mul_reduce(a, &one, 1);
        Boolean borrow = g.bool_const(false);
        for (int i = 0; i < 16; ++i) {  // (M - 1) - r >= 0 limb by limb
            uint32_t mi = m[i];
            if (i == 0) mi -= 1;  // M is odd
            zk_var outs[2] = {g.cs.alloc_var(), g.cs.alloc_var()};  // diff, borrow
            zk_var ins[3] = {g.constant(mi), r.limbs[i], borrow.v};
            g.cs.emit_op(ZK_OP_USUB, 16, 0, ins, 3, outs, 2, nullptr, 0);
            zk_var vars[5] = {r.limbs[i], outs[0], borrow.v, ins[0], outs[1]};  // r + diff + bin = m + 2^16 * bout
            uint64_t k = 1ull << 16;
            g.cs.place_gate(ZK_GATE_UINTX_ADD, vars, 5, &k, 1);
            g.cs.place_gate(ZK_GATE_BOOLEAN, &outs[1], 1, nullptr, 0);
            (void)range_check_u16(outs[0]);
            borrow = Boolean{outs[1]};
        }
        g.enforce_zero(borrow.v);
        return r;
    }
};

}  // namespace zkgl
