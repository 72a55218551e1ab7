// circuits/opcode_defs.cpp — [EXT] this build's `zkevm_opcode_defs` blob (include/zkgl_vm.h).
//
// The crate zkevm_opcode_defs (Cargo.toml:18) is not in /root/reference; what the circuit needs from it is data only.  The blob
// built here has the reference's SHAPE — 2^11 rows of (price, 51-bit property bitmask), 16 opcode-type bits / 10 variant bits /
// 2 flag bits / 6 + 4 addressing bits / 3 aux bits (src/main_vm/opcode_bitmask.rs:21-27,60-128; decoded_opcode.rs:81-84) — and
// this build's own enumeration: families in Opcode::variant_idx order, then variant, source mode, destination mode, flag bits.
// Constants are recollections of zkevm_opcode_defs v1.4.1 and are marked so; nothing in the circuit depends on their values.
#include <cstring>
#include "../../../include/zkgl_vm.h"

namespace {

struct VariantSpec { uint32_t sub_idx; bool kernel_only, static_ok; uint32_t price; uint32_t n_flag_bits; bool src_imm_ok; };
struct FamilySpec {
    uint32_t family;
    uint32_t src_modes;  // bit mask over zk_vm_operand_mode
    uint32_t dst_modes;
    uint32_t can_write_dst0_into_memory;
    uint32_t n_variants;
    VariantSpec variants[10];
};

constexpr uint32_t M_REG = 1u << ZK_VMM_REG_ONLY;
constexpr uint32_t M_ALL_SRC = 0x3f, M_ALL_DST = 0x0f;
constexpr uint32_t AVERAGE = 6, RICH = 8;  // [EXT] AVERAGE_OPCODE_ERGS / RICH_ADDRESSING_OPCODE_ERGS

const FamilySpec FAMILIES[] = {
    {ZK_VMF_INVALID, M_REG, M_REG, 0, 1, {{0, false, true, 0xffffffffu, 0, false}}},
    {ZK_VMF_NOP, M_ALL_SRC, M_ALL_DST, 1, 1, {{0, false, true, RICH, 0, false}}},
    {ZK_VMF_ADD, M_ALL_SRC, M_ALL_DST, 1, 1, {{0, false, true, RICH, 1, false}}},
    {ZK_VMF_SUB, M_ALL_SRC, M_ALL_DST, 1, 1, {{0, false, true, RICH, 2, false}}},
    {ZK_VMF_MUL, M_ALL_SRC, M_ALL_DST, 1, 1, {{0, false, true, RICH, 1, false}}},
    {ZK_VMF_DIV, M_ALL_SRC, M_ALL_DST, 1, 1, {{0, false, true, RICH, 2, false}}},
    {ZK_VMF_JUMP, M_ALL_SRC, M_REG, 0, 1, {{0, false, true, RICH, 0, false}}},
    {ZK_VMF_CONTEXT, M_REG, M_REG, 0, 10,
     {{0, false, true, AVERAGE, 0, false}, {1, false, true, AVERAGE, 0, false}, {2, false, true, AVERAGE, 0, false}, {3, false, true, AVERAGE, 0, false},
      {4, false, true, AVERAGE, 0, false}, {5, false, true, AVERAGE, 0, false}, {6, false, true, AVERAGE, 0, false}, {7, true, false, AVERAGE, 0, false},
      {8, true, false, AVERAGE, 0, false}, {9, true, false, AVERAGE, 0, false}}},
    {ZK_VMF_SHIFT, M_ALL_SRC, M_ALL_DST, 1, 4, {{0, false, true, RICH, 2, false}, {1, false, true, RICH, 2, false}, {2, false, true, RICH, 2, false}, {3, false, true, RICH, 2, false}}},
    {ZK_VMF_BINOP, M_ALL_SRC, M_ALL_DST, 1, 3, {{0, false, true, RICH, 1, false}, {1, false, true, RICH, 1, false}, {2, false, true, RICH, 1, false}}},
    {ZK_VMF_PTR, M_ALL_SRC, M_ALL_DST, 1, 4, {{0, false, true, RICH, 1, false}, {1, false, true, RICH, 1, false}, {2, false, true, RICH, 1, false}, {3, false, true, RICH, 1, false}}},
    {ZK_VMF_NEAR_CALL, M_REG, M_REG, 0, 1, {{0, false, true, AVERAGE + 20, 0, false}}},
    {ZK_VMF_LOG, M_REG, M_REG, 0, 5,
     {{0, false, true, 160, 0, false}, {1, false, false, 320, 0, false}, {2, true, false, 156, 1, false}, {3, true, false, 46, 1, false}, {4, true, true, 16, 0, false}}},
    {ZK_VMF_FAR_CALL, M_REG, M_REG, 0, 3, {{0, false, true, 182, 2, false}, {1, false, true, 182, 2, false}, {2, true, true, 182, 2, false}}},
    {ZK_VMF_RET, M_REG, M_REG, 0, 3, {{0, false, true, AVERAGE, 1, false}, {1, false, true, AVERAGE, 1, false}, {2, false, true, AVERAGE, 1, false}}},
    {ZK_VMF_UMA, M_REG, M_REG, 0, 5,
     {{0, false, true, 13, 1, true}, {1, false, true, 13, 1, true}, {2, false, true, 13, 1, true}, {3, false, true, 13, 1, true}, {4, false, true, 9, 1, false}}},
};

}  // namespace

extern "C" int zk_opcode_defs_default(zk_opcode_defs* d) {
    if (!d) return ZK_ERR_INVALID;
    std::memset(d, 0, sizeof *d);
    d->version = 1;
    d->type_bits = 16; d->variant_bits = 10; d->flag_bits = 2; d->src_mode_bits = 6; d->dst_mode_bits = 4;
    d->description_bits_flattened = 48; d->aux_bits = 3;
    d->aux_kernel_mode = 0; d->aux_static_ok = 1; d->aux_explicit_panic = 2;
    const uint32_t vidx[ZK_VMV__COUNT] = {0, 1, 2, 3, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 4};
    std::memcpy(d->variant_idx, vidx, sizeof vidx);
    d->flag_idx[ZK_VMFL_SET_FLAGS] = 0; d->flag_idx[ZK_VMFL_SWAP_ARITH] = 1; d->flag_idx[ZK_VMFL_SWAP_PTR] = 0;
    d->flag_idx[ZK_VMFL_FIRST_MESSAGE] = 0; d->flag_idx[ZK_VMFL_UMA_INCREMENT] = 0; d->flag_idx[ZK_VMFL_FAR_CALL_STATIC] = 0;
    d->flag_idx[ZK_VMFL_FAR_CALL_SHARD] = 1; d->flag_idx[ZK_VMFL_RET_TO_LABEL] = 0;
    // [EXT] Condition::variant_index: Always, Gt, Lt, Eq, Ge, Le, Ne, GtOrLt = 0..7
    d->condition_idx[ZK_VMC_ALWAYS] = 0; d->condition_idx[ZK_VMC_GT] = 1; d->condition_idx[ZK_VMC_LT] = 2; d->condition_idx[ZK_VMC_EQ] = 3;
    d->condition_idx[ZK_VMC_GE] = 4; d->condition_idx[ZK_VMC_LE] = 5; d->condition_idx[ZK_VMC_NE] = 6; d->condition_idx[ZK_VMC_GT_OR_LT] = 7;

    const uint32_t variant_bit0 = d->type_bits, flag_bit0 = variant_bit0 + d->variant_bits, src_bit0 = flag_bit0 + d->flag_bits,
                   dst_bit0 = src_bit0 + d->src_mode_bits, aux_bit0 = d->description_bits_flattened;
    const uint64_t invalid_props = (1ull << ZK_VMF_INVALID) | (1ull << variant_bit0) | (1ull << (src_bit0 + ZK_VMM_REG_ONLY)) |
                                   (1ull << (dst_bit0 + ZK_VMM_REG_ONLY)) | (1ull << (aux_bit0 + d->aux_static_ok)) |
                                   (1ull << (aux_bit0 + d->aux_explicit_panic));
    uint32_t n = 0;
    for (const FamilySpec& f : FAMILIES) {
        d->can_write_dst0_into_memory[f.family] = f.can_write_dst0_into_memory;
        for (uint32_t vi = 0; vi < f.n_variants; ++vi) {
            const VariantSpec& v = f.variants[vi];
            for (uint32_t sm = 0; sm < ZK_VMM__COUNT; ++sm) {
                const bool src_ok = ((f.src_modes >> sm) & 1) || (v.src_imm_ok && sm == ZK_VMM_IMM16);
                if (!src_ok) continue;
                for (uint32_t dm = 0; dm < 4; ++dm) {
                    if (!((f.dst_modes >> dm) & 1)) continue;
                    for (uint32_t fl = 0; fl < (1u << v.n_flag_bits); ++fl) {
                        if (n >= ZK_VM_OPCODE_TABLE_ROWS) return ZK_ERR_CAPACITY;
                        uint64_t p = (1ull << f.family) | (1ull << (variant_bit0 + v.sub_idx)) | ((uint64_t)fl << flag_bit0) |
                                     (1ull << (src_bit0 + sm)) | (1ull << (dst_bit0 + dm));
                        if (v.kernel_only) p |= 1ull << (aux_bit0 + d->aux_kernel_mode);
                        if (v.static_ok) p |= 1ull << (aux_bit0 + d->aux_static_ok);
                        if (f.family == ZK_VMF_INVALID) p = invalid_props;
                        d->props[n] = p;
                        d->prices[n] = v.price;
                        ++n;
                    }
                }
            }
        }
    }
    d->n_valid = n;
    for (uint32_t i = n; i < ZK_VM_OPCODE_TABLE_ROWS; ++i) { d->props[i] = invalid_props; d->prices[i] = 0xffffffffu; }
    const int nop = zk_opcode_defs_find(d, ZK_VMF_NOP, 0, ZK_VMM_REG_ONLY, ZK_VMM_REG_ONLY, 0);
    const int panic = zk_opcode_defs_find(d, ZK_VMF_RET, d->variant_idx[ZK_VMV_RET_PANIC], ZK_VMM_REG_ONLY, ZK_VMM_REG_ONLY, 0);
    if (nop < 0 || panic < 0) return ZK_ERR_INVALID;
    d->nop_encoding = (uint64_t)nop;      // condition Always (0), registers r0, immediates 0
    d->panic_encoding = (uint64_t)panic;
    d->nop_bitspread = d->props[nop];
    d->panic_bitspread = d->props[panic];

    uint32_t* P = d->params;  // [EXT] recollected values of zkevm_opcode_defs::system_params v1.4.1
    P[ZK_VMP_VM_INITIAL_FRAME_ERGS] = 0xffffffffu;
    P[ZK_VMP_VM_MAX_STACK_DEPTH] = 0xffffffffu / 20 + 80;
    P[ZK_VMP_NEW_FRAME_MEMORY_STIPEND] = 1u << 12;
    P[ZK_VMP_NEW_MEMORY_PAGES_PER_FAR_CALL] = 8;
    P[ZK_VMP_UNMAPPED_PAGE] = 0;
    P[ZK_VMP_BOOTLOADER_BASE_PAGE] = 8;
    P[ZK_VMP_BOOTLOADER_CODE_PAGE] = 8;
    P[ZK_VMP_BOOTLOADER_CALLDATA_PAGE] = 7;
    P[ZK_VMP_STARTING_BASE_PAGE] = 16;  // first free page after the bootloader's frame (base 8: code 8, stack 9, heap 10, aux heap 11)
    P[ZK_VMP_STARTING_TIMESTAMP] = 1024;
    P[ZK_VMP_INITIAL_FRAME_FORMAL_EH_LOCATION] = 0xffff;
    P[ZK_VMP_BOOTLOADER_FORMAL_ADDRESS_LOW] = 0x8001;
    P[ZK_VMP_BOOTLOADER_MAX_MEMORY] = 1u << 24;
    P[ZK_VMP_DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW] = 0x8002;
    P[ZK_VMP_ERGS_PER_CODE_WORD_DECOMMITTMENT] = 4;
    P[ZK_VMP_INITIAL_STORAGE_WRITE_PUBDATA_BYTES] = 64;
    P[ZK_VMP_L1_MESSAGE_PUBDATA_BYTES] = 88;
    P[ZK_VMP_STORAGE_AUX_BYTE] = 0; P[ZK_VMP_EVENT_AUX_BYTE] = 1; P[ZK_VMP_L1_MESSAGE_AUX_BYTE] = 2; P[ZK_VMP_PRECOMPILE_AUX_BYTE] = 3;
    P[ZK_VMP_CODE_HASH_VERSION_BYTE] = 1; P[ZK_VMP_CODE_YET_CONSTRUCTED_MARKER] = 1; P[ZK_VMP_CODE_AT_REST_MARKER] = 0;
    P[ZK_VMP_FAR_CALL_FORWARDING_MODE_BYTE_IDX] = 28; P[ZK_VMP_FAR_CALL_SHARD_ID_BYTE_IDX] = 29;
    P[ZK_VMP_FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX] = 30; P[ZK_VMP_FAR_CALL_SYSTEM_CALL_BYTE_IDX] = 31;
    P[ZK_VMP_FORWARD_USE_HEAP] = 0; P[ZK_VMP_FORWARD_FAT_POINTER] = 1; P[ZK_VMP_FORWARD_USE_AUX_HEAP] = 2;
    P[ZK_VMP_CALL_IMPLICIT_PARAMETER_REG_IDX] = 14;
    P[ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_BEGIN] = 2; P[ZK_VMP_CALL_SYSTEM_ABI_REGISTERS_END] = 12;
    P[ZK_VMP_CALL_RESERVED_RANGE_BEGIN] = 12; P[ZK_VMP_CALL_RESERVED_RANGE_END] = 14;
    return ZK_OK;
}

extern "C" int zk_opcode_defs_find(const zk_opcode_defs* d, uint32_t family, uint32_t variant, uint32_t src_mode, uint32_t dst_mode, uint32_t flags) {
    if (!d || family >= ZK_VMF__COUNT || variant >= d->variant_bits || src_mode >= d->src_mode_bits || dst_mode >= d->dst_mode_bits ||
        flags >= (1u << d->flag_bits))
        return -1;
    const uint32_t variant_bit0 = d->type_bits, flag_bit0 = variant_bit0 + d->variant_bits, src_bit0 = flag_bit0 + d->flag_bits,
                   dst_bit0 = src_bit0 + d->src_mode_bits;
    const uint64_t want = (1ull << family) | (1ull << (variant_bit0 + variant)) | ((uint64_t)flags << flag_bit0) | (1ull << (src_bit0 + src_mode)) |
                          (1ull << (dst_bit0 + dst_mode));
    const uint64_t mask = (1ull << d->description_bits_flattened) - 1;
    for (uint32_t i = 0; i < d->n_valid; ++i)
        if ((d->props[i] & mask) == want) return (int)i;
    return -1;
}
