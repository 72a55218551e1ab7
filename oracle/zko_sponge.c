/* oracle/zko_sponge.c — CPU ORACLE (test infrastructure).
 * Sponge / queue / commitment / grand-product rules restated from the reference. */
#include "zko.h"
#include <string.h>

/* commit_encoding — /root/reference/src/fsm_input_output/mod.rs:281-326.
 * [EXT] assumption: apply_length_specialization writes the length into the LAST state
 * element (index 11); not derivable from the tree. */
void zko_commit_encoding(const uint64_t *input, size_t len, uint64_t out[4]) {
    uint64_t st[12];
    memset(st, 0, sizeof st);
    st[11] = zko_gl_reduce((uint64_t)len);
    size_t nchunks = (len + 7) / 8;
    for (size_t c = 0; c < nchunks; ++c) {
        for (size_t j = 0; j < 8; ++j) {
            size_t k = 8 * c + j;
            st[j] = k < len ? input[k] : 0; /* replace rate, keep capacity (mod.rs:316-321) */
        }
        zko_poseidon2_permute(st);
    }
    memcpy(out, st, 4 * sizeof(uint64_t));
}

/* produce_fs_challenges — /root/reference/src/utils.rs:12-78 */
void zko_fs_challenges(const uint64_t *fs_input, size_t len, uint64_t *out, size_t reps, size_t nchal) {
    uint64_t st[12];
    memset(st, 0, sizeof st);
    st[11] = zko_gl_reduce((uint64_t)len);
    size_t nchunks = (len + 7) / 8; /* full chunks then zero-padded remainder (utils.rs:39-55) */
    for (size_t c = 0; c < nchunks; ++c) {
        for (size_t j = 0; j < 8; ++j) {
            size_t k = 8 * c + j;
            st[j] = k < len ? fs_input[k] : 0;
        }
        zko_poseidon2_permute(st);
    }
    size_t can_take = 8;
    for (size_t r = 0; r < reps; ++r) {
        out[r * nchal] = 1; /* utils.rs:61-63: slot 0 is the constant one */
        for (size_t i = 1; i < nchal; ++i) {
            if (can_take == 0) { zko_poseidon2_permute(st); can_take = 8; }
            out[r * nchal + i] = st[8 - can_take];
            --can_take;
        }
    }
}

/* full-state queue push — /root/reference/src/main_vm/utils.rs:194-213 */
void zko_queue_full_push(uint64_t tail[12], const uint64_t enc[8]) {
    memcpy(tail, enc, 8 * sizeof(uint64_t));
    zko_poseidon2_permute(tail);
}

/* 4-wide tail queue, 20-element encoding — /root/reference/src/main_vm/opcodes/log.rs:508-585 */
void zko_queue_tail4_push20(uint64_t tail[4], const uint64_t enc[20]) {
    uint64_t st[12];
    memset(st, 0, sizeof st); /* create_empty_state, no length specialisation (log.rs:510-511) */
    memcpy(st, enc, 8 * sizeof(uint64_t));
    zko_poseidon2_permute(st);
    memcpy(st, enc + 8, 8 * sizeof(uint64_t));
    zko_poseidon2_permute(st);
    memcpy(st, enc + 16, 4 * sizeof(uint64_t));
    memcpy(st + 4, tail, 4 * sizeof(uint64_t));
    zko_poseidon2_permute(st);
    memcpy(tail, st, 4 * sizeof(uint64_t));
}

/* MemoryQuery::encode — /root/reference/src/base_structures/memory_query/mod.rs:103-221 */
void zko_memory_query_encode(const uint64_t q[13], uint64_t enc[8]) {
    const uint64_t *v = q + 5; /* value limbs */
    uint64_t b5[4], b6[4], b7[4];
    for (int i = 0; i < 4; ++i) {
        b5[i] = (v[5] >> (8 * i)) & 0xff;
        b6[i] = (v[6] >> (8 * i)) & 0xff;
        b7[i] = (v[7] >> (8 * i)) & 0xff;
    }
    enc[0] = q[0];
    enc[1] = q[1];
    enc[2] = q[2] + (q[3] << 32) + (q[4] << 33);
    enc[3] = v[0] + (b5[0] << 32) + (b5[1] << 40) + (b5[2] << 48);
    enc[4] = v[1] + (b5[3] << 32) + (b6[0] << 40) + (b6[1] << 48);
    enc[5] = v[2] + (b6[2] << 32) + (b6[3] << 40) + (b7[0] << 48);
    enc[6] = v[3] + (b7[1] << 32) + (b7[2] << 40) + (b7[3] << 48);
    enc[7] = v[4];
}

/* ExecutionContextRecord::encode — /root/reference/src/base_structures/vm_state/saved_context.rs:111-266 */
void zko_execution_context_encode(const uint64_t r[42], uint64_t e[32]) {
    const uint64_t *this_ = r, *caller = r + 5, *code_address = r + 10;
    uint64_t code_page = r[15], base_page = r[16], heap_ub = r[17], aux_heap_ub = r[18];
    const uint64_t *rq_head = r + 19, *rq_tail = r + 23;
    uint64_t seg_len = r[27], pc = r[28], sp = r[29], eh_loc = r[30], ergs = r[31];
    uint64_t is_static = r[32], is_kernel = r[33], this_shard = r[34], caller_shard = r[35], code_shard = r[36];
    const uint64_t *ctx128 = r + 37;
    uint64_t is_local = r[41];
    for (int i = 0; i < 4; ++i) { e[i] = rq_head[i]; e[4 + i] = rq_tail[i]; }
    for (int i = 0; i < 5; ++i) { e[8 + i] = code_address[i]; e[13 + i] = this_[i]; e[18 + i] = caller[i]; }
    for (int i = 0; i < 4; ++i) e[23 + i] = ctx128[i];
    e[27] = code_page + (pc << 32) + (this_shard << 48) + (is_static << 56);
    e[28] = base_page + (sp << 32) + (caller_shard << 48) + (is_kernel << 56);
    e[29] = ergs + (eh_loc << 32) + (code_shard << 48) + (is_local << 56);
    e[30] = heap_ub + ((seg_len & 0xff) << 32) + (((seg_len >> 8) & 0xff) << 40);
    e[31] = aux_heap_ub + (((seg_len >> 16) & 0xff) << 32) + (((seg_len >> 24) & 0xff) << 40);
}

/* accumulate_grand_products — /root/reference/src/utils.rs:81-137, one repetition */
void zko_grand_product(const uint64_t *enc, const uint8_t *flags, const uint64_t *ch,
                       size_t enc_len, size_t n, uint64_t init, uint64_t *acc_out) {
    uint64_t acc = init;
    for (size_t i = 0; i < n; ++i) {
        uint64_t contrib = ch[enc_len];
        for (size_t j = 0; j < enc_len; ++j)
            contrib = zko_gl_add(zko_gl_mul(enc[i * enc_len + j], ch[j]), contrib);
        if (flags[i]) acc = zko_gl_mul(acc, contrib);
        acc_out[i] = acc;
    }
}
