/*
 * zkgl_ir.h — the closed witness IR + placement tables the engine executes on the GPU.
 *
 * The reference produces witness values with opaque Rust closures handed to
 * `set_values_with_dependencies{,_vararg}` (e.g. /root/reference/src/main_vm/opcodes/add_sub.rs:177-217,
 * src/main_vm/cycle.rs:912-928) and gate instances with `Gate::add_to_cs`
 * (src/main_vm/utils.rs:74-86).  Closures cannot cross to a GPU, so the recorder lowers every
 * closure the circuits use to one of the ops below and every gate to a
 * row descriptor.  One recorded *scope* = one straight-line program executed SIMT with
 * lane == circuit instance (outer scope) or lane == (instance, loop iteration) (loop scope).
 *
 * Storage: wave-tiled, cells[((lane >> 6) * n_cells + cell) * 64 + (lane & 63)] with cell = slot * n_columns + column for
 * trace cells; scratch cells (variables that no gate references) follow.  A wavefront touching one cell of its 64 lanes
 * therefore issues one coalesced 512-byte access, and everything one wavefront touches lies in one contiguous tile.
 *
 * Program = array of u32 words.  Operand word: bit31 = 0 -> own-scope cell index;
 * bits31..30 = 10 -> constant-pool index; 11 -> outer-scope cell index (loop scope only).
 * Destination list per produced value: one or more cell words, bit31 set on every word but the
 * last one of the list (a value is written to every cell its variable occupies).
 */
#ifndef ZKGL_IR_H
#define ZKGL_IR_H
#include <stdint.h>

#define ZK_OPERAND_CONST 0x80000000u
#define ZK_OPERAND_OUTER 0xC0000000u
#define ZK_OPERAND_KIND_MASK 0xC0000000u
#define ZK_OPERAND_IDX_MASK 0x3FFFFFFFu
#define ZK_DEST_MORE 0x80000000u
/* destination word = cell index (30 bits) | ZK_DEST_MORE when another destination of the same value follows */
#define ZK_DEST_CELL_MASK 0x3FFFFFFFu

/* header word = opcode | (a << 8) | (b << 16) ; a,b are small op parameters */
enum zk_opcode {
    ZK_OP_END = 0,
    ZK_OP_CONST = 1,     /* [const operand] -> out                                   (allocate_constant) */
    ZK_OP_INPUT = 2,     /* [word index]    -> out : per-lane input stream word       (witness allocation) */
    ZK_OP_FMA = 3,       /* [q, l, a, b, c] -> q*a*b + l*c                            (FmaGate witness) */
    ZK_OP_LC4 = 4,       /* [k0..k3, t0..t3] -> sum k_i t_i                           (ReductionGate witness) */
    ZK_OP_SELECT = 5,    /* [s, a, b] -> s ? a : b                                    (SelectionGate witness) */
    ZK_OP_ISZERO = 6,    /* [x] -> flag = (x==0), aux = x^-1 or 0                     (ZeroCheckGate witness) */
    ZK_OP_UADD = 7,      /* a=bits; [x, y, cin] -> (x+y+cin) mod 2^bits, carry        (UIntXAddGate witness) */
    ZK_OP_USUB = 8,      /* a=bits; [x, y, bin] -> (x-y-bin) mod 2^bits, borrow */
    ZK_OP_DOT4 = 9,      /* [a0,b0,..,a3,b3] -> sum a_i b_i                           (DotProductGate<4>) */
    ZK_OP_MATMUL12 = 10, /* a=matrix id (0 external, 1 inner); [in0..11] -> out0..11  (MatrixMultiplicationGate) */
    ZK_OP_SPLIT = 11,    /* a=nchunks, b=bits per chunk; [x] -> chunks (LSB first)    (decompose_into_bytes etc.) */
    ZK_OP_LOOKUP = 12,   /* a=nkeys, b=nvals; [table id word, keys..] -> vals         (perform_lookup) */
    ZK_OP_POSEIDON2 = 13,/* [in0..11] -> out0..11 witness-only permutation            (simulate_round_function);
                          * a=1: [in0..11, execute] -> execute ? permutation : 0^12, the reference's gated form
                          * simulate_round_function(cs, state, execute) (/root/reference/src/main_vm/opcodes/log.rs:532) */
    ZK_OP_P2_ROUNDS = 14,/* in-circuit permutation macro-op: [in0..11, rc cell x118] -> every intermediate, see zkgl_ir docs */
    ZK_OP_LOOP_LAST = 15,/* outer scope, post phase: [loop cell word] -> value at the last iteration */
    ZK_OP_U32MULADD = 16,/* [a, b, c, d] -> lo, hi of a*b + c + d  (u32 each)         (UInt32::fma_with_carry) */
    ZK_OP_ADD_CONSTMUL = 17, /* reserved */
    ZK_OP_DIVREM = 18,   /* b=divisor (1..65535); [x] -> x / b, x % b as integers       (UInt32::div_by_constant) */
    ZK_OP_NN_MULMOD = 19,/* a=nA, b=nB (<=17 each); [m0..m15 modulus limbs (consts), A limbs, B limbs] -> q limbs (nA+nB-15),
                          * r limbs (16): integers A*B = q*M + r, 0 <= r < M; all limbs base 2^16, input limbs < 2^24
                          *                                                           (NonNativeFieldOverU16 mul/normalize) */
    /* seed-only macro-ops (zk_cs_seed_hint): second producers of values the trace program computes gate by gate; they
     * exist only in the cone seeding program, where they replace the whole decomposition behind them */
    ZK_OP_KECCAK_ABSORB = 20, /* [state bytes x200 (byte k of lane x+5y at 8(x+5y)+k), block bytes x136] -> state bytes x200:
                               * Keccak-f[1600](state ^ (block | 0^64))                        (keccak256_absorb_and_run_permutation) */
    ZK_OP_SHA256_COMPRESS = 21,/* [state bytes x32 (word w little-endian at 4w), block bytes x64 (word j little-endian at 4j)]
                               * -> state bytes x32                                            (round_function_over_uint32) */
    /* strand programs only (never recorded, never exported): a scope short of wavefronts runs as 8 strands per 64-lane tile,
     * one wavefront each, with a workgroup barrier between the dependency levels of the op graph (cs.cpp build_strands) */
    ZK_OP_BARRIER = 22,
    /* 256-bit integer witness ops of the VM's mul / div / shift closures (limbs are u32, least significant first) */
    ZK_OP_U256_MULWIDE = 23, /* [a0..7, b0..7] -> 16 limbs of a * b         (allocate_mul_result_unchecked,
                              * /root/reference/src/main_vm/opcodes/mul_div.rs:20-92: ethereum_types::U256::full_mul) */
    ZK_OP_U256_DIVREM = 24,  /* [a0..7, b0..7] -> q0..7, r0..7 of a / b; b == 0 => q = 0, r = a
                              *                                              (allocate_div_result_unchecked, mul_div.rs:96-172) */
    ZK_OP_U8X4FMA = 25,      /* [a0..3, b0..3, c0..3, d0..3] (little-endian bytes of four u32) -> lo0..3, hi0..3, k0, k1 (bytes):
                              * a*b + c + d = lo + 2^32 hi; k = k0 + 256 k1 = the carry of the low 32 bits,
                              * k = (sum_{i+j<4} a_i b_j 2^(8(i+j)) + c + d) >> 32          (UInt32::fma_with_carry over U8x4FMAGate,
                              * /root/reference/src/main_vm/opcodes/mod.rs:146-158) */
    ZK_OP_KECCAK_F = 26,     /* macro-op (kernel K8), recorded by the engine's own Keccak gadget only: [state bytes x200 (byte k of lane x+5y at
                              * 8(x+5y)+k)] -> EVERY intermediate the byte-table decomposition of Keccak-f[1600] constrains (24 rounds of theta /
                              * rho-pi / chi / iota: Xor8 / AndN8 / ByteSplit outputs and the rotated bytes), in the order of
                              * csrc/keccak_macro.hpp zkk::keccak_f — the last values written are the 200 output bytes' producers
                              *                                                               (keccak256_absorb_and_run_permutation,
                              * /root/reference/src/keccak256_round_function/mod.rs:796-838) */
    ZK_OP_SHA256_ROUNDS = 27,/* macro-op (kernel K8), recorded by the engine's own SHA-256 gadget only: [state bytes x32 (word w little-endian at 4w),
                              * block bytes x64 (word j little-endian at 4j)] -> EVERY intermediate of the byte-table decomposition of one
                              * compression (message schedule, 64 rounds, feed-forward), in the order of csrc/sha256_macro.hpp zks::compress
                              *                                                               (round_function_over_uint32,
                              * /root/reference/src/sha256_round_function/mod.rs:271-285) */
    /* plain device programs of a loop scope only (never recorded, never exported; cs.cpp emit_scope): b = n - 1; n x [store slot,
     * plane id] -> nothing.  Copies n values the program later uses as SELECT flags into per-wavefront BIT PLANES in LDS (bit l of
     * plane id = value of lane l != 0; a second plane = value > 1, for the fused SelectionGate check).  A ZK_OP_SELECT with a = 1
     * carries plane ids instead of flag slots: 8 bytes from LDS instead of 512 from L2 / HBM, and a wavefront whose 64 lanes agree
     * on the flag loads only the selected operand. */
    ZK_OP_FLAG_PLANES = 28,
    ZK_OP_BYTEBUF_FILL = 29, /* macro-op (kernel K8), recorded by the engine's keccak256 precompile circuit only (opt-in, ZKGL_BYTEBUF_MACRO=1):
                              * [buffer bytes x192, filled, input bytes x32, offset, meaningful] -> EVERY intermediate of ByteBuffer::fill_with_bytes
                              * (the shift by `offset`, the 192 position markers, the 32 conditional placements, the new `filled`) in the order of
                              * csrc/bytebuf_macro.hpp zkb::fill_with_bytes
                              *                               (/root/reference/src/keccak256_round_function/buffer/mod.rs:69-136) */
    ZK_OP__COUNT
};

/* gate kinds (row descriptors).  Relations are restated from SURVEY.md §8 a2 / Appendix F. */
enum zk_gate_kind {
    ZK_GATE_NOP = 0,
    ZK_GATE_CONST = 1,     /* 1 var, 1 const : v - c */
    ZK_GATE_BOOLEAN = 2,   /* 1 var          : v^2 - v */
    ZK_GATE_FMA = 3,       /* a,b,c,d ; q,l  : q*a*b + l*c - d */
    ZK_GATE_REDUCTION4 = 4,/* t0..t3,r ; k0..k3 : sum k_i t_i - r */
    ZK_GATE_SELECT = 5,    /* a,b,s,r        : s*(a-b) + b - r */
    ZK_GATE_ZEROCHECK = 6, /* x,aux,flag     : x*aux - (1-flag) ; x*flag */
    ZK_GATE_UINTX_ADD = 7, /* a,b,cin,c,cout ; const 2^X : a+b+cin - c - 2^X*cout */
    ZK_GATE_DOT4 = 8,      /* a0,b0..a3,b3,r : sum a_i b_i - r */
    ZK_GATE_MATMUL12_EXT = 9,  /* in0..11,out0..11 : out - M_E in (12 relations) */
    ZK_GATE_MATMUL12_INT = 10, /* in0..11,out0..11 : out - M_I in (12 relations) */
    ZK_GATE_PUBLIC_INPUT = 11, /* 1 var : marks the cell public, no relation */
    ZK_GATE_U32_FMA = 12,  /* a,b,c,d,lo,hi  : a*b + c + d - lo - 2^32 hi  (U8x4FMAGate role) */
    ZK_GATE_REDUCTION_BY_POWERS4 = 13, /* t0..t3,r ; c : t0 + c t1 + c^2 t2 + c^3 t3 - r  (ReductionByPowersGate<F,4>,
                                        * /root/reference/src/main_vm/decoded_opcode.rs:275, opcodes/binop.rs:203-217) */
    ZK_GATE_U8X4_FMA = 14, /* a0..3,b0..3,c0..3,d0..3,lo0..3,hi0..3,k0,k1 (26 byte variables), two relations over integers < 2^50:
                            *   sum_{i+j<4}  a_i b_j 2^(8(i+j))   + c + d - lo - 2^32 (k0 + 256 k1)
                            *   sum_{i+j>=4} a_i b_j 2^(8(i+j-4)) + (k0 + 256 k1) - hi
                            * (U8x4FMAGate, /root/reference/src/main_vm/opcodes/mod.rs:146; boojum's column order is [EXT]).  Unlike the
                            * one-relation ZK_GATE_U32_FMA, whose a*b + c + d and lo + 2^32 hi both range up to 2^64 - 1 > p and may
                            * therefore differ by p, no term here can wrap the field once the bytes are range-checked. */
    ZK_GATE__COUNT
};

/* per-slot gate row descriptor (uniform across lanes; lives in scalar/constant memory) */
typedef struct zk_row_desc {
    uint32_t kind;        /* zk_gate_kind */
    uint32_t n_instances; /* instance j occupies columns [j*width, (j+1)*width) */
    uint32_t const_off;   /* offset into the scope's row-constant pool */
    uint32_t n_consts;    /* constants shared by every instance of the row */
} zk_row_desc;

/* per-slot lookup row descriptor: up to `reps` tuples of `width` columns, one table per row */
typedef struct zk_lookup_row_desc {
    uint32_t table;   /* table id, 0xffffffff = no lookups in this row */
    uint32_t n_tuples;
} zk_lookup_row_desc;

/* table descriptor on the device */
typedef struct zk_table_desc {
    uint32_t word_off;  /* offset (in u64 words) into the packed table storage; rows are row-major */
    uint32_t mult_off;  /* global row number of row 0 (index into the multiplicity vector) */
    uint32_t n_rows;
    uint32_t n_keys;
    uint32_t n_vals;
    uint32_t dense;     /* 1: row index = sum key_i << key_shift[i] */
    uint32_t key_shift[3];
} zk_table_desc;

/* copy-constraint records */
typedef struct zk_copy_pair { uint32_t cell; uint32_t home; } zk_copy_pair;
enum zk_link_kind {
    ZK_LINK_CARRY = 0, /* loop: in_cell[k+1] == out_cell[k] */
    ZK_LINK_FIRST = 1, /* loop in_cell[k=0] == outer cell */
    ZK_LINK_LAST = 2,  /* outer cell == loop out_cell[k=limit-1] */
    ZK_LINK_BCAST = 3  /* loop cell[k] == outer cell for every k */
};
typedef struct zk_link { uint32_t kind; uint32_t loop_cell; uint32_t other_cell; uint32_t pad; } zk_link;
/* Stream link: two periodic word sequences of the loop scope carry the same data with different chunking.  For every
 * global index k < n_total:  cell a[k % pa] of iteration k / pa  ==  cell b[k % pb] of iteration k / pb  (copy
 * constraints across iterations; e.g. the blob bytes of eip_4844 seen as 31-byte chunks and as 136-byte Keccak blocks).
 * Serialised as: pa, pb, n_total, a cells[pa], b cells[pb]. */

#endif
