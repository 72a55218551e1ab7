"""A second, independently produced opcode table (round-3 VERDICT item 8 / round-4 item 8).

`csrc/circuits/opcode_defs.cpp` builds this build's `zkevm_opcode_defs` blob from nested C++ aggregates indexed by enum values.  This
file states the same ISA as a LISTING BY NAME — one line per instruction variant, as the zkEVM ISA primer lists them: which operand
addressing it admits, which modifier flags it has, whether it is kernel-only, whether a static context may run it, its price class —
expands the listing by a different procedure (a dictionary keyed by names, `itertools.product` over named axes, bit positions looked
up by name in the layout of /root/reference/src/main_vm/opcode_bitmask.rs:21-27,60-128) and compares the two tables row for row.
It also checks what any such table must satisfy whatever its enumeration: one-hot groups, unique descriptions, the row count formula,
the find() round trip, the NOP / PANIC rows.  [EXT] the crate itself is absent (Cargo.toml:18); both sides are restatements, but no
line of one was derived from the other."""
import itertools

import zkgl
from zkgl import VM_FAMILY as F

# ---- the layout, by name (opcode_bitmask.rs:21-27): [16 opcode types | 10 variants | 2 flags | 6 source modes | 4 destination modes], aux from bit 48
GROUPS = [("type", 16), ("variant", 10), ("flag", 2), ("src", 6), ("dst", 4)]
AUX_BASE, AUX = 48, {"kernel_only": 0, "static_ok": 1, "explicit_panic": 2}
SRC_MODES = ["reg", "stack_pop", "stack_relative", "stack_absolute", "imm16", "code_page"]      # ImmMemHandlerFlags order
DST_MODES = ["reg", "stack_push", "stack_relative", "stack_absolute"]
FAMILY_ORDER = ["invalid", "nop", "add", "sub", "mul", "div", "jump", "context", "shift", "binop", "ptr", "near_call", "log", "far_call", "ret", "uma"]
RICH, AVERAGE = 8, 6                                                                            # price classes (ergs)
ANY_SRC, ANY_DST, REG = SRC_MODES, DST_MODES, ["reg"]

# ---- the listing: name -> (source modes, destination modes, modifier flags, kernel only, allowed in static context, price)
ISA = {
    "invalid":                         (REG, REG, [], False, True, 0xFFFFFFFF),
    "nop":                             (ANY_SRC, ANY_DST, [], False, True, RICH),
    "add":                             (ANY_SRC, ANY_DST, ["set_flags"], False, True, RICH),
    "sub":                             (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "mul":                             (ANY_SRC, ANY_DST, ["set_flags"], False, True, RICH),
    "div":                             (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "jump":                            (ANY_SRC, REG, [], False, True, RICH),
    "context.this":                    (REG, REG, [], False, True, AVERAGE),
    "context.caller":                  (REG, REG, [], False, True, AVERAGE),
    "context.code_address":            (REG, REG, [], False, True, AVERAGE),
    "context.meta":                    (REG, REG, [], False, True, AVERAGE),
    "context.ergs_left":               (REG, REG, [], False, True, AVERAGE),
    "context.sp":                      (REG, REG, [], False, True, AVERAGE),
    "context.get_context_u128":        (REG, REG, [], False, True, AVERAGE),
    "context.set_context_u128":        (REG, REG, [], True, False, AVERAGE),
    "context.set_ergs_per_pubdata":    (REG, REG, [], True, False, AVERAGE),
    "context.increment_tx_number":     (REG, REG, [], True, False, AVERAGE),
    "shift.shl":                       (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "shift.shr":                       (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "shift.rol":                       (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "shift.ror":                       (ANY_SRC, ANY_DST, ["set_flags", "swap"], False, True, RICH),
    "binop.xor":                       (ANY_SRC, ANY_DST, ["set_flags"], False, True, RICH),
    "binop.and":                       (ANY_SRC, ANY_DST, ["set_flags"], False, True, RICH),
    "binop.or":                        (ANY_SRC, ANY_DST, ["set_flags"], False, True, RICH),
    "ptr.add":                         (ANY_SRC, ANY_DST, ["swap"], False, True, RICH),
    "ptr.sub":                         (ANY_SRC, ANY_DST, ["swap"], False, True, RICH),
    "ptr.pack":                        (ANY_SRC, ANY_DST, ["swap"], False, True, RICH),
    "ptr.shrink":                      (ANY_SRC, ANY_DST, ["swap"], False, True, RICH),
    "near_call":                       (REG, REG, [], False, True, AVERAGE + 20),
    "log.storage_read":                (REG, REG, [], False, True, 160),
    "log.storage_write":               (REG, REG, [], False, False, 320),
    "log.to_l1_message":               (REG, REG, ["first"], True, False, 156),
    "log.event":                       (REG, REG, ["first"], True, False, 46),
    "log.precompile_call":             (REG, REG, [], True, True, 16),
    "far_call.normal":                 (REG, REG, ["static", "shard"], False, True, 182),
    "far_call.delegate":               (REG, REG, ["static", "shard"], False, True, 182),
    "far_call.mimic":                  (REG, REG, ["static", "shard"], True, True, 182),
    "ret.ok":                          (REG, REG, ["to_label"], False, True, AVERAGE),
    "ret.revert":                      (REG, REG, ["to_label"], False, True, AVERAGE),
    "ret.panic":                       (REG, REG, ["to_label"], False, True, AVERAGE),
    "uma.heap_read":                   (REG + ["imm16"], REG, ["increment"], False, True, 13),
    "uma.heap_write":                  (REG + ["imm16"], REG, ["increment"], False, True, 13),
    "uma.aux_heap_read":               (REG + ["imm16"], REG, ["increment"], False, True, 13),
    "uma.aux_heap_write":              (REG + ["imm16"], REG, ["increment"], False, True, 13),
    "uma.fat_ptr_read":                (REG, REG, ["increment"], False, True, 9),
}
CAN_WRITE_DST0_INTO_MEMORY = {"nop", "add", "sub", "mul", "div", "shift", "binop", "ptr"}


def bit_of(group, index):
    base = 0
    for name, width in GROUPS:
        if name == group:
            assert index < width
            return base + index
        base += width
    raise KeyError(group)


def expand():
    """rows in this build's enumeration order: family, variant within the family, source mode, destination mode, flag value"""
    rows = []
    position_in_family = {}
    for name in ISA:
        fam = name.split(".")[0]
        position_in_family[name] = sum(1 for other in position_in_family if other.split(".")[0] == fam)
    ordered = sorted(ISA, key=lambda nm: (FAMILY_ORDER.index(nm.split(".")[0]), position_in_family[nm]))
    for name in ordered:
        srcs, dsts, flags, kernel_only, static_ok, price = ISA[name]
        fam = name.split(".")[0]
        src_sorted = sorted(set(srcs), key=SRC_MODES.index)
        dst_sorted = sorted(set(dsts), key=DST_MODES.index)
        for src, dst, flag_value in itertools.product(src_sorted, dst_sorted, range(1 << len(flags))):
            word = 0
            word |= 1 << bit_of("type", FAMILY_ORDER.index(fam))
            word |= 1 << bit_of("variant", position_in_family[name])
            for k in range(len(flags)):
                if (flag_value >> k) & 1:
                    word |= 1 << bit_of("flag", k)
            word |= 1 << bit_of("src", SRC_MODES.index(src))
            word |= 1 << bit_of("dst", DST_MODES.index(dst))
            if kernel_only:
                word |= 1 << (AUX_BASE + AUX["kernel_only"])
            if static_ok:
                word |= 1 << (AUX_BASE + AUX["static_ok"])
            if fam == "invalid":
                word |= 1 << (AUX_BASE + AUX["explicit_panic"])
            rows.append((name, src, dst, flag_value, word, price))
    return rows


def test_the_listing_by_name_expands_to_the_blob_row_for_row():
    d = zkgl.opcode_defs_default()
    rows = expand()
    assert d.n_valid == len(rows) <= 2048
    assert (d.type_bits, d.variant_bits, d.flag_bits, d.src_mode_bits, d.dst_mode_bits) == tuple(w for _, w in GROUPS)
    assert d.description_bits_flattened == AUX_BASE
    assert (d.aux_kernel_mode, d.aux_static_ok, d.aux_explicit_panic) == (AUX["kernel_only"], AUX["static_ok"], AUX["explicit_panic"])
    for i, (name, src, dst, fl, word, price) in enumerate(rows):
        assert d.props[i] == word, (i, name, src, dst, fl, hex(d.props[i]), hex(word))
        assert d.prices[i] == price, (i, name)
    invalid_word = rows[0][4]
    for i in range(len(rows), 2048):                      # the unused rows decode as Invalid at the prohibitive price
        assert d.props[i] == invalid_word and d.prices[i] == 0xFFFFFFFF
    for fam_name in FAMILY_ORDER:
        assert d.can_write_dst0_into_memory[FAMILY_ORDER.index(fam_name)] == int(fam_name in CAN_WRITE_DST0_INTO_MEMORY), fam_name


def test_row_count_formula_and_structural_invariants():
    d = zkgl.opcode_defs_default()
    expected = sum(len(set(s)) * len(set(t)) * (1 << len(fl)) for s, t, fl, *_ in ISA.values())
    assert d.n_valid == expected
    seen = set()
    widths = [w for _, w in GROUPS]
    for i in range(d.n_valid):
        w = d.props[i]
        desc = w & ((1 << AUX_BASE) - 1)
        assert desc not in seen, i                         # a description names one row
        seen.add(desc)
        base = 0
        for (name, width) in GROUPS:
            field = (w >> base) & ((1 << width) - 1)
            if name != "flag":
                assert bin(field).count("1") == 1, (i, name)   # type / variant / source mode / destination mode are one-hot
            base += width
        assert desc >> sum(widths) == 0                    # bits 38..47 of the description are unused
        assert (w >> (AUX_BASE + 3)) == 0
    kernel = {n for n, v in ISA.items() if v[3]}
    assert kernel == {"context.set_context_u128", "context.set_ergs_per_pubdata", "context.increment_tx_number", "log.to_l1_message", "log.event",
                      "log.precompile_call", "far_call.mimic"}
    not_static = {n for n, v in ISA.items() if not v[4]}
    assert not_static == {"context.set_context_u128", "context.set_ergs_per_pubdata", "context.increment_tx_number", "log.storage_write",
                          "log.to_l1_message", "log.event"}


def test_find_round_trip_and_the_nop_and_panic_rows():
    d = zkgl.opcode_defs_default()
    rows = expand()
    position = {}
    for name in ISA:
        fam = name.split(".")[0]
        position[name] = sum(1 for other in position if other.split(".")[0] == fam)
    for i, (name, src, dst, fl, word, _) in enumerate(rows):
        if name == "invalid":
            continue
        fam = name.split(".")[0]
        assert d.find(FAMILY_ORDER.index(fam), position[name], SRC_MODES.index(src), DST_MODES.index(dst), fl) == i
    nop = next(i for i, r in enumerate(rows) if r[:4] == ("nop", "reg", "reg", 0))
    panic = next(i for i, r in enumerate(rows) if r[:4] == ("ret.panic", "reg", "reg", 0))
    assert (d.nop_encoding, d.panic_encoding) == (nop, panic)
    assert (d.nop_bitspread, d.panic_bitspread) == (rows[nop][4], rows[panic][4])
    assert F["NOP"] == FAMILY_ORDER.index("nop") and F["UMA"] == FAMILY_ORDER.index("uma") and len(F) == 16
