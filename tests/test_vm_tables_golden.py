"""Golden checks of the three self-contained VM lookup tables: every row computed here from the formulas of the reference's table
files (cited per table) and compared with the rows the recorded main_vm circuit registers (the engine's exported table words)."""
import numpy as np

import vm_programs as vp
from oracle import zko


def _tables():
    cs = vp.vm_cs(2)
    ex = zko.parse_export(cs.export(False))
    words = np.array(ex["table_words"], dtype=np.uint64)
    out = []
    for t in ex["tables"]:
        w = t["n_keys"] + t["n_vals"]
        out.append((t["n_keys"], t["n_vals"], words[t["word_off"]:t["word_off"] + t["n_rows"] * w].reshape(t["n_rows"], w) if t["n_rows"] else None))
    return out


def _find(tables, n_rows, n_keys, n_vals, nth=0):
    hits = [r for k, v, r in tables if r is not None and r.shape[0] == n_rows and k == n_keys and v == n_vals]
    return hits[nth]


def _same_rows(got, want):
    want = np.array(sorted(want), dtype=np.uint64)
    got = np.array(sorted(map(tuple, got.tolist())), dtype=np.uint64)
    return got.shape == want.shape and np.array_equal(got, want)


def test_bitshift_table():
    # /root/reference/src/tables/bitshift.rs:12-40: for shift in 0..256, idx in 0..4: [shift + (idx << 8), limb 2 idx, limb 2 idx + 1] of 1 << shift
    want = []
    for shift in range(256):
        modulus = 1 << shift
        for idx in range(4):
            y = modulus & 0xFFFFFFFF
            modulus >>= 32
            z = modulus & 0xFFFFFFFF
            modulus >>= 32
            want.append((shift + (idx << 8), y, z))
    assert _same_rows(_find(_tables(), 1024, 1, 2), want)


def test_uma_ptr_read_cleanup_table():
    # /root/reference/src/tables/uma_ptr_read_cleanup.rs:11-40: key a in 0..32 -> a == 0 ? 2^32 - 1 : 2^32 - 1 - (2^a - 1), second value 0
    full = (1 << 32) - 1
    want = [(a, full if a == 0 else full - ((1 << a) - 1), 0) for a in range(32)]
    tables = _tables()
    hits = [r for k, v, r in tables if r is not None and r.shape[0] == 32 and k == 1 and v == 2]
    assert any(_same_rows(r, want) for r in hits)


def test_integer_to_boolean_mask_tables():
    # /root/reference/src/tables/integer_to_boolean_mask.rs:21-45 create_integer_to_bitmask_table(num_bits): a -> a == 0 ? 0 : 1 << (a - 1);
    # :68-70 subpc = 2 bits; the register mask = 4 bits (REGISTER_ENCODING_BITS); :47-66 create_integer_set_ith_bit_table: a -> 1 << a
    tables = _tables()
    for bits in (4, 2):
        want = [(a, 0 if a == 0 else 1 << (a - 1), 0) for a in range(1 << bits)]
        assert _same_rows(_find(tables, 1 << bits, 1, 2), want)
    want = [(a, 1 << a, 0) for a in range(32)]
    hits = [r for k, v, r in tables if r is not None and r.shape[0] == 32 and k == 1 and v == 2]
    assert any(_same_rows(r, want) for r in hits)
