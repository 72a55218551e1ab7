"""Shared test helpers: circuit builders through the C ABI and witness fixtures."""
import json
import os

import numpy as np

import zkgl
from oracle import ram_native as rn
from oracle import zko

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = zko.P
G, OP, LINK = zkgl.GATE, zkgl.OP, zkgl.LINK
ALL_GATES = list(range(1, 14))


def load_fixture():
    f = json.load(open(os.path.join(GOLD, "ram_fixture.json")))
    conv = lambda lst: [rn.mq(a, rn.BOOTLOADER_HEAP_PAGE if b == "BOOTLOADER_HEAP_PAGE" else b, c, d, e, v) for a, b, c, d, e, v in lst]
    return conv(f["unsorted"]), conv(f["sorted"]), f["limit"]


_RAM_CS = {}


def ram_cs(limit):
    """recorded + finalized ram_permutation CS (cached per limit; recording needs no GPU)"""
    if limit not in _RAM_CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_ram_permutation()
        cs.ram_permutation_entry_point(limit)
        cs.pad_and_shrink()
        _RAM_CS[limit] = cs
    return _RAM_CS[limit]


def oracle_run(cs, outer, loop, batch, table_rows=65536):
    run = zko.CircuitRun(cs.export(False), cs.export(True), batch, table_rows)
    run.resolve(outer, loop)
    return run


def new_cs(cols=100, lookups=True, gates=ALL_GATES, max_trace_len=1 << 20):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(cols, 0, 8, 4), max_trace_len=max_trace_len)
    if lookups:
        cs.allow_lookup(3, 8, True)
    for g in gates:
        cs.allow_gate(g)
    return cs


class Rec:
    """Tiny recording front-end over the raw C ABI used by the op/gate unit tests: every helper
    emits the witness op AND the gate, like the C++ gadget layer does."""

    def __init__(self, cs):
        self.cs = cs
        self.n_in = 0

    def inp(self):
        v = self.cs.input(self.n_in)
        self.n_in += 1
        return v

    def const(self, c):
        return self.cs.allocate_constant(c)

    def fma(self, q, a, b, l, c):
        d = self.cs.alloc_variable_without_value()
        self.cs.emit_op(OP["FMA"], [a, b, c], [d], [q, l])
        self.cs.place_gate(G["FMA"], [a, b, c, d], [q, l])
        return d

    def lc4(self, terms, ks):
        r = self.cs.alloc_variable_without_value()
        self.cs.emit_op(OP["LC4"], terms, [r], ks)
        self.cs.place_gate(G["REDUCTION4"], terms + [r], ks)
        return r

    def select(self, s, a, b):
        r = self.cs.alloc_variable_without_value()
        self.cs.emit_op(OP["SELECT"], [s, a, b], [r])
        self.cs.place_gate(G["SELECT"], [a, b, s, r])
        return r

    def iszero(self, x):
        f, aux = self.cs.alloc_multiple_variables_without_values(2)
        self.cs.emit_op(OP["ISZERO"], [x], [f, aux])
        self.cs.place_gate(G["ZEROCHECK"], [x, aux, f])
        return f, aux

    def uadd(self, bits, a, b, cin):
        c, co = self.cs.alloc_multiple_variables_without_values(2)
        self.cs.emit_op(OP["UADD"], [a, b, cin], [c, co], a=bits)
        self.cs.place_gate(G["UINTX_ADD"], [a, b, cin, c, co], [1 << bits])
        return c, co

    def usub(self, bits, a, b, bin_):
        d, bo = self.cs.alloc_multiple_variables_without_values(2)
        self.cs.emit_op(OP["USUB"], [a, b, bin_], [d, bo], a=bits)
        self.cs.place_gate(G["UINTX_ADD"], [b, d, bin_, a, bo], [1 << bits])
        return d, bo

    def dot4(self, a, b):
        r = self.cs.alloc_variable_without_value()
        ins = [x for p in zip(a, b) for x in p]
        self.cs.emit_op(OP["DOT4"], ins, [r])
        self.cs.place_gate(G["DOT4"], ins + [r])
        return r

    def matmul(self, matrix, ins):
        outs = self.cs.alloc_multiple_variables_without_values(12)
        self.cs.emit_op(OP["MATMUL12"], ins, outs, a=matrix)
        self.cs.place_gate(G["MATMUL12_EXT"] if matrix == 0 else G["MATMUL12_INT"], ins + outs)
        return outs

    def split(self, x, n, bits, ks):
        outs = self.cs.alloc_multiple_variables_without_values(n)
        self.cs.emit_op(OP["SPLIT"], [x], outs, a=n, b=bits)
        assert n == 4
        self.cs.place_gate(G["REDUCTION4"], outs + [x], ks)
        return outs

    def u32muladd(self, a, b, c, d):
        lo, hi = self.cs.alloc_multiple_variables_without_values(2)
        self.cs.emit_op(OP["U32MULADD"], [a, b, c, d], [lo, hi])
        self.cs.place_gate(G["U32_FMA"], [a, b, c, d, lo, hi])
        return lo, hi

    def poseidon2_witness_only(self, ins):
        outs = self.cs.alloc_multiple_variables_without_values(12)
        self.cs.emit_op(OP["POSEIDON2"], ins, outs)
        return outs


def rand_fe(rng, n):
    return [int(x) for x in (rng.integers(0, 2**63, size=n, dtype=np.uint64).astype(object) * 2 + rng.integers(0, 2, size=n)) % P]


def random_instances(seed, n_inst, n_items, limit):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_inst):
        u, s, nd = rn.random_ram_witness(rng, n_items)
        out.append(rn.instance(u, s, limit, nd))
    return out


_STORAGE_CS = {}


def storage_cs(limit, enforce_permutation=True):
    key = (limit, enforce_permutation)
    if key not in _STORAGE_CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
        cs.configure_storage_validity()
        cs.sort_and_deduplicate_storage_access_entry_point(limit, enforce_permutation)
        cs.pad_and_shrink()
        _STORAGE_CS[key] = cs
    return _STORAGE_CS[key]


def load_storage_fixture():
    from oracle import storage_native as sn
    f = json.load(open(os.path.join(GOLD, "storage_fixture.json")))
    conv = lambda d: sn.log_query(address=int(d["address"]), key=int(d["key"]), read_value=int(d["read_value"]),
                                  written_value=int(d["written_value"]), rw_flag=int(d["rw_flag"]), aux_byte=int(d["aux_byte"]),
                                  rollback=int(d["rollback"]), is_service=int(d["is_service"]), shard_id=int(d["shard_id"]),
                                  tx_number_in_block=int(d["tx_number_in_block"]), timestamp=int(d["timestamp"]))
    unsorted = [conv(d) for d in f["unsorted"]]
    sorted_records = [(conv(d), int(d["record_timestamp"])) for d in f["sorted"]]
    return unsorted, sorted_records, f["limit"]


def emulated_device() -> bool:
    """True when ZKGL_LIB names the test suite's emulated-device build of the library (tests/emu/README.md): the -m gpu tests then run the device
    SOURCE on host fibers.  Test infrastructure: the product library has no such symbol and the product package does not ask."""
    import zkgl
    return hasattr(zkgl.lib(), "zk_emu_divergent_wave_sites")
