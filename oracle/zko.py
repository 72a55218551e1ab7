"""oracle/zko.py — Python face of the CPU ORACLE (test infrastructure; see oracle/zko.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It loads oracle/libzko.so (plain C, built by oracle/Makefile with gcc) and never touches the
product library.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzko.so")
P = 0xFFFFFFFF00000001

_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64 = C.c_uint64
        for name in ("zko_gl_add", "zko_gl_sub", "zko_gl_mul", "zko_gl_pow"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64]
        L.zko_gl_inv.restype = u64
        L.zko_gl_inv.argtypes = [u64]
        L.zko_poseidon_round_constants.restype = C.POINTER(u64)
        L.zko_scope_parse.restype = C.c_void_p
        L.zko_scope_parse.argtypes = [C.c_void_p, C.c_size_t]
        L.zko_scope_free.argtypes = [C.c_void_p]
        L.zko_scope_field.restype = C.c_uint32
        L.zko_scope_field.argtypes = [C.c_void_p, C.c_int]
        L.zko_scope_run.restype = C.c_int
        L.zko_scope_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint32]
        L.zko_scope_run_seq.restype = C.c_int
        L.zko_scope_run_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]
        L.zko_scope_multiplicities.restype = None
        L.zko_scope_multiplicities.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.zko_scope_check.restype = C.c_uint64
        L.zko_scope_check.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]
        L.zko_links_check.restype = C.c_uint64
        L.zko_links_check.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]
        L.zko_two_adic_root.restype = u64
        L.zko_two_adic_root.argtypes = [C.c_uint]
        L.zko_ntt_naive.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, u64]
        L.zko_ntt_batch.argtypes = [C.c_void_p, C.c_uint, C.c_size_t, C.c_size_t, C.c_int, u64]
        L.zko_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, u64]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def u64arr(x) -> np.ndarray:
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


# ---- field ----
def gl_add(a, b): return int(lib().zko_gl_add(a, b))
def gl_sub(a, b): return int(lib().zko_gl_sub(a, b))
def gl_mul(a, b): return int(lib().zko_gl_mul(a, b))
def gl_inv(a): return int(lib().zko_gl_inv(a))


def gl_fma_cols(a, b, c, q, l):
    a, b, c = u64arr(a), u64arr(b), u64arr(c)
    out = np.zeros_like(a)
    lib().zko_gl_fma_cols(_p(out), _p(a), _p(b), _p(c), C.c_uint64(q), C.c_uint64(l), C.c_size_t(a.size))
    return out


# ---- poseidon2 / sponge ----
def round_constants() -> np.ndarray:
    rc = lib().zko_poseidon_round_constants()
    return np.array([rc[i] for i in range(360)], dtype=np.uint64)


def poseidon2_permute(state):
    s = u64arr(state).copy()
    assert s.size == 12
    lib().zko_poseidon2_permute(_p(s))
    return [int(x) for x in s]


def poseidon2_permute_batch(states: np.ndarray) -> np.ndarray:
    """states: [n, 12] AoS -> permuted copy"""
    s = np.ascontiguousarray(states, dtype=np.uint64).copy()
    lib().zko_poseidon2_permute_batch(_p(s), C.c_size_t(s.shape[0]))
    return s


def mds_external(state):
    s = u64arr(state).copy(); lib().zko_poseidon2_mds_external(_p(s)); return [int(x) for x in s]


def mds_inner(state):
    s = u64arr(state).copy(); lib().zko_poseidon2_mds_inner(_p(s)); return [int(x) for x in s]


def commit_encoding(values):
    v = u64arr(values)
    out = np.zeros(4, dtype=np.uint64)
    lib().zko_commit_encoding(_p(v), C.c_size_t(v.size), _p(out))
    return [int(x) for x in out]


def fs_challenges(fs_input, reps, nchal):
    v = u64arr(fs_input)
    out = np.zeros(reps * nchal, dtype=np.uint64)
    lib().zko_fs_challenges(_p(v), C.c_size_t(v.size), _p(out), C.c_size_t(reps), C.c_size_t(nchal))
    return [[int(out[r * nchal + i]) for i in range(nchal)] for r in range(reps)]


def queue_full_push(tail, enc):
    t = u64arr(tail).copy(); e = u64arr(enc)
    lib().zko_queue_full_push(_p(t), _p(e))
    return [int(x) for x in t]


def queue_tail4_push20(tail, enc):
    t = u64arr(tail).copy(); e = u64arr(enc)
    lib().zko_queue_tail4_push20(_p(t), _p(e))
    return [int(x) for x in t]


def memory_query_encode(q13):
    q = u64arr(q13); out = np.zeros(8, dtype=np.uint64)
    lib().zko_memory_query_encode(_p(q), _p(out))
    return [int(x) for x in out]


def execution_context_encode(rec42):
    r = u64arr(rec42); out = np.zeros(32, dtype=np.uint64)
    assert r.size == 42
    lib().zko_execution_context_encode(_p(r), _p(out))
    return [int(x) for x in out]


def grand_product(enc: np.ndarray, flags, challenges, init=1):
    """enc [n, enc_len] row-major; returns running accumulator after each item"""
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    n, L = enc.shape
    fl = np.ascontiguousarray(np.array(flags, dtype=np.uint8))
    ch = u64arr(challenges)
    out = np.zeros(n, dtype=np.uint64)
    lib().zko_grand_product(_p(enc), _p(fl), _p(ch), C.c_size_t(L), C.c_size_t(n), C.c_uint64(init), _p(out))
    return out


# ---- hashes ----
def sha256_rounds_stream(a: int, state, block):
    """zko_sha256_rounds_stream: (outputs of one ZK_OP_SHA256_ROUNDS with header a, final state) for 8 + 16 little-endian byte words"""
    st = np.array(state, dtype=np.uint32); blk = np.array(block, dtype=np.uint32); out = np.zeros(32768, dtype=np.uint64)
    f = lib().zko_sha256_rounds_stream
    f.restype = C.c_size_t
    n = f(C.c_uint32(a), st.ctypes.data_as(C.c_void_p), blk.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy(), [int(x) for x in st]


def keccak256(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    lib().zko_keccak256(msg, C.c_size_t(len(msg)), out)
    return bytes(out)


def keccak_f1600(state25):
    s = u64arr(state25).copy(); lib().zko_keccak_f1600(_p(s)); return [int(x) for x in s]


def sha256_compress(state8, block: bytes):
    s = np.ascontiguousarray(np.array(state8, dtype=np.uint32))
    lib().zko_sha256_compress(_p(s), block)
    return [int(x) for x in s]


# ---- engine interpreter ----
class Scope:
    """A serialised zkgl scope (ConstraintSystem.export) loaded into the oracle interpreter."""

    def __init__(self, words: np.ndarray):
        self.words = np.ascontiguousarray(words, dtype=np.uint32)
        self.h = lib().zko_scope_parse(_p(self.words), C.c_size_t(self.words.size))
        if not self.h:
            raise ValueError("zko_scope_parse: malformed scope export")
        f = lambda i: int(lib().zko_scope_field(self.h, i))
        self.is_loop, self.n_cells, self.n_trace_cells, self.n_slots = f(0), f(1), f(2), f(3)
        self.n_input_words, self.limit, self.pre_words, self.n_prog = f(4), f(5), f(6), f(7)
        self.n_copies, self.n_links, self.n_copy_cols, self.lookup_width = f(8), f(9), f(10), f(11)

    def __del__(self):
        try:
            if self.h:
                lib().zko_scope_free(self.h)
        except Exception:
            pass


def stride_for(lanes: int) -> int:
    return (lanes + 63) // 64 * 64


class CircuitRun:
    """Run an exported circuit (outer + loop scope) on the CPU oracle for a batch of instances."""

    def __init__(self, outer_words, loop_words, batch: int, total_table_rows: int = 0):
        self.outer, self.loop = Scope(outer_words), Scope(loop_words)
        self.B = batch
        self.limit = self.loop.limit
        self.so, self.sl = stride_for(batch), stride_for(batch * max(self.limit, 0))
        self.oc = np.zeros((self.outer.n_cells, self.so), dtype=np.uint64)
        self.lc = np.zeros((max(self.loop.n_cells, 1), max(self.sl, 32)), dtype=np.uint64)
        self.mult = np.zeros(max(batch * total_table_rows, 1), dtype=np.uint32)
        self.total_rows = total_table_rows

    def resolve(self, outer_inputs: np.ndarray, loop_inputs: np.ndarray):
        """outer_inputs [words, B]; loop_inputs [words, B*limit] (lane-minor)"""
        L = lib()
        oi = np.ascontiguousarray(outer_inputs, dtype=np.uint64)
        li = np.ascontiguousarray(loop_inputs, dtype=np.uint64)
        nl = self.B * self.limit
        mult = None      # multiplicities are counted from the lookup tuples of the resolved trace, below
        rc = L.zko_scope_run(self.outer.h, 0, self.outer.pre_words, _p(self.oc), self.so, self.B, _p(oi), None, 0,
                             _p(self.lc), self.lc.shape[1], self.limit, mult, self.total_rows)
        assert rc == 0
        if self.limit:
            rc = L.zko_scope_run(self.loop.h, 0, self.loop.n_prog, _p(self.lc), self.lc.shape[1], nl, _p(li), _p(self.oc),
                                 self.so, None, 0, self.limit, mult, self.total_rows)
            assert rc == 0
        rc = L.zko_scope_run(self.outer.h, self.outer.pre_words, self.outer.n_prog, _p(self.oc), self.so, self.B, _p(oi),
                             None, 0, _p(self.lc), self.lc.shape[1], self.limit, mult, self.total_rows)
        assert rc == 0
        if self.total_rows:
            self.mult[:] = 0
            L.zko_scope_multiplicities(self.outer.h, _p(self.oc), C.c_size_t(self.so), self.B, 1, _p(self.mult), self.total_rows)
            if self.limit:
                L.zko_scope_multiplicities(self.loop.h, _p(self.lc), C.c_size_t(self.lc.shape[1]), nl, self.limit, _p(self.mult), self.total_rows)

    def seed(self, outer_inputs: np.ndarray, loop_inputs: np.ndarray) -> np.ndarray:
        """sequential seeding: returns loop_inputs with the carried words filled in (oracle twin of
        zk_cs_seed_carried_inputs)"""
        L = lib()
        oi = np.ascontiguousarray(outer_inputs, dtype=np.uint64)
        li = np.ascontiguousarray(loop_inputs, dtype=np.uint64).copy()
        rc = L.zko_scope_run(self.outer.h, 0, self.outer.pre_words, _p(self.oc), self.so, self.B, _p(oi), None, 0,
                             _p(self.lc), self.lc.shape[1], self.limit, None, 0)
        assert rc == 0
        rc = L.zko_scope_run_seq(self.loop.h, _p(self.lc), self.lc.shape[1], self.B, _p(li), _p(self.oc), self.so)
        assert rc == 0
        return li

    def check(self):
        """-> (n_violations, n_relations_evaluated)"""
        L = lib()
        first = C.c_uint64(); nrel = C.c_uint64(); total_rel = 0
        bad = int(L.zko_scope_check(self.outer.h, _p(self.oc), self.so, self.B, C.byref(first), C.byref(nrel)))
        total_rel += nrel.value
        if self.limit:
            bad += int(L.zko_scope_check(self.loop.h, _p(self.lc), self.lc.shape[1], self.B * self.limit, C.byref(first), C.byref(nrel)))
            total_rel += nrel.value
            bad += int(L.zko_links_check(self.loop.h, _p(self.lc), self.lc.shape[1], self.B * self.limit, _p(self.oc), self.so))
        return bad, total_rel


# ---- K10 lookup-argument oracle (pure Python integers; CPU ORACLE, test infrastructure) ----
def parse_export(words) -> dict:
    """sections of a serialised scope (ConstraintSystem.export): header fields, lookup rows, table descriptors / words"""
    w = [int(x) for x in np.asarray(words, dtype=np.uint32)]
    assert w[0] == 0x5a4b4733
    h = dict(is_loop=w[1], n_cells=w[2], n_trace_cells=w[3], n_slots=w[4], n_copy_cols=w[5], lookup_width=w[6], n_input_words=w[7],
             limit=w[8], pre_words=w[9], n_prog=w[10], n_consts=w[11], n_rows=w[12], n_rowconsts=w[13], n_lrows=w[14], n_copies=w[15],
             n_tables=w[16], n_table_words=w[17], n_links=w[18], n_carries=w[19], n_stream_words=w[20])
    p = 21 + h["n_prog"] + 2 * h["n_consts"] + 4 * h["n_rows"] + 2 * h["n_rowconsts"]
    q = 21 + h["n_prog"] + 2 * h["n_consts"]
    h["rows"] = [tuple(w[q + 4 * i: q + 4 * i + 4]) for i in range(h["n_rows"])]  # (kind, n_instances, const_off, n_consts)
    h["lrows"] = [(w[p + 2 * i], w[p + 2 * i + 1]) for i in range(h["n_lrows"])]
    p += 2 * h["n_lrows"]
    h["copies"] = [(w[p + 2 * i], w[p + 2 * i + 1]) for i in range(h["n_copies"])]  # (cell, partner cell)
    p += 2 * h["n_copies"]
    h["tables"] = [dict(word_off=w[p + 9 * i], mult_off=w[p + 9 * i + 1], n_rows=w[p + 9 * i + 2], n_keys=w[p + 9 * i + 3], n_vals=w[p + 9 * i + 4])
                   for i in range(h["n_tables"])]
    p += 9 * h["n_tables"]
    h["table_words"] = [w[p + 2 * i] | (w[p + 2 * i + 1] << 32) for i in range(h["n_table_words"])]
    p += 2 * h["n_table_words"]
    h["links"] = [tuple(w[p + 4 * i: p + 4 * i + 3]) for i in range(h["n_links"])]  # (kind, loop cell, other cell)
    p += 4 * h["n_links"] + 4 * h["n_carries"]
    h["streams"] = []
    end = p + h["n_stream_words"]
    while p < end:
        pa, pb, n_total = w[p: p + 3]
        h["streams"].append((w[p + 3: p + 3 + pa], w[p + 3 + pa: p + 3 + pa + pb], n_total))
        p += 3 + pa + pb
    return h


def e2_mul(x, y):
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def e2_inv(x):
    n = pow((x[0] * x[0] - 7 * x[1] * x[1]) % P, P - 2, P)
    return (x[0] * n % P, (P - x[1]) * n % P)


def lookup_argument(run: "CircuitRun", outer_words, loop_words, beta, gamma, n_cols: int):
    """K10 restatement (csrc/kernels_lookup_arg.hpp): per instance (A, B) with A = sum over every lookup tuple of the trace of
    1/f, B = sum over table rows of multiplicity/f, f = beta + sum_j gamma^j c_j + gamma^W table (W = lookup width)."""
    ho = parse_export(outer_words)
    Wd = ho["lookup_width"]
    gp = [(1, 0), tuple(gamma)]
    while len(gp) <= Wd:
        gp.append(e2_mul(gp[-1], gamma))

    def f(c, t):
        c = list(c) + [0] * (Wd - len(c)) + [t]                 # the table id takes the power gamma^W
        return ((beta[0] + sum(gp[j][0] * c[j] for j in range(Wd + 1))) % P, (beta[1] + sum(gp[j][1] * c[j] for j in range(Wd + 1))) % P)

    def add(x, y):
        return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)

    out = []
    hl = parse_export(loop_words) if run.limit else None
    for inst in range(run.B):
        A = (0, 0)
        for h, cells, lanes in ((ho, run.oc, [inst]), (hl, run.lc, range(inst * run.limit, (inst + 1) * run.limit))):
            if h is None:
                continue
            W, C0 = h["lookup_width"], h["n_copy_cols"]
            for lane in lanes:
                for slot, (table, n) in enumerate(h["lrows"]):
                    if table == 0xFFFFFFFF:
                        continue
                    for u in range(n):
                        c0 = slot * n_cols + C0 + u * W
                        A = add(A, e2_inv(f([int(cells[c0 + j, lane]) for j in range(W)], table)))
        B = (0, 0)
        total = run.total_rows
        for t, td in enumerate(ho["tables"]):
            if t == 0:
                continue
            w = td["n_keys"] + td["n_vals"]
            for r in range(td["n_rows"]):
                m = int(run.mult[inst * total + td["mult_off"] + r])
                if m:
                    row = ho["table_words"][td["word_off"] + r * w: td["word_off"] + (r + 1) * w]
                    i = e2_inv(f(row[:Wd], t))
                    B = add(B, (i[0] * m % P, i[1] * m % P))
        out.append(A + B)
    return out


# ---- K11: NTT / coset LDE (zko_ntt.c) ----
def two_adic_root(log_n: int) -> int:
    return int(lib().zko_two_adic_root(log_n))


def ntt_naive(a: np.ndarray, shift: int = 1) -> np.ndarray:
    """the definition: out[bitrev(k)] = sum_i a[i] (shift * omega^k)^i, by Horner at every point"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().zko_ntt_naive(a.ctypes.data, out.ctypes.data, int(a.size).bit_length() - 1, shift)
    return out


def ntt(a: np.ndarray, inverse: bool = False, shift: int = 1) -> np.ndarray:
    """a: [n_polys, N] (or [N]); returns the transformed copy"""
    a = np.array(a, dtype=np.uint64, order="C", copy=True)
    v = a.reshape(1, -1) if a.ndim == 1 else a
    lib().zko_ntt_batch(v.ctypes.data, int(v.shape[1]).bit_length() - 1, v.shape[0], v.shape[1], int(inverse), shift)
    return a


def lde(coeffs: np.ndarray, log_blowup: int, shift: int = 1) -> np.ndarray:
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = np.empty(coeffs.size << log_blowup, dtype=np.uint64)
    lib().zko_lde(coeffs.ctypes.data, out.ctypes.data, int(coeffs.size).bit_length() - 1, log_blowup, shift)
    return out


# ---- K12: copy-permutation grand product (pure Python integers; CPU ORACLE, test infrastructure) ----
GATE_WIDTH = [0, 1, 1, 4, 5, 4, 3, 5, 9, 24, 24, 1, 6, 5]  # columns per gate instance, by zk_gate_kind
LINK_CARRY, LINK_FIRST, LINK_LAST, LINK_BCAST = 0, 1, 2, 3


def copy_classes(ho: dict, hl: dict, limit: int):
    """union-find over the trace-cell labels (outer cell c -> c; loop cell c of iteration k -> NT_outer + k NT_loop + c) from the
    exported structure alone: copy pairs of both scopes, links, stream links.  Returns find()."""
    nto, ntl = ho["n_trace_cells"], hl["n_trace_cells"] if limit else 0
    parent = list(range(nto + limit * ntl))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    def join(a, b):
        parent[find(a)] = find(b)

    lab = lambda k, c: nto + k * ntl + c
    for c, partner in ho["copies"]:
        join(c, partner)
    for k in range(limit):
        for c, partner in hl["copies"]:
            join(lab(k, c), lab(k, partner))
    for kind, lc, oc in (hl["links"] if limit else []):
        if lc >= ntl or oc >= (ntl if kind == LINK_CARRY else nto):
            continue  # a variable outside the trace (no gate references it)
        if kind == LINK_CARRY:
            for k in range(1, limit):
                join(lab(k, lc), lab(k - 1, oc))
        elif kind == LINK_FIRST:
            join(lab(0, lc), oc)
        elif kind == LINK_LAST:
            join(lab(limit - 1, lc), oc)
        else:
            for k in range(limit):
                join(lab(k, lc), oc)
    for a, b, n_total in (hl["streams"] if limit else []):
        if any(c >= ntl for c in list(a) + list(b)):
            continue
        for i in range(n_total):
            join(lab(i // len(a), a[i % len(a)]), lab(i // len(b), b[i % len(b)]))
    return find


def sigma_matches_classes(sigma, ho: dict, hl: dict, limit: int) -> bool:
    """sigma (list over all labels) is a permutation whose cycles are exactly the copy classes"""
    n = len(sigma)
    if sorted(int(x) for x in sigma) != list(range(n)):
        return False
    find = copy_classes(ho, hl, limit)
    cyc = [-1] * n
    for start in range(n):
        if cyc[start] >= 0:
            continue
        x = start
        while cyc[x] < 0:
            cyc[x] = start
            x = int(sigma[x])
    rep_of_cycle, cycle_of_rep = {}, {}
    for x in range(n):
        r = find(x)
        if rep_of_cycle.setdefault(cyc[x], r) != r or cycle_of_rep.setdefault(r, cyc[x]) != cyc[x]:
            return False
    return True


def copy_permutation_z(ho: dict, hl: dict, limit: int, outer_col, loop_cols, sigma, beta, gamma, n_cols: int):
    """z over the rows of one instance.  outer_col: values of the outer trace cells; loop_cols[k]: values of iteration k;
    sigma: list over all labels.  Returns [(a, b)] of length rows + 1."""
    nto, ntl = ho["n_trace_cells"], hl["n_trace_cells"] if limit else 0
    z = [(1, 0)]

    def rows_of(h, values, base):
        W, C0 = h["lookup_width"], h["n_copy_cols"]
        for slot in range(h["n_slots"]):
            kind, ninst = h["rows"][slot][0], h["rows"][slot][1]
            cols = list(range(ninst * GATE_WIDTH[kind])) + [C0 + i for i in range(h["lrows"][slot][1] * W)]
            num, den = (1, 0), (1, 0)
            for col in cols:
                cell = slot * n_cols + col
                w, lab = int(values[cell]), base + cell
                tn = ((w + beta[0] * lab + gamma[0]) % P, (beta[1] * lab + gamma[1]) % P)
                sg = int(sigma[lab])
                td = ((w + beta[0] * sg + gamma[0]) % P, (beta[1] * sg + gamma[1]) % P)
                num, den = e2_mul(num, tn), e2_mul(den, td)
            z.append(e2_mul(z[-1], e2_mul(num, e2_inv(den))))

    for k in range(limit):
        rows_of(hl, loop_cols[k], nto + k * ntl)
    rows_of(ho, outer_col, 0)
    return z
