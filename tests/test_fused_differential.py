"""Differential test of the two evaluation modes of zk_cs_resolve_and_check against the CPU checker (round-3 VERDICT, task 1b).

Random programs made of the constructs whose soundness rests on a gate the witness op does NOT imply — bit and nibble decompositions
of raw inputs (SPLIT + BOOLEAN / lookup range checks + recomposition), selects under unconstrained selectors, integer add / multiply
relations over raw operands, enforcements between given variables — are run on ADVERSARIAL inputs (out-of-range words, non-boolean
flags, values near p, broken equalities).  For every case:

    verdict(fused) == verdict(ZK_CHECK_STORED: every gate from stored values) == verdict(oracle checker on the oracle's own witness)

10 080 cases on the GPU (24 programs x 420 input vectors, one instance per call through a window of one resident stream) plus the
same batches in one call each (the first failing instance must be the first bad case); the oracle leg alone runs on CPU for a subset.
"""
import numpy as np
import pytest

import zkgl
from helpers import LINK, Rec, new_cs
from oracle import zko

P = zko.P
G, OP = zkgl.GATE, zkgl.OP

N_PROGRAMS = 24
CASES_PER_PROGRAM = 420


class Prog:
    pass


def hazard_circuit(seed, n_ops=60, with_loop=False):
    """-> Prog: cs, outer input kinds, loop input kinds (kind = how a GOOD value of that raw input looks)"""
    rng = np.random.default_rng(1000 + seed)
    cs = new_cs(cols=int(rng.choice([40, 64])), max_trace_len=1 << 20)
    xor4 = np.array([[a, b, a ^ b] for a in range(16) for b in range(16)], dtype=np.uint64)
    t_xor = cs.add_lookup_table(901, 2, 1, xor4)
    r = Rec(cs)
    kinds = []            # per raw input of the scope being recorded: ("bits", n) | ("bool",) | ("field",) | ("eq", other_input_index)

    def raw(kind):
        kinds.append(kind)
        return r.inp()

    def body(pool, n):
        for _ in range(n):
            k = int(rng.integers(0, 9))
            pick = lambda: pool[int(rng.integers(0, len(pool)))]
            if k == 0:      # spread_into_bits of a raw input: SPLIT(4, 1) + recomposition + BOOLEAN per output
                x = raw(("bits", 4))
                bits = r.split(x, 4, 1, [1, 2, 4, 8])
                for b in bits:
                    cs.place_gate(G["BOOLEAN"], [b])
                pool.extend(bits)
            elif k == 1:    # nibble decomposition of a raw 16-bit input, every nibble range-checked by a lookup
                x = raw(("bits", 16))
                nib = r.split(x, 4, 4, [1, 16, 256, 4096])
                for q in nib:
                    (z,) = cs.perform_lookup(t_xor, [q, q], 1)
                    pool.append(z)
                pool.extend(nib)
            elif k == 2:    # select under a raw selector; the circuit constrains it to 0 / 1 only half of the time
                s = raw(("bool",))
                if rng.integers(0, 2):
                    cs.place_gate(G["BOOLEAN"], [s])
                pool.append(r.select(s, pick(), pick()))
            elif k == 3:    # 4-bit add / sub over raw operands that only a lookup keeps in range
                a, b = raw(("bits", 4)), raw(("bits", 4))
                (z,) = cs.perform_lookup(t_xor, [a, b], 1)
                c, co = r.uadd(4, a, b, r.const(0))
                d, bo = r.usub(4, a, b, r.const(0))
                pool.extend([z, c, co, d, bo])
            elif k == 4:    # u32 multiply-add over raw operands (no range checks recorded: the relation is all there is)
                a, b = raw(("bits", 32)), raw(("bits", 32))
                lo, hi = r.u32muladd(a, b, a, b)
                pool.extend([lo, hi])
            elif k == 5:    # enforce_equal between two given variables (an FMA whose output is given)
                a = raw(("field",))
                b = raw(("eq", len(kinds) - 1))
                cs.place_gate(G["FMA"], [a, r.const(1), a, b], [1, 0])
                pool.append(b)
            elif k == 6:    # is_zero + its flag used as a selector + BOOLEAN on the flag (mirrored: the op yields 0 / 1)
                f, _ = r.iszero(pick())
                cs.place_gate(G["BOOLEAN"], [f])
                pool.append(r.select(f, pick(), pick()))
            elif k == 7:    # 2-chunk split (low bits + residual) of a raw input with a lookup on both halves
                x = raw(("bits", 8))
                outs = cs.alloc_multiple_variables_without_values(2)
                cs.emit_op(OP["SPLIT"], [x], outs, a=2, b=4)
                cs.place_gate(G["REDUCTION4"], [outs[0], outs[1], r.const(0), r.const(0), x], [1, 16, 0, 0])
                (z,) = cs.perform_lookup(t_xor, [outs[0], outs[1]], 1)
                pool.extend([outs[0], outs[1], z])
            else:
                pool.append(r.fma(int(rng.integers(1, 1 << 62)), pick(), pick(), int(rng.integers(0, 1 << 62)), pick()))

    pr = Prog()
    pool = [r.const(1), raw(("field",)), raw(("field",))]
    body(pool, n_ops)
    pr.outer_kinds = list(kinds)
    pr.limit = 0
    pr.loop_kinds = []
    if with_loop:
        first = pool[-1]
        pr.limit = 3
        cs.loop_begin(pr.limit)
        r.n_in = 0
        kinds.clear()
        acc_in = raw(("carried",))
        cs.link(LINK["FIRST"], acc_in, first)
        lpool = [acc_in, r.const(1), cs.loop_import(pool[1]), raw(("field",))]
        body(lpool, n_ops)
        acc_out = r.fma(1, lpool[-1], lpool[-2], 1, acc_in)
        cs.link(LINK["CARRY"], acc_in, acc_out)
        pr.loop_kinds = list(kinds)
        cs.loop_end()
        pool.append(cs.loop_last(acc_out))
    cs.place_gate(G["PUBLIC_INPUT"], [r.fma(1, pool[-1], pool[-2], 1, pool[1])])
    cs.pad_and_shrink()
    pr.cs = cs
    return pr


ADVERSARIAL = [2, 3, 16, 17, 255, 256, 1 << 16, (1 << 16) + 1, (1 << 32) - 1, 1 << 32, (1 << 32) + 5, 1 << 63, P - 1, P - 2, P >> 1]


def make_inputs(seed, kinds, lanes, adversarial_rate):
    """[words, lanes] u64: good values by kind, then a share of the lanes gets one or two words replaced by adversarial values"""
    rng = np.random.default_rng(7000 + seed)
    w = np.zeros((len(kinds), lanes), dtype=np.uint64)
    for i, k in enumerate(kinds):
        if k[0] == "bits":
            w[i] = rng.integers(0, 1 << k[1], lanes, dtype=np.uint64)
        elif k[0] == "bool":
            w[i] = rng.integers(0, 2, lanes, dtype=np.uint64)
        elif k[0] == "field":
            w[i] = rng.integers(0, 1 << 63, lanes, dtype=np.uint64) % np.uint64(P)
        elif k[0] == "eq":
            w[i] = w[k[1]]
        # "carried": seeded
    touched = np.zeros(lanes, dtype=bool)
    cand = [i for i, k in enumerate(kinds) if k[0] != "carried"]
    for lane in range(lanes):
        if rng.random() >= adversarial_rate or not cand:
            continue
        touched[lane] = True
        for _ in range(int(rng.integers(1, 3))):
            i = cand[int(rng.integers(0, len(cand)))]
            k = kinds[i]
            mode = int(rng.integers(0, 4))
            if mode == 0 and k[0] == "bits":
                v = (1 << k[1]) + int(rng.integers(0, 3))                 # just past the range
            elif mode == 1 and k[0] == "bits" and k[1] < 63:
                v = int(rng.integers(1 << k[1], 1 << min(62, k[1] + 8)))   # a few bits too many
            elif mode == 2:
                v = int(w[i, lane]) ^ 1                                   # off by one bit: breaks equalities, flips flags (stays in range for most kinds)
            else:
                v = ADVERSARIAL[int(rng.integers(0, len(ADVERSARIAL)))]
            w[i, lane] = np.uint64(v % P)
    return w, touched


def oracle_verdicts(pr, outer, loop):
    """per-case verdict of the CPU checker on the CPU interpreter's own witness (one instance per run)"""
    eo, el = pr.cs.export(False), pr.cs.export(True)
    out = []
    for c in range(outer.shape[1]):
        lo = loop[:, c * pr.limit:(c + 1) * pr.limit] if pr.limit else np.zeros((0, 0), dtype=np.uint64)
        run = zko.CircuitRun(eo, el, 1, 256)
        oc = np.ascontiguousarray(outer[:, c:c + 1])
        if pr.limit:
            lo = run.seed(oc, np.ascontiguousarray(lo))
            run = zko.CircuitRun(eo, el, 1, 256)
        run.resolve(oc, lo)
        out.append(run.check()[0] == 0)
    return np.array(out)


def program_and_cases(p, n_cases):
    pr = hazard_circuit(p, n_ops=40 + 7 * (p % 5), with_loop=(p % 3 == 2))
    outer, t_o = make_inputs(p, pr.outer_kinds, n_cases, 0.6 if not pr.limit else 0.35)
    loop = np.zeros((max(len(pr.loop_kinds), 1), n_cases * max(pr.limit, 1)), dtype=np.uint64)
    if pr.limit:
        loop, t_l = make_inputs(p + 500, pr.loop_kinds, n_cases * pr.limit, 0.12)
    return pr, outer, loop


@pytest.mark.parametrize("p", [0, 2, 5])
def test_oracle_rejects_exactly_the_adversarial_cases_it_should(p):
    """CPU leg: the generator produces both verdicts, and an untouched case is always satisfied"""
    pr, outer, loop = program_and_cases(p, 40)
    v = oracle_verdicts(pr, outer, loop)
    assert v.any() and not v.all()
    good_outer, _ = make_inputs(p, pr.outer_kinds, 40, 0.0)
    good_loop = loop
    if pr.limit:
        good_loop, _ = make_inputs(p + 500, pr.loop_kinds, 40 * pr.limit, 0.0)
    assert oracle_verdicts(pr, good_outer, good_loop).all()


@pytest.mark.gpu
@pytest.mark.parametrize("p", list(range(N_PROGRAMS)))
def test_fused_equals_stored_equals_oracle(zk, p, monkeypatch):
    monkeypatch.delenv("ZKGL_VERIFY_STORED", raising=False)
    n = CASES_PER_PROGRAM
    pr, outer, loop = program_and_cases(p, n)
    cs = pr.cs
    want = oracle_verdicts(pr, outer, loop)
    assert want.any() and not want.all(), "the generator must produce both verdicts"
    d_o = zk.DeviceBuffer.from_numpy(outer)
    # the loop stream is seeded once for all cases (carried accumulator), like a host would
    d_l = zk.DeviceBuffer.from_numpy(loop)
    if pr.limit:
        cs.seed_stream(n, d_o, d_l)
    got = {}
    # (a) one case per call: a window of one instance over the resident streams
    cs.set_batch(1)
    for stored in (False, True):
        cs.set_check_mode(stored)
        v = np.zeros(n, dtype=bool)
        for c in range(n):
            cs.bind_inputs(False, d_o, outer.shape[0], lane_stride=n, lane_offset=c)
            if pr.limit:
                cs.bind_inputs(True, d_l, loop.shape[0], lane_stride=n * pr.limit, lane_offset=c * pr.limit)
            ok, f = cs.resolve_and_check()
            v[c] = ok
        got[stored] = v
    assert np.array_equal(got[False], got[True]), ("fused and stored verdicts differ", np.nonzero(got[False] != got[True])[0][:10])
    assert np.array_equal(got[False], want), ("device and oracle verdicts differ", np.nonzero(got[False] != want)[0][:10])
    # (b) all cases in one batch: unsatisfied, and the failure names the first bad case; strand and plain forms of the programs
    first_bad = int(np.nonzero(~want)[0][0])
    bad_outer = np.nonzero(~oracle_scope_ok(pr, outer, loop))[0]
    cs.set_batch(n)
    cs.bind_inputs(False, d_o, outer.shape[0])
    if pr.limit:
        cs.bind_inputs(True, d_l, loop.shape[0])
    for strands in ("0", "1"):
        monkeypatch.setenv("ZKGL_STRANDS", strands)
        for stored in (False, True):
            cs.set_check_mode(stored)
            ok, f = cs.resolve_and_check()
            assert not ok
            # outer-scope failures are reported before loop-scope ones: compare instances only when the scopes agree
            if f.scope == 0 and len(bad_outer):
                assert f.instance == int(bad_outer[0]), (strands, stored, f)
            else:
                assert not want[f.instance], (strands, stored, f)
            assert f.instance >= first_bad
    cs.set_check_mode(False)


def oracle_scope_ok(pr, outer, loop):
    """which cases satisfy the outer scope's gates (the device reports outer-scope failures first)"""
    import ctypes as C
    L = zko.lib()
    eo, el = pr.cs.export(False), pr.cs.export(True)
    out = []
    for c in range(outer.shape[1]):
        run = zko.CircuitRun(eo, el, 1, 256)
        oc = np.ascontiguousarray(outer[:, c:c + 1])
        lo = np.zeros((0, 0), dtype=np.uint64)
        if pr.limit:
            lo = run.seed(oc, np.ascontiguousarray(loop[:, c * pr.limit:(c + 1) * pr.limit]))
            run = zko.CircuitRun(eo, el, 1, 256)
        run.resolve(oc, lo)
        first = C.c_uint64(); nrel = C.c_uint64()
        bad = int(L.zko_scope_check(run.outer.h, zko._p(run.oc), run.so, 1, C.byref(first), C.byref(nrel)))
        out.append(bad == 0)
    return np.array(out)
