"""The cone seeding program (CS::build_seed_program): a regression guard on its size per circuit — the backward slice must stay
small for the queue circuits and the seed hints must keep the hash decompositions out of it — plus structural checks that need
no GPU (every circuit with carried state gets a cone program; slot budget respected)."""
import pytest

import zkgl

CASES = [
    # (name, configure, entry, geometry cols, max seed ops, max seed slots)
    ("ram_permutation", "configure_ram_permutation", lambda c: c.ram_permutation_entry_point(8), 100, 200, 128),
    ("storage_validity", "configure_storage_validity", lambda c: c.sort_and_deduplicate_storage_access_entry_point(8, True), 100, 600, 256),
    ("log_sorter", "configure_log_sorter", lambda c: c.sort_and_deduplicate_events_entry_point(8), 100, 450, 256),
    ("demux_log_queue", "configure_demux_log_queue", lambda c: c.demultiplex_storage_logs_entry_point(8), 100, 400, 256),
    ("sort_decommits", "configure_sort_decommits", lambda c: c.sort_and_deduplicate_code_decommittments_entry_point(8), 100, 400, 256),
    ("sha256_round_function", "configure_sha256", lambda c: c.sha256_round_function_entry_point(4), 100, 400, 256),
    ("code_unpacker", "configure_code_unpacker", lambda c: c.unpack_code_into_memory_entry_point(4), 100, 400, 256),
    ("keccak256_blocks", "configure_keccak", lambda c: c.keccak256_blocks_entry_point(4), 100, 400, 640),
    ("eip_4844", "configure_eip_4844", lambda c: c.eip_4844_entry_point(27), 60, 1000, 640),
    ("vm_shaped", "configure_vm_shaped", lambda c: c.vm_shaped_entry_point(6), 140, 1100, 400),
]


@pytest.mark.parametrize("name,configure,entry,cols,max_ops,max_slots", CASES, ids=[c[0] for c in CASES])
def test_cone_program_is_small(name, configure, entry, cols, max_ops, max_slots):
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(cols, 0, 8, 8 if name == "vm_shaped" else 4), 1 << 22, 1 << 28)
    getattr(cs, configure)()
    entry(cs)
    cs.pad_and_shrink()
    st = cs.stats()
    assert 0 < st["seed_ops"] <= max_ops, st["seed_ops"]
    assert 0 < st["seed_slots"] <= max_slots, st["seed_slots"]
    assert st["seed_ops"] < st["loop_ops"]
    assert st["seed_slots"] + cs.input_words()[1] <= 5120      # LDS budget of k_seed_cone (kernels_engine.hpp SEED_LDS_WORDS)


def test_hash_fsm_keeps_its_byte_buffer_in_the_cone():
    """keccak FSM: the Keccak-f decomposition is replaced by the hint, the ByteBuffer selects are genuine carried state"""
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_keccak()
    cs.keccak256_round_function_entry_point(2)
    cs.pad_and_shrink()
    st = cs.stats()
    # (round 4: the loop program carries Keccak-f as ONE macro-op, ZK_OP_KECCAK_F, so it is barely longer than the cone; what is left in
    # both is the ByteBuffer muxing)
    assert 40000 < st["seed_ops"] < 50000 and st["loop_ops"] > st["seed_ops"]


# ---- gated witness-only permutations in the cone: the cone runs ZK_OP_POSEIDON2 a = 1 ungated, which is right only when its
# outputs reach the carried words through a select on the op's own flag — verified at finalize, not assumed
def _gated_chain(select_on_flag: bool):
    from helpers import LINK, Rec
    from zkgl import GATE as G, OP
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4))
    for k in ("CONST", "BOOLEAN", "FMA", "SELECT", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    start = r.inp()                                       # outer word 0: the chain's initial value
    cs.loop_begin(5)
    r.n_in = 0
    acc_in = r.inp()                                      # loop word 0: carried
    cs.link(LINK["FIRST"], acc_in, start)
    flag = r.inp()                                        # loop word 1: `execute`
    cs.place_gate(G["BOOLEAN"], [flag])
    one = r.const(1)
    state = [acc_in] + [one] * 11
    outs = cs.alloc_multiple_variables_without_values(12)
    cs.emit_op(OP["POSEIDON2"], state + [flag], outs, a=1)   # simulate_round_function(cs, state, execute): zeros where the flag is off
    nxt = r.select(flag, outs[0], acc_in) if select_on_flag else r.fma(1, outs[0], one, 1, acc_in)
    cs.link(LINK["CARRY"], acc_in, nxt)
    cs.loop_end()
    cs.place_gate(G["PUBLIC_INPUT"], [cs.loop_last(nxt)])
    cs.pad_and_shrink()
    return cs


# (GPU test of the two circuits above: tests/test_zz_round5_gpu.py)


def _two_gated_permutations(flags_differ: bool, const_flag: bool = False):
    """A gated by f feeds B gated by g, then select(g, B.out, acc): with g = 1, f = 0 the ungated cone would compute P(P(in)) where the trace
    holds P(0 ..): the taint of B's state inputs must survive B.  Same flag twice is the reference's own chain shape (the
    3-permutation log push under one `execute`, /root/reference/src/main_vm/opcodes/log.rs:532-595) and stays clean."""
    from helpers import LINK, Rec
    from zkgl import GATE as G, OP
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(60, 0, 8, 4))
    for k in ("CONST", "BOOLEAN", "FMA", "SELECT", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    start = r.inp()
    cs.loop_begin(5)
    r.n_in = 0
    acc_in = r.inp()
    cs.link(LINK["FIRST"], acc_in, start)
    f = r.inp()
    g = r.inp() if flags_differ else f
    cs.place_gate(G["BOOLEAN"], [f])
    if flags_differ:
        cs.place_gate(G["BOOLEAN"], [g])
    one = r.const(1)
    a_out = cs.alloc_multiple_variables_without_values(12)
    cs.emit_op(OP["POSEIDON2"], [acc_in] + [one] * 11 + [one if const_flag else f], a_out, a=1)
    b_out = cs.alloc_multiple_variables_without_values(12)
    cs.emit_op(OP["POSEIDON2"], list(a_out) + [g], b_out, a=1)
    nxt = r.select(g, b_out[0], acc_in)
    cs.link(LINK["CARRY"], acc_in, nxt)
    cs.loop_end()
    cs.place_gate(G["PUBLIC_INPUT"], [cs.loop_last(nxt)])
    cs.pad_and_shrink()
    return cs


def test_cone_taint_survives_a_gated_permutation_under_another_flag():
    assert _gated_chain(True).stats()["seed_cone_unsupported"] == 0
    assert _gated_chain(False).stats()["seed_cone_unsupported"] == 1
    assert _two_gated_permutations(False).stats()["seed_cone_unsupported"] == 0      # one flag: wrong exactly where the select looks away
    assert _two_gated_permutations(True).stats()["seed_cone_unsupported"] == 1       # two flags: g on, f off reads P(P(in)) for P(0)
    assert _two_gated_permutations(False, const_flag=True).stats()["seed_cone_unsupported"] == 1   # a flag that is no variable: not reasoned about
