"""Host-side view (no GPU) of what the fused mode of zk_cs_resolve_and_check leaves to the check kernels: zk_stats splits
constraints_per_instance into the relations evaluated from stored values and the relations evaluated by the witness kernels."""
import zkgl
from helpers import Rec
from zkgl import GATE as G


def test_residual_bit_of_a_one_bit_split_stays_in_the_check_program():
    n = 4
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(40, 0, 8, 4))
    for k in ("CONST", "BOOLEAN", "REDUCTION4", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    x = r.inp()
    bits = r.split(x, n, 1, [1 << i for i in range(n)])
    for b in bits:
        cs.place_gate(G["BOOLEAN"], [b])
    cs.place_gate(G["PUBLIC_INPUT"], [bits[-1]])
    cs.pad_and_shrink()
    st = cs.stats()
    # REDUCTION4 (binding: its output x is an input) + the BOOLEAN of the residual chunk are read from the store; the three masked
    # bits are 0 / 1 whatever x is
    assert st["constraints_per_instance"] == st["constraints_from_store_fused"] + st["constraints_in_witness_fused"]
    assert st["constraints_in_witness_fused"] == n - 1
    assert st["constraints_from_store_fused"] >= 2


def test_main_vm_split_between_store_and_witness():
    import vm_programs as vp
    st = vp.vm_cs(16).stats()
    assert st["constraints_per_instance"] == st["constraints_from_store_fused"] + st["constraints_in_witness_fused"]
    assert 0 < st["constraints_from_store_fused"] < st["constraints_in_witness_fused"]


def test_census_of_values_bounded_by_the_constraints():
    """zk_stats.values_below_2_32_*: variables that are < 2^32 in every satisfying witness, from the constraints alone (CS::bound_values)"""
    n = 4
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(40, 0, 8, 4))
    for k in ("CONST", "BOOLEAN", "FMA", "REDUCTION4", "SELECT", "PUBLIC_INPUT"):
        cs.allow_gate(G[k])
    r = Rec(cs)
    x, y = r.inp(), r.inp()                       # unconstrained inputs: no bound
    bits = [r.inp() for _ in range(n)]
    for b in bits:
        cs.place_gate(G["BOOLEAN"], [b])          # 4 booleans
    v = r.lc4(bits, [1, 2, 4, 8])                 # < 16: a non-wrapping reduction of bounded terms
    w = r.lc4([x, bits[0], bits[1], bits[2]], [1, 1, 1, 1])   # x is unbounded -> so is w
    s = r.select(bits[0], v, bits[1])             # boolean selector, both branches bounded
    t = r.select(x, v, bits[1])                   # selector not constrained to 0 / 1 -> unbounded
    big = r.fma(1 << 40, v, v, 0, y)              # 2^40 * 15 * 15 > 2^32 (bounded, but not narrow)
    cs.place_gate(G["PUBLIC_INPUT"], [w]); cs.place_gate(G["PUBLIC_INPUT"], [s]); cs.place_gate(G["PUBLIC_INPUT"], [t]); cs.place_gate(G["PUBLIC_INPUT"], [big])
    cs.pad_and_shrink()
    st = cs.stats()
    assert st["values_below_2_32_outer"] == n + 2   # the bits, v, s   (constants aside: none allocated)
    import vm_programs as vp
    st = vp.vm_cs(16).stats()
    assert 0.25 * st["cells_written_loop"] < st["values_below_2_32_loop"] < 0.5 * st["cells_written_loop"]
