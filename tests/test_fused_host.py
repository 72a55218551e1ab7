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
