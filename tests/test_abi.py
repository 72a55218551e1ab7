"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the three headers under include/
declare (zkgl.h, zkgl_vm.h, zkgl_witness.h), returns status codes (never aborts) on misuse, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import zkgl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("zkgl.h", "zkgl_vm.h", "zkgl_witness.h")


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)                       # macros
    text = re.sub(r"typedef\s+[^;{]*\(\s*\*\s*zk_[a-z0-9_]+\s*\)[^;]*;", "", text)   # function-pointer typedefs (zk_job_fn)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,at_least", [("zkgl.h", 100), ("zkgl_vm.h", 8), ("zkgl_witness.h", 30)])
def test_library_exports_every_declared_symbol(header, at_least):
    L = zkgl.lib()
    syms = declared_symbols(header)
    assert len(syms) >= at_least, (header, len(syms))
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_headers_are_self_contained_c():
    """every header compiles on its own as plain C (what a cgo / bindgen / ctypes user feeds a C compiler)"""
    import subprocess, tempfile
    for h in HEADERS:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "t.c")
            open(src, "w").write(f'#include "{os.path.join(ROOT, "include", h)}"\nint main(void) {{ return 0; }}\n')
            subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", src], check=True)


def test_test_only_circuits_are_not_in_the_product_library():
    assert not hasattr(zkgl.lib(), "zk_test_circuit_vm_shaped") and not hasattr(zkgl.lib(), "zk_circuit_vm_shaped")
    assert hasattr(zkgl.testlib(), "zk_test_circuit_vm_shaped")


def test_round_constants_host_side_match_oracle(oracle):
    assert np.array_equal(zkgl.poseidon_round_constants(), oracle.round_constants())


def test_no_gpu_means_loud_failure_not_fallback():
    if zkgl.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(zkgl.ZkError) as e:
        zkgl.init(0)
    assert e.value.code == zkgl.ZK_ERR_HIP
    # compute entries refuse as well
    rc = zkgl.lib().zk_poseidon2_permute_aos(C.c_void_p(0), C.c_size_t(4), None)
    assert rc == zkgl.ZK_ERR_HIP
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs.configure_ram_permutation()
    cs.ram_permutation_entry_point(2)
    cs.pad_and_shrink()  # recording/placement are host work
    with pytest.raises(zkgl.ZkError) as e:
        cs.set_batch(1)  # execution is not
    assert e.value.code == zkgl.ZK_ERR_HIP


def test_misuse_returns_codes():
    with pytest.raises(zkgl.ZkError) as e:
        zkgl.ConstraintSystem(zkgl.CSGeometry(8, 0, 8, 4))  # too narrow for MatrixMultiplicationGate<12>
    assert e.value.code == zkgl.ZK_ERR_INVALID
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    with pytest.raises(zkgl.ZkError) as e:  # reference: unimplemented!() when a gate is missing (src/main_vm/utils.rs:87-89)
        cs.place_gate(zkgl.GATE["FMA"], cs.alloc_multiple_variables_without_values(4), [1, 0])
    assert e.value.code == zkgl.ZK_ERR_GATE_NOT_ALLOWED
    with pytest.raises(zkgl.ZkError) as e:  # reference: expect("table must be added before") src/main_vm/utils.rs:95-97
        cs.get_table_id_for_marker(77)
    assert "table must be added before" in e.value.msg
    cs.allow_gate(zkgl.GATE["FMA"])
    with pytest.raises(zkgl.ZkError):
        cs.place_gate(zkgl.GATE["FMA"], [0xFFFFFFFF, 0, 0, 0], [1, 0])  # Variable::placeholder()
    with pytest.raises(zkgl.ZkError):
        cs.place_gate(zkgl.GATE["FMA"], cs.alloc_multiple_variables_without_values(3), [1, 0])  # arity
    with pytest.raises(zkgl.ZkError):
        cs.allocate_constant(zkgl.P)  # non-canonical
    with pytest.raises(zkgl.ZkError):
        cs.loop_end()
    # a gate over variables nobody assigns: reference = resolver never completes; here a status code
    cs2 = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4))
    cs2.allow_gate(zkgl.GATE["FMA"])
    cs2.place_gate(zkgl.GATE["FMA"], cs2.alloc_multiple_variables_without_values(4), [1, 0])
    with pytest.raises(zkgl.ZkError) as e:
        cs2.pad_and_shrink()
    assert e.value.code == zkgl.ZK_ERR_UNRESOLVED


def test_capacity_error():
    cs = zkgl.ConstraintSystem(zkgl.CSGeometry(100, 0, 8, 4), max_trace_len=1 << 10)
    cs.configure_ram_permutation()
    cs.ram_permutation_entry_point(16)  # 16*94 + 1066 rows > 1024
    with pytest.raises(zkgl.ZkError) as e:
        cs.pad_and_shrink()
    assert e.value.code == zkgl.ZK_ERR_CAPACITY
