"""Synthetic zkEVM programs + stream packing for the main_vm tests (tests/test_main_vm_host.py, the GPU twins in
tests/test_gpu_main_vm.py, bench.py's workload).  The programs are this build's own (the reference crate has no main_vm test,
SURVEY.md §4); together they execute every one of the eleven opcode families of /root/reference/src/main_vm/opcodes/ at least
once, including far_call / ret (ok, revert, panic) / UMA (aligned, unaligned, fat-pointer) / log (storage, event, L1, precompile)."""
import numpy as np

import zkgl
from oracle import main_vm_native as vn
from oracle import zko

_DEFS = None


def defs():
    """(ctypes blob for the product, Defs view of the same data for the native restatement)"""
    global _DEFS
    if _DEFS is None:
        d = zkgl.opcode_defs_default()
        _DEFS = (d, vn.defs_from_ctypes(d, zkgl.VM_VARIANT, zkgl.VM_FLAG, zkgl.VM_CONDITION, zkgl.VM_PARAM))
    return _DEFS


_CS = {}


def vm_cs(limit, max_trace_len=1 << 22):
    key = limit
    if key not in _CS:
        cs = zkgl.ConstraintSystem(zkgl.CSGeometry(140, 0, 8, 8), max_trace_len, 1 << 28)  # reference_vm_geometry, cycle.rs:959-966
        cs.configure_main_vm(defs()[0])
        cs.main_vm_entry_point(limit)
        cs.pad_and_shrink()
        _CS[key] = cs
    return _CS[key]


class Asm:
    """tiny assembler over Defs.asm: registers are 1..15 (0 = none)"""

    def __init__(self, D):
        self.D = D
        self.ops = []

    def emit(self, *a, **k):
        self.ops.append(self.D.asm(*a, **k))
        return len(self.ops) - 1

    def li(self, reg, value):
        """load a small constant: add imm16, r0 -> reg"""
        assert 0 <= value < 65536
        return self.emit("ADD", src_mode="IMM16", imm0=value, src1=0, dst0=reg)

    def load_u256(self, reg, value, tmp=14, tmp2=13):
        """build an arbitrary 256-bit constant in `reg` from 16-bit pieces (shl by 16 + or), starting at the top non-zero piece"""
        pieces = [(value >> (16 * k)) & 0xFFFF for k in range(16)]
        top = max([k for k in range(16) if pieces[k]] or [0])
        self.li(reg, pieces[top])
        if top:
            self.li(tmp2, 16)
        for k in range(top - 1, -1, -1):
            self.emit("SHIFT", "SHIFT_SHL", src0=reg, src1=tmp2, dst0=reg)
            if pieces[k]:
                self.li(tmp, pieces[k])
                self.emit("BINOP", "BINOP_OR", src0=reg, src1=tmp, dst0=reg)


def program_arith(D, seed=0):
    a = Asm(D)
    a.li(1, (0x1234 + 977 * seed) & 0xFFFF)
    a.li(2, 77 + seed)
    a.emit("ADD", src0=1, src1=2, dst0=3, flags=("SET_FLAGS",))
    a.emit("SUB", src0=2, src1=1, dst0=4, flags=("SET_FLAGS",))                      # underflow: of = 1
    a.emit("SUB", src0=2, src1=1, dst0=5, flags=("SET_FLAGS", "SWAP_ARITH"))         # swapped: r1 - r2
    a.emit("MUL", src0=4, src1=4, dst0=6, dst1=7, flags=("SET_FLAGS",))              # (2^256 - x)^2: high part non-zero
    a.emit("DIV", src0=6, src1=1, dst0=8, dst1=9, flags=("SET_FLAGS",))
    a.emit("DIV", src0=6, src1=0, dst0=8, dst1=9, flags=("SET_FLAGS",))              # division by zero: of
    a.emit("BINOP", "BINOP_XOR", src0=4, src1=6, dst0=10, flags=("SET_FLAGS",))
    a.emit("BINOP", "BINOP_AND", src0=4, src1=6, dst0=11)
    a.emit("BINOP", "BINOP_OR", src0=4, src1=6, dst0=12)
    a.li(13, 200)
    a.emit("SHIFT", "SHIFT_SHL", src0=4, src1=13, dst0=3)
    a.emit("SHIFT", "SHIFT_SHR", src0=4, src1=13, dst0=3, flags=("SET_FLAGS",))
    a.emit("SHIFT", "SHIFT_ROL", src0=4, src1=13, dst0=5)
    a.emit("SHIFT", "SHIFT_ROR", src0=4, src1=13, dst0=5)
    a.emit("SHIFT", "SHIFT_ROR", src0=4, src1=0, dst0=5)                             # rotation by zero
    a.emit("SHIFT", "SHIFT_SHL", src0=13, src1=4, dst0=5, flags=("SWAP_ARITH",))     # swapped operands
    # stack traffic: push r4, push r6, pop into add, relative and absolute addressing
    a.emit("ADD", src0=4, src1=0, dst_mode="STACK_PUSH_POP", imm1=1)
    a.emit("ADD", src0=6, src1=0, dst_mode="STACK_PUSH_POP", imm1=1)
    a.emit("ADD", src_mode="STACK_OFFSET", imm0=1, src1=1, dst0=3)                   # sp - 1
    a.emit("ADD", src_mode="STACK_PUSH_POP", imm0=1, src1=0, dst0=5)                 # pop
    a.emit("MUL", src_mode="ABSOLUTE_STACK", imm0=0, src1=2, dst_mode="STACK_OFFSET", imm1=1, dst1=7)
    a.emit("SUB", src_mode="CODE_PAGE", imm0=1, src1=1, dst_mode="ABSOLUTE_STACK", imm1=5)   # constant from the code page
    a.emit("NOP", src_mode="STACK_PUSH_POP", imm0=1, dst_mode="STACK_PUSH_POP", imm1=3)      # nop only moves sp
    # conditions: eq is clear after the shr above? set flags explicitly, then conditional ops
    a.emit("SUB", src0=1, src1=1, dst0=3, flags=("SET_FLAGS",))                      # eq
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="NE")                                 # masked into NOP
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="EQ")
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="GT")                                 # NOP
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="LE")
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="LT")                                 # NOP
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="GE")
    a.emit("ADD", src0=1, src1=2, dst0=3, cond="GT_OR_LT")                           # NOP
    j = len(a.ops)
    a.emit("JUMP", src_mode="IMM16", imm0=j + 3)                                     # skip two opcodes
    a.emit("ADD", src0=1, src1=1, dst0=1)
    a.emit("ADD", src0=1, src1=1, dst0=1)
    a.emit("CONTEXT", "CTX_THIS", dst0=3)
    a.emit("CONTEXT", "CTX_CALLER", dst0=3)
    a.emit("CONTEXT", "CTX_CODE_ADDRESS", dst0=3)
    a.emit("CONTEXT", "CTX_META", dst0=3)
    a.emit("CONTEXT", "CTX_ERGS_LEFT", dst0=3)
    a.emit("CONTEXT", "CTX_SP", dst0=3)
    a.emit("CONTEXT", "CTX_SET_CONTEXT_U128", src0=6)
    a.emit("CONTEXT", "CTX_GET_CONTEXT_U128", dst0=3)
    a.emit("CONTEXT", "CTX_SET_ERGS_PER_PUBDATA", src0=2)
    a.emit("CONTEXT", "CTX_INC_TX_NUMBER")
    a.emit("RET", "RET_OK", src0=0)   # the bootloader frame must exit with pc == 0 (mod.rs:115-122)
    return a.ops


def program_memory_and_logs(D, seed=0):
    a = Asm(D)
    a.load_u256(1, 0x0102030405060708090A0B0C0D0E0F101112131415161718191A1B1C1D1E1F20 ^ (seed * 0x0101010101010101))
    a.li(2, 64)
    a.emit("UMA", "UMA_HEAP_WRITE", src0=2, src1=1, dst0=3, flags=("UMA_INCREMENT",))     # aligned write at 64, r3 = 96
    a.li(2, 65 + (12 + 5 * seed) % 31)
    a.emit("UMA", "UMA_HEAP_WRITE", src0=2, src1=1)                                        # unaligned write (offset 77 for seed 0): two cells
    a.emit("UMA", "UMA_HEAP_READ", src0=2, dst0=4, dst1=5, flags=("UMA_INCREMENT",))      # unaligned read, r5 = 109
    a.emit("UMA", "UMA_HEAP_READ", src_mode="IMM16", imm0=64, dst0=4)                      # aligned read, immediate offset
    a.emit("UMA", "UMA_AUX_HEAP_WRITE", src0=5, src1=4)
    a.emit("UMA", "UMA_AUX_HEAP_READ", src0=5, dst0=6)
    # a panicking access (offset with high limbs set -> pending exception -> PANIC next cycle) inside a near-call frame, so that
    # the panic unwinds to the root frame's handler instead of ending the bootloader
    a.li(7, 10000)                                                                         # ergs passed: the panics burn them all
    nc = a.emit("NEAR_CALL", src0=7, imm0=0, imm1=0)
    nc2 = a.emit("NEAR_CALL", src0=7, imm0=0, imm1=0)
    a.emit("RET", "RET_OK", src0=0)
    body = len(a.ops)
    a.emit("UMA", "UMA_HEAP_READ", src0=1, dst0=6)
    a.emit("NOP")                                                                          # replaced by the pending PANIC
    body2 = len(a.ops)
    a.emit("PTR", "PTR_ADD", src0=2, src1=2, dst0=6)                                       # ptr.add on a non-pointer: panic as well
    a.emit("NOP")
    a.ops[nc] = (a.ops[nc] & 0xFFFFFFFF) | (body << 32) | ((nc + 1) << 48)
    a.ops[nc2] = (a.ops[nc2] & 0xFFFFFFFF) | (body2 << 32) | ((nc2 + 1) << 48)
    return a.ops


def program_logs(D, seed=0):
    a = Asm(D)
    a.li(1, 5 + seed)        # key
    a.li(2, 0x3333 + seed)   # value
    a.li(9, 3)
    a.emit("CONTEXT", "CTX_SET_ERGS_PER_PUBDATA", src0=9)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=2)
    a.emit("LOG", "LOG_STORAGE_READ", src0=1, dst0=3)
    a.emit("LOG", "LOG_EVENT", src0=1, src1=2, flags=("FIRST_MESSAGE",))
    a.emit("LOG", "LOG_EVENT", src0=2, src1=1)
    a.emit("LOG", "LOG_TO_L1", src0=1, src1=2)
    a.li(4, 1000)
    a.emit("LOG", "LOG_PRECOMPILE_CALL", src0=1, src1=4, dst0=5)
    a.emit("LOG", "LOG_STORAGE_READ", src0=2, dst0=3)   # unknown key: 0
    a.emit("RET", "RET_OK", src0=0)
    return a.ops


FAR_ABI_FORWARD_FAT_POINTER = 1 << (8 * 28)
FAR_ABI_AUX_HEAP = 2 << (8 * 28)


def program_calls(D, seed=0, callee_a=0x10001, callee_b=0x10002, callee_c=0x10004, missing=0x10003):
    """near calls (ok / revert with writes inside, nested), far calls (ok with returndata, static violation -> panic, revert with
    writes, missing code -> exception), fat pointers in both directions.  A far return clears r2..r15, so the ABI registers are
    rebuilt before every far call."""
    a = Asm(D)
    a.li(1, 5 + seed)
    a.li(2, 0x4444 + 3 * seed)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=2)                 # a write in the root frame before any call
    a.li(3, 0)                                                         # near call abi: pass all ergs
    nc1 = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)                  # patched below
    nc2 = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)
    a.li(6, 64)
    a.load_u256(7, 0xA1A2A3A4A5A6A7A8A9AAABACADAEAFB0B1B2B3B4B5B6B7B8B9BABBBCBDBEBFC0 ^ seed)
    a.emit("UMA", "UMA_HEAP_WRITE", src0=6, src1=7)                    # calldata at heap [64, 96)
    abi = (64 << 64) | (64 << 96) | (100000 << 192)                    # start 64, length 64, ergs passed 100000, forwarding = heap

    def far_call(variant, target, eh, flags=()):
        a.load_u256(4, abi)
        a.load_u256(5, target)
        return a.emit("FAR_CALL", variant, src0=4, src1=5, imm0=eh, flags=flags)

    fc1 = far_call("FAR_NORMAL", callee_a, 0)
    a.li(3, 0)
    a.emit("PTR", "PTR_ADD", src0=1, src1=3, dst0=8)                   # r1 = returndata fat pointer (+0)
    a.emit("UMA", "UMA_FAT_PTR_READ", src0=1, dst0=9)                  # read the returndata through the pointer
    fc2 = far_call("FAR_NORMAL", callee_b, 0, flags=("FAR_CALL_STATIC",))   # callee writes in a static context: panic
    fc4 = far_call("FAR_NORMAL", callee_c, 0)                          # callee writes, then reverts
    fc3 = far_call("FAR_DELEGATE", missing, 0)                         # no code: exception -> panic in the new frame -> eh
    a.emit("NOP")
    a.li(10, 1)
    a.emit("RET", "RET_OK", src0=0)                                    # the root frame returns: execution ends
    a.emit("NOP")
    # ---- near call bodies
    body1 = len(a.ops)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=1)
    a.emit("LOG", "LOG_EVENT", src0=2, src1=1)
    a.emit("RET", "RET_OK", src0=0)
    body2 = len(a.ops)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=2, src1=2)
    inner = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)                # nested ok frame inside the reverting one
    a.emit("LOG", "LOG_TO_L1", src0=1, src1=2)
    a.emit("RET", "RET_REVERT", src0=0, flags=("RET_TO_LABEL",), imm0=nc2 + 1)
    body3 = len(a.ops)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=2)
    a.emit("RET", "RET_OK", src0=0)

    def patch(i, imm0, imm1):
        a.ops[i] = (a.ops[i] & 0xFFFFFFFF) | (imm0 << 32) | (imm1 << 48)

    patch(nc1, body1, 0xFFFF)
    patch(nc2, body2, nc2 + 1)
    patch(inner, body3, 0xFFFF)
    for fc in (fc1, fc2, fc3, fc4):
        patch(fc, fc + 1, 0)                                           # exception handler = the next opcode
    # ---- callee A: reads calldata through the fat pointer, writes storage, returns 32 bytes of its heap
    ca = Asm(D)
    ca.emit("UMA", "UMA_FAT_PTR_READ", src0=1, dst0=2, dst1=3, flags=("UMA_INCREMENT",))
    ca.emit("UMA", "UMA_FAT_PTR_READ", src0=3, dst0=4)                  # second word
    ca.emit("PTR", "PTR_SHRINK", src0=1, src1=0, dst0=5)
    ca.li(6, 9)
    ca.emit("LOG", "LOG_STORAGE_WRITE", src0=6, src1=2)
    ca.li(7, 0)
    ca.emit("UMA", "UMA_HEAP_WRITE", src0=7, src1=2)
    ca.load_u256(8, (0 << 64) | (32 << 96))                             # returndata: heap [0, 32)
    ca.emit("RET", "RET_OK", src0=8)
    # ---- callee B (called statically): a read is fine, the write is masked into PANIC
    cb = Asm(D)
    cb.li(6, 9)
    cb.emit("LOG", "LOG_STORAGE_READ", src0=6, dst0=2)
    cb.emit("LOG", "LOG_STORAGE_WRITE", src0=6, src1=6)
    cb.emit("RET", "RET_OK", src0=0)
    # ---- callee C: writes, then reverts with returndata in the aux heap
    cc = Asm(D)
    cc.li(6, 11)
    cc.emit("LOG", "LOG_STORAGE_WRITE", src0=6, src1=6)
    cc.li(7, 12)
    cc.emit("LOG", "LOG_STORAGE_WRITE", src0=7, src1=6)
    cc.load_u256(8, (0 << 64) | (40 << 96) | FAR_ABI_AUX_HEAP)
    cc.emit("RET", "RET_REVERT", src0=8)
    return a.ops, {callee_a: ca.ops, callee_b: cb.ops, callee_c: cc.ops}


def program_bench_loop(D, seed=0, callee_a=0x10001, callee_c=0x10004, realistic=False):
    """endless mixed workload for the throughput runs (bench.py config C2): every family, calls nested two deep, the root frame
    never returns.  One trip around the loop is ~150 cycles.
    realistic=True: the same trip behind a counted inner loop of register / stack / heap arithmetic (40 x 21 instructions), so that
    the mix looks like compiled contract code — ~1 % storage / event logs, ~0.3 % near calls, ~0.2 % far calls, ~10 % heap accesses,
    the rest arithmetic, jumps and stack traffic — instead of every family every 150 cycles."""
    a = Asm(D)
    a.li(9, 3)
    a.emit("CONTEXT", "CTX_SET_ERGS_PER_PUBDATA", src0=9)
    top = len(a.ops)
    if realistic:
        a.li(15, 40)
        a.li(14, 1)
        a.li(1, (0x1234 + 977 * seed) & 0xFFFF)
        a.li(2, 77 + seed)
        a.li(13, 7)
        a.li(12, 96)
        inner_top = len(a.ops)
        a.emit("ADD", src0=1, src1=2, dst0=3)
        a.emit("SUB", src0=3, src1=14, dst0=4, flags=("SET_FLAGS",))
        a.emit("BINOP", "BINOP_AND", src0=3, src1=4, dst0=5)
        a.emit("BINOP", "BINOP_XOR", src0=5, src1=1, dst0=6)
        a.emit("SHIFT", "SHIFT_SHL", src0=6, src1=13, dst0=7)
        a.emit("ADD", src0=7, src1=0, dst_mode="STACK_PUSH_POP", imm1=1)                    # push
        a.emit("ADD", src0=4, src1=0, dst_mode="STACK_PUSH_POP", imm1=1)                    # push
        a.emit("MUL", src0=3, src1=5, dst0=8, dst1=9)
        a.emit("ADD", src_mode="STACK_PUSH_POP", imm0=1, src1=8, dst0=10)                   # pop
        a.emit("ADD", src_mode="STACK_PUSH_POP", imm0=1, src1=10, dst0=11)                  # pop
        a.emit("UMA", "UMA_HEAP_WRITE", src0=12, src1=11)
        a.emit("BINOP", "BINOP_OR", src0=11, src1=2, dst0=1)
        a.emit("UMA", "UMA_HEAP_READ", src0=12, dst0=2)
        a.emit("ADD", src0=2, src1=14, dst0=2)
        a.emit("SUB", src0=1, src1=2, dst0=3, flags=("SET_FLAGS",))
        a.emit("ADD", src0=3, src1=1, dst0=1, cond="GT")
        a.emit("SHIFT", "SHIFT_SHR", src0=1, src1=13, dst0=4)
        a.emit("ADD", src_mode="CODE_PAGE", imm0=1, src1=4, dst0=5)
        a.emit("DIV", src0=8, src1=13, dst0=6, dst1=7)
        a.emit("SUB", src0=15, src1=14, dst0=15, flags=("SET_FLAGS",))
        a.emit("JUMP", src_mode="IMM16", imm0=inner_top, cond="NE")
    a.li(1, (0x1234 + 977 * seed) & 0xFFFF)
    a.li(2, 77 + seed)
    a.emit("SUB", src0=2, src1=1, dst0=4, flags=("SET_FLAGS",))
    a.emit("MUL", src0=4, src1=4, dst0=6, dst1=7, flags=("SET_FLAGS",))
    a.emit("DIV", src0=6, src1=1, dst0=8, dst1=9)
    a.emit("BINOP", "BINOP_XOR", src0=4, src1=6, dst0=10, flags=("SET_FLAGS",))
    a.emit("BINOP", "BINOP_AND", src0=4, src1=8, dst0=11)
    a.li(13, 100 + seed)
    a.emit("SHIFT", "SHIFT_ROL", src0=4, src1=13, dst0=3)
    a.emit("SHIFT", "SHIFT_SHR", src0=6, src1=13, dst0=5, flags=("SET_FLAGS",))
    a.emit("ADD", src0=4, src1=0, dst_mode="STACK_PUSH_POP", imm1=1)
    a.emit("ADD", src_mode="STACK_PUSH_POP", imm0=1, src1=3, dst0=5, cond="NE")
    a.emit("ADD", src_mode="STACK_PUSH_POP", imm0=1, src1=3, dst0=5, cond="EQ")
    a.li(12, 64)
    a.emit("UMA", "UMA_HEAP_WRITE", src0=12, src1=6, dst0=12, flags=("UMA_INCREMENT",))
    a.li(12, 77)
    a.emit("UMA", "UMA_HEAP_WRITE", src0=12, src1=10)
    a.emit("UMA", "UMA_HEAP_READ", src0=12, dst0=8, dst1=9, flags=("UMA_INCREMENT",))
    a.emit("UMA", "UMA_AUX_HEAP_WRITE", src0=9, src1=8)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=2)
    a.emit("LOG", "LOG_STORAGE_READ", src0=1, dst0=3)
    a.emit("LOG", "LOG_EVENT", src0=1, src1=2, flags=("FIRST_MESSAGE",))
    a.emit("LOG", "LOG_TO_L1", src0=1, src1=2)
    a.li(3, 1000)
    a.emit("LOG", "LOG_PRECOMPILE_CALL", src0=1, src1=3, dst0=5)
    a.emit("CONTEXT", "CTX_META", dst0=3)
    a.emit("CONTEXT", "CTX_INC_TX_NUMBER")
    a.li(3, 0)
    nc1 = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)
    nc2 = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)
    abi = (64 << 64) | (64 << 96) | (100000 << 192)

    def far_call(variant, target, flags=()):
        a.load_u256(4, abi)
        a.load_u256(5, target)
        i = a.emit("FAR_CALL", variant, src0=4, src1=5, imm0=0, flags=flags)
        a.ops[i] |= (i + 1) << 32
        return i

    far_call("FAR_NORMAL", callee_a)
    a.li(3, 0)
    a.emit("PTR", "PTR_ADD", src0=1, src1=3, dst0=8)
    a.emit("UMA", "UMA_FAT_PTR_READ", src0=1, dst0=9)
    far_call("FAR_NORMAL", callee_c)
    a.emit("JUMP", src_mode="IMM16", imm0=top)
    body1 = len(a.ops)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=1, src1=1)
    inner = a.emit("NEAR_CALL", src0=3, imm0=0, imm1=0)
    a.emit("RET", "RET_OK", src0=0)
    body2 = len(a.ops)
    a.emit("LOG", "LOG_STORAGE_WRITE", src0=2, src1=2)
    a.emit("LOG", "LOG_EVENT", src0=2, src1=1)
    a.emit("RET", "RET_REVERT", src0=0, flags=("RET_TO_LABEL",), imm0=nc2 + 1)
    body3 = len(a.ops)
    a.emit("LOG", "LOG_EVENT", src0=1, src1=2)
    a.emit("RET", "RET_OK", src0=0)
    for i, (dst, eh) in ((nc1, (body1, 0xFFFF)), (nc2, (body2, nc2 + 1)), (inner, (body3, 0xFFFF))):
        a.ops[i] = (a.ops[i] & 0xFFFFFFFF) | (dst << 32) | (eh << 48)
    ca = Asm(D)
    ca.emit("UMA", "UMA_FAT_PTR_READ", src0=1, dst0=2, dst1=3, flags=("UMA_INCREMENT",))
    ca.emit("UMA", "UMA_FAT_PTR_READ", src0=3, dst0=4)
    ca.li(6, 9)
    ca.emit("LOG", "LOG_STORAGE_WRITE", src0=6, src1=2)
    ca.li(7, 0)
    ca.emit("UMA", "UMA_HEAP_WRITE", src0=7, src1=2)
    ca.load_u256(8, (0 << 64) | (32 << 96))
    ca.emit("RET", "RET_OK", src0=8)
    cc = Asm(D)
    cc.li(6, 11)
    cc.emit("LOG", "LOG_STORAGE_WRITE", src0=6, src1=6)
    cc.load_u256(8, (0 << 64) | (40 << 96) | FAR_ABI_AUX_HEAP)
    cc.emit("RET", "RET_REVERT", src0=8)
    return a.ops, {callee_a: ca.ops, callee_c: cc.ops}


def make_world_factory(D, boot_ops, contracts=None):
    def make():
        w = vn.World()
        w.load_code(D.p("BOOTLOADER_CODE_PAGE"), boot_ops)
        for addr, ops in (contracts or {}).items():
            w.deploy(D, addr, ops)
        return w
    return make


# ------------------------------------------------------------------------------------------------ streams
def pack_instance_streams(cs, D, run: "vn.VmRun", limit, n_instances, first_cycle=0):
    """-> (outer [words, B], loop [words, B * limit]) for consecutive chunks of one execution (instance i continues instance i-1)"""
    lay = cs.main_vm_layout()
    ow, lw = cs.input_words()
    outer = np.zeros((ow, n_instances), dtype=np.uint64)
    loop = np.zeros((lw, n_instances * limit), dtype=np.uint64)

    def put(arr, scope, name, col, words):
        first, n = lay[scope][name]
        assert len(words) == n, (name, len(words), n)
        arr[first:first + n, col] = np.array([int(x) for x in words], dtype=np.uint64)

    zporter, default_aa = run.gctx
    for i in range(n_instances):
        c0 = first_cycle + i * limit
        put(outer, "outer", "start_flag", i, [1 if c0 == 0 else 0])
        put(outer, "outer", "rollback_queue_tail_for_block", i, run.rollback_tail_for_block)
        put(outer, "outer", "memory_queue_initial_tail", i, [0] * 12)
        put(outer, "outer", "memory_queue_initial_length", i, [0])
        put(outer, "outer", "decommitment_queue_initial_tail", i, [0] * 12)
        put(outer, "outer", "decommitment_queue_initial_length", i, [0])
        put(outer, "outer", "zkporter_is_available", i, [zporter])
        put(outer, "outer", "default_aa_code_hash", i, vn.limbs(default_aa))
        put(outer, "outer", "hidden_fsm_input", i, [0] * 243 if c0 == 0 else run.states[c0].flatten())
        for k in range(limit):
            st, W = run.rows[c0 + k]
            col = i * limit + k
            put(loop, "loop", "state", col, st.flatten())
            for name, words in W.items():
                if not name.startswith("_"):
                    put(loop, "loop", name, col, words)
    return outer, loop


def expected_commitment(D, run, limit, instance, first_cycle=0):
    """input_commitment of main_vm_entry_point for one chunk (mod.rs:112-231, fsm_input_output/mod.rs:178-253)"""
    c0 = first_cycle + instance * limit
    start = 1 if c0 == 0 else 0
    final = run.states[c0 + limit]
    done = int(final.depth == 0)
    zporter, default_aa = run.gctx
    obs_in = list(run.rollback_tail_for_block) + [0] * 13 + [0] * 13 + [zporter] + vn.limbs(default_aa)
    z = lambda n: [0] * n
    obs_out = (z(4) + (list(final.fwd_tail) if done else z(4)) + [final.fwd_len if done else 0] +
               z(12) + (list(final.mem_tail) if done else z(12)) + [final.mem_len if done else 0] +
               z(12) + (list(final.dec_tail) if done else z(12)) + [final.dec_len if done else 0])
    fsm_in = [0] * 243 if start else run.states[c0].flatten()
    c_obs_in, c_obs_out = zko.commit_encoding(obs_in), zko.commit_encoding(obs_out)
    c_fsm_in, c_fsm_out = zko.commit_encoding(fsm_in), zko.commit_encoding(final.flatten())
    compact = [start, done] + c_obs_in + (c_obs_out if done else z(4)) + (z(4) if start else c_fsm_in) + (z(4) if done else c_fsm_out)
    return zko.commit_encoding(compact)


def mixed_batch(cs, D, limit, min_instances, seeds=None):
    """instances drawn from the four programs with different seeds, each execution cut into consecutive `limit`-cycle chunks up to
    (and including) the chunk in which the bootloader frame returns.  -> (outer, loop, [expected commitment per instance], info)"""
    outers, loops, commits, info = [], [], [], []
    seed = 0
    while sum(o.shape[1] for o in outers) < min_instances:
        for name in ("arith", "memory", "logs", "calls"):
            contracts = None
            if name == "calls":
                ops, contracts = program_calls(D, seed)
            else:
                ops = dict(arith=program_arith, memory=program_memory_and_logs, logs=program_logs)[name](D, seed)
            probe = vn.VmRun(D, make_world_factory(D, ops, contracts), 4 * len(ops) + 64)
            done_at = next(i for i, s in enumerate(probe.states) if s.depth == 0)
            n_inst = (done_at + 1 + limit - 1) // limit
            vrun = vn.VmRun(D, make_world_factory(D, ops, contracts), n_inst * limit)
            # the streams the circuit is fed come from the product's packer (zk_pack_main_vm_witness over the oracle FIFOs); the native
            # model's own placement of the answers only checks them, and supplies the expected per-cycle VmLocalState
            o, l = pack_through_the_c_abi(cs, vrun, limit, n_inst)[:2]
            want_o, want_l = pack_instance_streams(cs, D, vrun, limit, n_inst)
            assert np.array_equal(o, want_o) and np.array_equal(l[243:], want_l[243:]) and not l[:243].any(), (name, seed)
            l[:243] = want_l[:243]   # expected state, blanked again by the tests before it reaches the device
            outers.append(o); loops.append(l)
            commits += [expected_commitment(D, vrun, limit, i) for i in range(n_inst)]
            info += [(name, seed, i) for i in range(n_inst)]
        seed += 1
    return np.concatenate(outers, axis=1), np.concatenate(loops, axis=1), commits, info


# ------------------------------------------------------------------------------------------------ the product-side input path
def oracle_queues(run: "vn.VmRun", first_cycle=0, n_cycles=None):
    """the reference WitnessOracle's per-getter FIFOs (witness_oracle.rs:45-91) of an execution, from the calls the native model
    answered under execute == true — what a host holding a VmCircuitWitness has (NOT placed at cycles)"""
    q = zkgl.VmOracleQueues()
    rows = run.rows[first_cycle:] if n_cycles is None else run.rows[first_cycle:first_cycle + n_cycles]
    for _, W in rows:
        for call in W["_oracle_calls"]:
            kind = call[0]
            if kind == "memory_read":
                q.memory_reads.append((call[1], call[2]))
            elif kind == "storage_read":
                q.storage_reads.append(call[1])
            elif kind == "refund":
                q.refunds.append(call[1])
            elif kind == "rollback_queue_witness":
                q.rollback_queue_witness.append(call[1])
            elif kind == "rollback_tail_for_call":
                q.rollback_tails_for_call.append(call[1])
            elif kind == "callstack":
                q.callstack.append((call[1], call[2]))
            elif kind == "decommit_page":
                q.decommit_pages.append(call[1])
            else:
                raise KeyError(kind)
    return q.freeze()


def closed_form_input(run: "vn.VmRun", c0):
    """VmCircuitInputOutputWitness of the chunk that starts at cycle c0 (circuit_inputs/main_vm.rs:7-62)"""
    cf = zkgl.VmClosedFormInput()
    cf.start_flag = 1 if c0 == 0 else 0
    cf.rollback_queue_tail_for_block[:] = [int(x) for x in run.rollback_tail_for_block]
    zporter, default_aa = run.gctx
    cf.zkporter_is_available = int(zporter)
    cf.default_aa_code_hash[:] = vn.limbs(default_aa)
    if c0:
        cf.hidden_fsm_input[:] = [int(x) for x in run.states[c0].flatten()]
    return cf


def pack_through_the_c_abi(cs, run: "vn.VmRun", limit, n_instances, first_cycle=0, fill_state=False):
    """(outer, loop, reports): consecutive chunks of one execution through zk_pack_main_vm_witness, the FIFOs continued chunk to chunk"""
    ow, lw = cs.input_words()
    outer = np.zeros((ow, n_instances), dtype=np.uint64)
    loop = np.zeros((lw, n_instances * limit), dtype=np.uint64)
    q = oracle_queues(run, first_cycle, n_instances * limit)
    used = [0] * 7
    reports = []
    for i in range(n_instances):
        rep = cs.pack_main_vm_witness(closed_form_input(run, first_cycle + i * limit), q.view(used), i, n_instances, outer, loop,
                                      zkgl.VM_PACK_FILL_STATE if fill_state else 0)
        got = [rep.used_memory_reads, rep.used_storage_reads, rep.used_refunds, rep.used_rollback_queue_witness, rep.used_rollback_tails_for_call,
               rep.used_callstack, rep.used_decommit_pages]
        used = [a + b for a, b in zip(used, got)]
        reports.append(rep)
    return outer, loop, reports
