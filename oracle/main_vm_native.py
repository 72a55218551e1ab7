"""oracle/main_vm_native.py — CPU ORACLE (test infrastructure): native, value-level restatement of the main_vm circuit
straight from the Rust (/root/reference/src/main_vm/): one `vm_cycle` (cycle.rs:28-795) = create_prestate (pre_state.rs:71-519),
perform_initial_decoding (decoded_opcode.rs:42-220), the eleven opcode families (opcodes/*.rs, opcodes/call_ret_impl/*.rs) and
the state-diff application, over Python integers.  It plays three roles for the tests:

  * the expected VmLocalState after every cycle (compared word for word with what the recorded circuit derives),
  * the WitnessOracle (witness_oracle.rs:45-91) of a small synthetic world (memory, storage, decommitter, callstack), i.e. the
    raw words of the per-cycle input stream,
  * the rollback-queue non-determinism: `get_rollback_queue_witness` / `get_rollback_queue_tail_witness_for_call` need knowledge
    of the future (saved_context.rs:16-35), so a run is two passes — pass 1 records the log / call / ret events, `plan_rollbacks`
    walks them backwards from every frame's end, pass 2 replays with the planned values.

Everything zkevm_opcode_defs supplies comes from the same blob the product takes (include/zkgl_vm.h), handed over as a dict.
"""
from __future__ import annotations

import copy

import numpy as np

from . import zko
from .storage_native import encode as log_encode

P = zko.P
NREG = 15
M32, M16, M256 = 0xFFFFFFFF, 0xFFFF, (1 << 256) - 1

FAM = dict(INVALID=0, NOP=1, ADD=2, SUB=3, MUL=4, DIV=5, JUMP=6, CONTEXT=7, SHIFT=8, BINOP=9, PTR=10, NEAR_CALL=11, LOG=12, FAR_CALL=13, RET=14, UMA=15)
MODE = dict(REG_ONLY=0, STACK_PUSH_POP=1, STACK_OFFSET=2, ABSOLUTE_STACK=3, IMM16=4, CODE_PAGE=5)


def limbs(x, n=8):
    return [(x >> (32 * i)) & M32 for i in range(n)]


def from_limbs(l):
    return sum(int(v) << (32 * i) for i, v in enumerate(l))


class Defs:
    """view of the zk_opcode_defs blob (dict with the struct's field names; arrays as lists, enums by name)"""

    def __init__(self, d: dict):
        self.__dict__.update(d)
        self.variant_bit0 = self.type_bits
        self.flag_bit0 = self.variant_bit0 + self.variant_bits
        self.src_bit0 = self.flag_bit0 + self.flag_bits
        self.dst_bit0 = self.src_bit0 + self.src_mode_bits
        self.aux_bit0 = self.description_bits_flattened
        self.props_mask = (1 << self.description_bits_flattened) - 1

    def p(self, name):
        return self.params[name]

    def find(self, family, variant=0, src_mode=0, dst_mode=0, flags=0):
        want = (1 << family) | (1 << (self.variant_bit0 + variant)) | (flags << self.flag_bit0) | (1 << (self.src_bit0 + src_mode)) | (1 << (self.dst_bit0 + dst_mode))
        for i in range(self.n_valid):
            if self.props[i] & self.props_mask == want:
                return i
        raise KeyError((family, variant, src_mode, dst_mode, flags))

    def asm(self, family, variant=None, src_mode="REG_ONLY", dst_mode="REG_ONLY", flags=(), cond="ALWAYS", src0=0, src1=0, dst0=0, dst1=0, imm0=0, imm1=0):
        """64-bit opcode word (layout: decoded_opcode.rs:408-514); registers are 0 (none) or 1..15"""
        fam = FAM[family]
        var = self.variant_idx[variant] if variant else 0
        fl = 0
        for f in flags:
            fl |= 1 << self.flag_idx[f]
        idx = self.find(fam, var, MODE[src_mode], MODE[dst_mode], fl)
        return idx | (self.condition_idx[cond] << 13) | (src0 << 16) | (src1 << 20) | (dst0 << 24) | (dst1 << 28) | (imm0 << 32) | (imm1 << 48)


def defs_from_ctypes(d, variant_names, flag_names, cond_names, param_names):
    """zkgl.OpcodeDefs -> Defs (the test passes zkgl's name tables; the oracle never imports the product)"""
    return Defs(dict(
        version=d.version, n_valid=d.n_valid, props=[int(x) for x in d.props], prices=[int(x) for x in d.prices],
        type_bits=d.type_bits, variant_bits=d.variant_bits, flag_bits=d.flag_bits, src_mode_bits=d.src_mode_bits, dst_mode_bits=d.dst_mode_bits,
        description_bits_flattened=d.description_bits_flattened, aux_bits=d.aux_bits, aux_kernel_mode=d.aux_kernel_mode,
        aux_static_ok=d.aux_static_ok, aux_explicit_panic=d.aux_explicit_panic,
        variant_idx={n: int(d.variant_idx[i]) for n, i in variant_names.items()}, flag_idx={n: int(d.flag_idx[i]) for n, i in flag_names.items()},
        condition_idx={n: int(d.condition_idx[i]) for n, i in cond_names.items()},
        can_write_dst0_into_memory=[int(x) for x in d.can_write_dst0_into_memory], nop_encoding=int(d.nop_encoding),
        panic_encoding=int(d.panic_encoding), nop_bitspread=int(d.nop_bitspread), panic_bitspread=int(d.panic_bitspread),
        params={n: int(d.params[i]) for n, i in param_names.items()}))


# ------------------------------------------------------------------------------------------------ state
class Ctx:
    """ExecutionContextRecord (saved_context.rs:37-68); addresses as integers < 2^160"""
    FIELDS = ("this", "caller", "code_address", "code_page", "base_page", "heap_bound", "aux_heap_bound", "rq_head", "rq_tail", "rq_len", "pc", "sp",
              "eh", "ergs", "is_static", "is_kernel", "this_shard", "caller_shard", "code_shard", "ctx_u128", "is_local")

    def __init__(self):
        self.this = self.caller = self.code_address = 0
        self.code_page = self.base_page = self.heap_bound = self.aux_heap_bound = 0
        self.rq_head, self.rq_tail, self.rq_len = [0] * 4, [0] * 4, 0
        self.pc = self.sp = self.eh = self.ergs = 0
        self.is_static = self.is_kernel = 0
        self.this_shard = self.caller_shard = self.code_shard = 0
        self.ctx_u128 = 0
        self.is_local = 0

    def flatten(self):  # flatten_as_variables, saved_context.rs:279-323
        return (limbs(self.this, 5) + limbs(self.caller, 5) + limbs(self.code_address, 5) + [self.code_page, self.base_page, self.heap_bound, self.aux_heap_bound] +
                list(self.rq_head) + list(self.rq_tail) + [self.rq_len, self.pc, self.sp, self.eh, self.ergs, self.is_static, self.is_kernel, self.this_shard,
                                                           self.caller_shard, self.code_shard] + limbs(self.ctx_u128, 4) + [self.is_local])

    @staticmethod
    def unflatten(f):
        c = Ctx()
        c.this, c.caller, c.code_address = from_limbs(f[0:5]), from_limbs(f[5:10]), from_limbs(f[10:15])
        c.code_page, c.base_page, c.heap_bound, c.aux_heap_bound = f[15:19]
        c.rq_head, c.rq_tail, c.rq_len = list(f[19:23]), list(f[23:27]), f[27]
        c.pc, c.sp, c.eh, c.ergs, c.is_static, c.is_kernel, c.this_shard, c.caller_shard, c.code_shard = f[28:37]
        c.ctx_u128 = from_limbs(f[37:41])
        c.is_local = f[41]
        return c

    def encode(self):  # ExecutionContextRecord::encode, saved_context.rs:111-266
        d = [(self.rq_len >> (8 * k)) & 0xFF for k in range(4)]
        v = list(self.rq_head) + list(self.rq_tail) + limbs(self.code_address, 5) + limbs(self.this, 5) + limbs(self.caller, 5) + limbs(self.ctx_u128, 4)
        v.append(self.code_page + (self.pc << 32) + (self.this_shard << 48) + (self.is_static << 56))
        v.append(self.base_page + (self.sp << 32) + (self.caller_shard << 48) + (self.is_kernel << 56))
        v.append(self.ergs + (self.eh << 32) + (self.code_shard << 48) + (self.is_local << 56))
        v.append(self.heap_bound + (d[0] << 32) + (d[1] << 40))
        v.append(self.aux_heap_bound + (d[2] << 32) + (d[3] << 40))
        return v


class VmState:
    """VmLocalState (src/base_structures/vm_state/mod.rs:92-109)"""

    def __init__(self):
        self.prev_code_word = 0
        self.regs = [(0, 0) for _ in range(NREG)]  # (is_pointer, value)
        self.flags = (0, 0, 0)                     # of, eq, gt
        self.timestamp = self.page_counter = self.tx_number = self.prev_code_page = self.prev_super_pc = 0
        self.pending_exception = 0
        self.ergs_per_pubdata = 0
        self.ctx = Ctx()
        self.fwd_tail, self.fwd_len = [0] * 4, 0
        self.depth = 0
        self.sponge = [0] * 12
        self.mem_tail, self.mem_len = [0] * 12, 0
        self.dec_tail, self.dec_len = [0] * 12, 0
        self.ctx_u128 = 0

    def flatten(self):
        o = limbs(self.prev_code_word)
        for ptr, v in self.regs:
            o += [ptr] + limbs(v)
        o += list(self.flags) + [self.timestamp, self.page_counter, self.tx_number, self.prev_code_page, self.prev_super_pc, self.pending_exception,
                                 self.ergs_per_pubdata]
        o += self.ctx.flatten() + list(self.fwd_tail) + [self.fwd_len, self.depth] + list(self.sponge)
        o += list(self.mem_tail) + [self.mem_len] + list(self.dec_tail) + [self.dec_len] + limbs(self.ctx_u128, 4)
        assert len(o) == 243
        return o


def initial_bootloader_state(D: Defs, mem_len, mem_tail, dec_len, dec_tail, rollback_tail):
    """loading.rs:11-226"""
    s = VmState()
    c = s.ctx
    c.base_page, c.code_page = D.p("BOOTLOADER_BASE_PAGE"), D.p("BOOTLOADER_CODE_PAGE")
    c.eh, c.ergs = D.p("INITIAL_FRAME_FORMAL_EH_LOCATION"), D.p("VM_INITIAL_FRAME_ERGS")
    c.code_address = c.this = D.p("BOOTLOADER_FORMAL_ADDRESS_LOW")
    c.rq_tail, c.rq_head = list(rollback_tail), list(rollback_tail)
    c.is_kernel = 1
    c.heap_bound = c.aux_heap_bound = D.p("BOOTLOADER_MAX_MEMORY")
    empty = Ctx()
    empty.rq_tail, empty.rq_head, empty.is_kernel = list(rollback_tail), list(rollback_tail), 1
    sponge = [0] * 12
    enc = empty.encode()
    for r in range(4):
        sponge = zko.poseidon2_permute(enc[8 * r:8 * r + 8] + sponge[8:12])
    s.sponge, s.depth = sponge, 1
    s.mem_len, s.mem_tail, s.dec_len, s.dec_tail = mem_len, list(mem_tail), dec_len, list(dec_tail)
    s.timestamp, s.page_counter = D.p("STARTING_TIMESTAMP"), D.p("STARTING_BASE_PAGE")
    s.regs[0] = (1, D.p("BOOTLOADER_CALLDATA_PAGE") << 32)
    return s, empty


def memory_query_encode(ts, page, index, rw, is_ptr, value):  # MemoryQuery::encode, src/base_structures/memory_query/mod.rs:103-221
    l = limbs(value)
    b = [(x >> (8 * k)) & 0xFF for x in l[5:8] for k in range(4)]
    return [ts, page, index + (rw << 32) + (is_ptr << 33)] + [l[i] + (b[3 * i] << 32) + (b[3 * i + 1] << 40) + (b[3 * i + 2] << 48) for i in range(4)] + [l[4]]


def decommit_query_encode(code_hash, page, is_first, timestamp):  # DecommitQuery::encode, src/base_structures/decommit_query/mod.rs:33-113
    h = limbs(code_hash)
    p, t = [(page >> (8 * k)) & 0xFF for k in range(4)], [(timestamp >> (8 * k)) & 0xFF for k in range(4)]
    return [h[0] + (p[0] << 32) + (p[1] << 40) + (p[2] << 48), h[1] + (p[3] << 32) + (t[0] << 40) + (t[1] << 48),
            h[2] + (t[2] << 32) + (t[3] << 40) + (is_first << 48)] + h[3:8]


def log_words(address, key, read_value, written_value, rw, aux, rollback, is_service, shard, tx, ts):
    return limbs(address, 5) + limbs(key) + limbs(read_value) + limbs(written_value) + [aux, rw, rollback, is_service, shard, tx, ts]


def push4(tail, enc):
    return zko.queue_tail4_push20(tail, enc)


def push12(tail, enc8):
    return zko.poseidon2_permute(list(enc8) + list(tail[8:12]))


# ------------------------------------------------------------------------------------------------ the synthetic world / WitnessOracle
class World:
    def __init__(self):
        self.memory = {}       # (page, index) -> (value, is_ptr)
        self.storage = {}      # (shard, address, key) -> value
        self.contracts = {}    # address -> (code_hash, [256-bit code words])
        self.decommitted = {}  # code_hash -> page
        self.callstack = []    # (Ctx, sponge before the push)
        self.refund = 0

    def load_code(self, page, opcodes):
        """four 64-bit opcodes per 32-byte word, first opcode in the most significant 8 bytes (pre_state.rs:184-206)"""
        ops = list(opcodes) + [0] * (-len(opcodes) % 4)
        words = []
        for i in range(0, len(ops), 4):
            w = (ops[i] << 192) | (ops[i + 1] << 128) | (ops[i + 2] << 64) | ops[i + 3]
            self.memory[(page, i // 4)] = (w, 0)
            words.append(w)
        return words

    def deploy(self, D: Defs, address, opcodes, shard=0, marker=None):
        words_needed = (len(opcodes) + 3) // 4
        marker = D.p("CODE_AT_REST_MARKER") if marker is None else marker
        body = (0xC0DE0000 + address) & ((1 << 224) - 1)
        code_hash = body | ((words_needed | (marker << 16) | (D.p("CODE_HASH_VERSION_BYTE") << 24)) << 224)
        self.contracts[code_hash & ~(0xFF << 240)] = list(opcodes)   # keyed by the at-rest form (marker byte cleared)
        self.storage[(shard, D.p("DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW"), address)] = code_hash
        return code_hash


class Recorder:
    """pass 1: log / call / ret events for plan_rollbacks; pass 2: the planned answers"""

    def __init__(self, plan=None):
        self.plan = plan
        self.events = []

    def log_prev_head(self):
        i = len([e for e in self.events if e[0] == "write"])
        return self.plan["write"][i] if self.plan else [0] * 4

    def call_tail(self):
        i = len([e for e in self.events if e[0] == "call"])
        return self.plan["call"][i] if self.plan else [0] * 4


def plan_rollbacks(events, final_anchor=None):
    """Walk the recorded events and solve the rollback chains backwards (saved_context.rs:16-35, ret.rs:314-418).
    events: ("fwd", enc20) forward-queue push; ("write", rollback_enc20) revertable log (also pushed forward by its own "fwd");
    ("call",) frame start; ("ret", reverted: bool) frame end.  Returns {"write": [prev_head per write], "call": [tail per call],
    "root_tail": declared tail of the root frame}."""
    n_write = sum(1 for e in events if e[0] == "write")
    n_call = sum(1 for e in events if e[0] == "call")
    plan = {"write": [None] * n_write, "call": [None] * n_call}
    root = {"items": [], "call_idx": None}
    stack = [root]
    fwd = None  # set by the caller through the first event ("init", tail)
    wi = ci = 0

    def finalize(frame, anchor):
        # items in time order: ("w", write index, enc) | ("c", call index, child frame merged) — flattened already
        head = list(anchor)
        for kind, idx, enc in reversed(frame["items"]):
            if kind == "w":
                plan["write"][idx] = list(head)       # the head AFTER this write is the claimed prev_head witness
                head = push4(head, enc)
            else:
                plan["call"][idx] = list(head)        # an ok-returned child: its declared tail = the parent's head at the call
        return head                                    # the frame's own declared tail

    for ev in events:
        if ev[0] == "init":
            fwd = list(ev[1])
        elif ev[0] == "fwd":
            fwd = push4(fwd, ev[1])
        elif ev[0] == "write":
            stack[-1]["items"].append(("w", wi, ev[1]))
            wi += 1
        elif ev[0] == "call":
            stack.append({"items": [], "call_idx": ci})
            ci += 1
        elif ev[0] == "ret":
            frame = stack.pop() if len(stack) > 1 else None
            if frame is None:      # the root frame returns
                if ev[1]:
                    tail = finalize(root, fwd)
                    plan["root_tail"] = tail
                    fwd = tail
                else:
                    plan["root_tail"] = finalize(root, fwd)
                root = {"items": [], "call_idx": None}
                stack = [root]
                plan["root_done"] = True
                continue
            if ev[1]:              # revert / panic: the frame's chain is anchored at the current forward tail
                tail = finalize(frame, fwd)
                plan["call"][frame["call_idx"]] = tail
                fwd = tail
            else:                  # ok: the child's writes continue the parent's chain
                stack[-1]["items"].append(("c", frame["call_idx"], None))
                stack[-1]["items"].extend(frame["items"])
    if not plan.get("root_done"):
        # unfinished frames: merge everything into the root, anchored at an arbitrary value (nothing checks it before the root returns)
        while len(stack) > 1:
            frame = stack.pop()
            stack[-1]["items"].append(("c", frame["call_idx"], None))
            stack[-1]["items"].extend(frame["items"])
        plan["root_tail"] = finalize(root, final_anchor if final_anchor is not None else [0] * 4)
    return plan


# ------------------------------------------------------------------------------------------------ one cycle
def u32_sub(a, b):
    return (a - b) & M32, int(a < b)


def u32_add(a, b):
    return (a + b) & M32, int(a + b > M32)


def vm_cycle(D: Defs, st: VmState, world: World, rec: Recorder, gctx):
    """-> (next state, witness dict name -> words).  Raises AssertionError where the circuit would be unsatisfiable."""
    st = copy.deepcopy(st)
    W = {}
    c = st.ctx
    # ---------------- create_prestate
    skip = int(st.depth == 0)
    pending = st.pending_exception
    should_try_read = (not skip) and (not pending)
    st.pending_exception = 0
    pc = c.pc
    pc_plus_one = (pc + 1) & M16
    super_pc, sub_pc = pc >> 2, pc & 3
    should_read_opcode = should_try_read and not (st.prev_code_page == c.code_page and super_pc == st.prev_super_pc)
    ts0 = st.timestamp
    ts1, ts2, ts3 = ts0 + 1, ts0 + 2, ts0 + 3
    code_word = world.memory.get((c.code_page, super_pc), (0, 0))[0] if should_read_opcode else 0
    W["code_word"] = limbs(code_word)
    calls = W["_oracle_calls"] = []   # the getters answered under execute == true, in call order (witness_oracle.rs:45-91)
    if should_read_opcode:
        calls.append(("memory_read", limbs(code_word), 0))
    if should_read_opcode:
        st.mem_tail = push12(st.mem_tail, memory_query_encode(ts0, c.code_page, super_pc, 0, 0, code_word))
        st.mem_len += 1
    else:
        code_word = st.prev_code_word
    opcode = (code_word >> (64 * (3 - sub_pc))) & ((1 << 64) - 1)
    if skip:
        opcode = D.nop_encoding
    if pending:
        opcode = D.panic_encoding
    st.prev_code_word, st.prev_code_page = code_word, c.code_page
    if not skip:
        c.pc, st.prev_super_pc, st.timestamp = pc_plus_one, super_pc, ts0 + 4
    is_kernel, is_static = c.is_kernel, c.is_static
    callstack_is_full = st.depth == D.p("VM_MAX_STACK_DEPTH")
    of, eq, gt = st.flags
    # ---------------- perform_initial_decoding
    variant, cond = opcode & 0x7FF, (opcode >> 13) & 7
    src_byte, dst_byte = (opcode >> 16) & 0xFF, (opcode >> 24) & 0xFF
    imm0, imm1 = (opcode >> 32) & M16, (opcode >> 48) & M16
    price, props_full = D.prices[variant], D.props[variant]
    cname = {v: k for k, v in D.condition_idx.items()}[cond]
    condition = dict(ALWAYS=1, LT=of, EQ=eq, GT=gt, GE=gt | eq, LE=of | eq, NE=1 - eq, GT_OR_LT=gt | of)[cname]
    aux = props_full >> D.aux_bit0
    requires_kernel, can_static, explicit_panic = (aux >> D.aux_kernel_mode) & 1, (aux >> D.aux_static_ok) & 1, (aux >> D.aux_explicit_panic) & 1
    cost = 0 if skip else price
    ergs_left, out_of_ergs = u32_sub(c.ergs, cost)
    if out_of_ergs:
        ergs_left = 0
    mask_into_panic = explicit_panic or out_of_ergs or (requires_kernel and not is_kernel) or (is_static and not can_static) or callstack_is_full
    props = props_full & D.props_mask
    mask_into_nop = (not mask_into_panic) and not condition
    if mask_into_panic:
        props = D.panic_bitspread & D.props_mask
    elif mask_into_nop:
        props = D.nop_bitspread & D.props_mask
    if mask_into_panic or mask_into_nop:
        src_byte = dst_byte = 0
    fam = [i for i in range(D.type_bits) if (props >> i) & 1]
    assert len(fam) == 1 and fam[0] != FAM["INVALID"], "INVALID opcode bit must never be set after masking"
    fam = fam[0]
    var = lambda name: (props >> (D.variant_bit0 + D.variant_idx[name])) & 1
    flag = lambda name: (props >> (D.flag_bit0 + D.flag_idx[name])) & 1
    src_mode = lambda name: (props >> (D.src_bit0 + MODE[name])) & 1
    dst_mode = lambda name: (props >> (D.dst_bit0 + MODE[name])) & 1
    src0_idx, src1_idx, dst0_idx, dst1_idx = src_byte & 15, src_byte >> 4, dst_byte & 15, dst_byte >> 4
    c.ergs = ergs_left
    preliminary_ergs_left = ergs_left
    reg = lambda idx: st.regs[idx - 1] if idx else (0, 0)
    draft_src0, src1_register = reg(src0_idx), reg(src1_idx)
    src0_reg_lowest, dst0_reg_lowest = draft_src0[1] & M16, reg(dst0_idx)[1] & M16
    stack_page, heap_page, aux_heap_page = c.base_page + 1, c.base_page + 2, c.base_page + 3
    is_nop = fam == FAM["NOP"]
    # resolve_memory_region_and_index_for_source / _for_dest (utils.rs:237-384)
    use_code, use_abs, use_rel, use_pp = src_mode("CODE_PAGE"), src_mode("ABSOLUTE_STACK"), src_mode("STACK_OFFSET"), src_mode("STACK_PUSH_POP")
    idx_abs = (src0_reg_lowest + imm0) & M16
    idx_rel = (c.sp - idx_abs) & M16
    use_stack = use_abs or use_rel or use_pp
    should_read_src0 = int((use_stack or use_code) and not is_nop)
    src0_page = stack_page if use_stack else c.code_page
    src0_index = idx_abs if (use_code or use_abs) else idx_rel
    sp_after_src0 = idx_rel if use_pp else c.sp
    d_abs, d_rel, d_pp = dst_mode("ABSOLUTE_STACK"), dst_mode("STACK_OFFSET"), dst_mode("STACK_PUSH_POP")
    didx_abs = (dst0_reg_lowest + imm1) & M16
    didx_rel_push, didx_rel = (sp_after_src0 + didx_abs) & M16, (sp_after_src0 - didx_abs) & M16
    dst0_in_memory = int((d_abs or d_rel or d_pp) and not is_nop)
    dst0_page = stack_page
    dst0_index = didx_abs if d_abs else (sp_after_src0 if d_pp else didx_rel)
    c.sp = didx_rel_push if d_pp else sp_after_src0
    # may_be_read_memory_for_source_operand (utils.rs:388-522)
    mem_val, mem_ptr = world.memory.get((src0_page, src0_index), (0, 0)) if should_read_src0 else (0, 0)
    W["src0_read_value"], W["src0_read_is_ptr"] = limbs(mem_val), [mem_ptr]
    if should_read_src0:
        calls.append(("memory_read", limbs(mem_val), mem_ptr))
    if should_read_src0:
        st.mem_tail = push12(st.mem_tail, memory_query_encode(ts0, src0_page, src0_index, 0, mem_ptr, mem_val))
        st.mem_len += 1
    src0 = draft_src0 if src_mode("REG_ONLY") else (mem_ptr, mem_val)
    if src_mode("IMM16"):
        src0 = (0, imm0)
    src1 = src1_register
    swap = (fam in (FAM["SUB"], FAM["DIV"], FAM["SHIFT"]) and flag("SWAP_ARITH")) or (fam == FAM["PTR"] and flag("SWAP_PTR"))
    if swap:
        src0, src1 = src1, src0
    erase_mask = ~(((1 << 64) - 1) << 32) & M256  # limbs 1 and 2
    if src0[0] and fam not in (FAM["RET"], FAM["PTR"], FAM["UMA"], FAM["FAR_CALL"]) and not is_kernel:
        src0 = (0, src0[1] & erase_mask)
    if src1[0] and not is_kernel:
        src1 = (0, src1[1] & erase_mask)
    s0p, s0 = src0
    s1p, s1 = src1
    s0l, s1l = limbs(s0), limbs(s1)

    draft = copy.deepcopy(st)   # "draft_vm_state": what the opcodes read
    dc = draft.ctx
    # ---------------- opcodes: diffs of the one that applies
    dst0 = dst1 = None           # (is_ptr, value)
    dst0_may_go_to_memory = False
    new_flags = None
    new_pc = new_ergs = None
    pend = 0
    for n in ("log_pubdata_refund",):
        W[n] = [0]
    W["log_storage_read_value"] = [0] * 8
    W["log_rollback_queue_prev_head"] = [0] * 4
    W["near_call_rollback_queue_tail"] = [0] * 4
    W["far_call_code_hash_read_value"] = [0] * 8
    W["far_call_decommit_suggested_page"] = [0]
    W["far_call_rollback_queue_tail"] = [0] * 4
    W["ret_popped_context"] = [0] * 42
    W["ret_previous_callstack_state"] = [0] * 12
    W["uma_read_a"], W["uma_read_b"] = [0] * 8, [0] * 8
    set_flags = flag("SET_FLAGS")
    can_mem = lambda f: bool(D.can_write_dst0_into_memory[FAM[f]])

    if fam in (FAM["ADD"], FAM["SUB"]):           # add_sub.rs
        if fam == FAM["ADD"]:
            r, o = (s0 + s1) & M256, int(s0 + s1 > M256)
        else:
            r, o = (s0 - s1) & M256, int(s0 < s1)
        dst0, dst0_may_go_to_memory = (0, r), can_mem("ADD")
        if set_flags:
            new_flags = (o, int(r == 0), int(not (o or r == 0)))
    elif fam == FAM["JUMP"]:                      # jump.rs
        new_pc = s0 & M16
    elif fam == FAM["BINOP"]:                     # binop.rs
        r = s0 & s1 if var("BINOP_AND") else (s0 | s1 if var("BINOP_OR") else s0 ^ s1)
        dst0, dst0_may_go_to_memory = (0, r), can_mem("BINOP")
        if set_flags:
            new_flags = (0, int(r == 0), 0)
    elif fam == FAM["CONTEXT"]:                   # context.rs
        if var("CTX_SET_CONTEXT_U128"):
            st.ctx_u128 = s0 & ((1 << 128) - 1)
        elif var("CTX_SET_ERGS_PER_PUBDATA"):
            st.ergs_per_pubdata = s0l[0]
        elif var("CTX_INC_TX_NUMBER"):
            st.tx_number = (draft.tx_number + 1) & M32
        else:
            if var("CTX_META"):
                r = from_limbs([draft.ergs_per_pubdata, 0, dc.heap_bound, dc.aux_heap_bound, 0, 0, 0, dc.this_shard | (dc.caller_shard << 8) | (dc.code_shard << 16)])
            elif var("CTX_CODE_ADDRESS"):
                r = dc.code_address
            elif var("CTX_CALLER"):
                r = dc.caller
            elif var("CTX_THIS"):
                r = dc.this
            elif var("CTX_GET_CONTEXT_U128"):
                r = dc.ctx_u128
            elif var("CTX_ERGS_LEFT"):
                r = preliminary_ergs_left
            else:
                r = dc.sp
            dst0, dst0_may_go_to_memory = (0, r), can_mem("CONTEXT")
    elif fam == FAM["PTR"]:                       # ptr.rs
        panic = (not (s0p and not s1p))
        is_add, is_sub, is_pack, is_shrink = var("PTR_ADD"), var("PTR_SUB"), var("PTR_PACK"), var("PTR_SHRINK")
        panic = panic or ((is_add or is_sub) and (s1 >> 32) != 0) or (is_pack and (s1 & ((1 << 128) - 1)) != 0)
        ra, oa = u32_add(s0l[0], s1l[0])
        rs, us = u32_sub(s0l[0], s1l[0])
        rk, uk = u32_sub(s0l[3], s1l[0])
        panic = panic or (is_add and oa) or (is_sub and us) or (is_shrink and uk)
        if panic:
            pend = 1
        else:
            l = list(s0l)
            if is_add:
                l[0] = ra
            if is_sub:
                l[0] = rs
            if is_shrink:
                l[3] = rk
            if is_pack:
                l = [s0l[0], s0l[1], s0l[2], s0l[3]] + s1l[4:8]
            dst0, dst0_may_go_to_memory = (s0p, from_limbs(l)), can_mem("PTR")
    elif fam in (FAM["MUL"], FAM["DIV"]):         # mul_div.rs
        if fam == FAM["MUL"]:
            full = s0 * s1
            lo, hi = full & M256, full >> 256
            r0, r1 = lo, hi
            fl = (int(hi != 0), int(lo == 0), int(hi == 0 and lo != 0))
        else:
            if s1 == 0:
                q, r = 0, s0
            else:
                q, r = divmod(s0, s1)
            r0, r1 = q, (0 if s1 == 0 else r)
            fl = (int(s1 == 0), int(s1 != 0 and q == 0), int(s1 != 0 and r == 0))
        dst0, dst0_may_go_to_memory, dst1 = (0, r0), can_mem("MUL"), (0, r1)
        if set_flags:
            new_flags = fl
    elif fam == FAM["SHIFT"]:                     # shifts.rs
        shift = s1 & 0xFF
        is_rol, is_ror, is_shr = var("SHIFT_ROL"), var("SHIFT_ROR"), var("SHIFT_SHR")
        full_shift = (256 - shift) if (is_ror and shift) else shift
        is_cyclic, is_right = is_rol or is_ror, is_ror or is_shr
        if is_right and not is_cyclic:
            r = s0 >> full_shift
        else:
            full = s0 << full_shift
            r = (full & M256) + ((full >> 256) if is_cyclic else 0)
        dst0, dst0_may_go_to_memory = (0, r), can_mem("SHIFT")
        if set_flags:
            new_flags = (0, int(r == 0), 0)
    elif fam == FAM["LOG"]:                       # log.rs
        is_read, is_write, is_event, is_l1, is_pre = var("LOG_STORAGE_READ"), var("LOG_STORAGE_WRITE"), var("LOG_EVENT"), var("LOG_TO_L1"), var("LOG_PRECOMPILE_CALL")
        key = list(s0l)
        if is_pre and key[4] == 0:
            key[4] = heap_page
        if is_pre and key[5] == 0:
            key[5] = heap_page
        key = from_limbs(key)
        is_storage = is_read or is_write
        is_revertable = int(not (is_read or is_pre))
        aux_byte = (D.p("STORAGE_AUX_BYTE") if is_storage else 0) + (D.p("EVENT_AUX_BYTE") if is_event else 0) + (D.p("L1_MESSAGE_AUX_BYTE") if is_l1 else 0) + \
            (D.p("PRECOMPILE_AUX_BYTE") if is_pre else 0)
        refund = world.refund
        W["log_pubdata_refund"] = [refund]
        calls.append(("refund", refund))
        burn = 0
        if is_write and dc.this_shard == 0:
            burn = draft.ergs_per_pubdata * (D.p("INITIAL_STORAGE_WRITE_PUBDATA_BYTES") - refund)
        if is_pre:
            burn = s1l[0]
        if is_l1:
            burn = draft.ergs_per_pubdata * D.p("L1_MESSAGE_PUBDATA_BYTES")
        assert burn <= M32
        ergs_rem, not_enough = u32_sub(preliminary_ergs_left, burn)
        if not_enough:
            ergs_rem = 0
        execute = int(not not_enough)
        skey = (dc.this_shard, dc.this, key)
        read_w = world.storage.get(skey, 0) if (execute and is_storage) else 0
        W["log_storage_read_value"] = limbs(read_w)
        if execute and is_storage:
            calls.append(("storage_read", limbs(read_w)))
        read_value = read_w if is_storage else 0
        written = s1 if is_revertable else read_value
        q = log_words(dc.this, key, read_value, written, is_revertable, aux_byte, 0, flag("FIRST_MESSAGE"), dc.this_shard, draft.tx_number, ts1)
        fwd_enc = log_encode(q)
        rb_enc = list(fwd_enc)
        rb_enc[19] = 1
        execute_rollback = execute and is_revertable
        if execute:
            rec.events.append(("fwd", fwd_enc))
            st.ctx  # noqa
            st.fwd_tail, st.fwd_len = push4(draft.fwd_tail, fwd_enc), draft.fwd_len + 1
            if is_write:
                world.storage[skey] = s1
        if execute_rollback:
            prev_head = rec.log_prev_head()
            W["log_rollback_queue_prev_head"] = list(prev_head)
            calls.append(("rollback_queue_witness", list(prev_head)))
            rec.events.append(("write", rb_enc))
            if rec.plan:
                assert push4(prev_head, rb_enc) == list(dc.rq_head), "rollback head claim does not hash to the current head"
            c.rq_head, c.rq_len = list(prev_head), dc.rq_len + 1
        new_ergs = ergs_rem
        if is_read:
            dst0 = (0, read_value)
        elif is_pre:
            dst0 = (0, execute)
    elif fam == FAM["UMA"]:                       # uma.rs
        is_hr, is_hw, is_ar, is_aw, is_fp = var("UMA_HEAP_READ"), var("UMA_HEAP_WRITE"), var("UMA_AUX_HEAP_READ"), var("UMA_AUX_HEAP_WRITE"), var("UMA_FAT_PTR_READ")
        inc = flag("UMA_INCREMENT")
        access_heap, access_aux = is_hr or is_hw, is_ar or is_aw
        not_a_ptr = int(is_fp and not s0p)
        offset, page, start, length = s0l[0:4]
        skip_legit = int(is_fp and not (offset < length))
        formal_start = start if is_fp else 0
        absolute_address = (formal_start + offset) & M32
        incremented_offset, non_addr = u32_add(offset, 32)
        non_addr = non_addr or incremented_offset == M32
        q_panic = not_a_ptr or non_addr
        q_skip = not_a_ptr or skip_legit or non_addr
        boob, ufb = u32_sub(incremented_offset, length)
        boob = 0 if (q_skip or ufb) else boob
        bytes_to_cleanup = boob % 32
        growth = 0
        new_heap_bound = new_aux_bound = None
        if access_heap:
            g0, uf = u32_sub(incremented_offset, dc.heap_bound)
            growth = 0 if uf else g0
            new_heap_bound = dc.heap_bound if uf else incremented_offset
        if access_aux:
            g0, uf = u32_sub(incremented_offset, dc.aux_heap_bound)
            growth = 0 if uf else g0
            new_aux_bound = dc.aux_heap_bound if uf else incremented_offset
        oob = (access_heap or access_aux) and ((s0 >> 32) != 0 or non_addr)
        if oob:
            growth = M32
        ergs_after, ufe = u32_sub(preliminary_ergs_left, growth)
        set_panic = q_panic or ufe or oob
        if ufe:
            ergs_after = 0
        skip_mem = q_skip or set_panic
        cell, unalign = absolute_address // 32, absolute_address % 32
        mem_page = heap_page if access_heap else (aux_heap_page if access_aux else page)
        cell_b = (cell + 1) & M32
        read_a, read_b = int(not skip_mem), int((not skip_mem) and unalign != 0)
        va = world.memory.get((mem_page, cell), (0, 0))[0] if read_a else 0
        vb = world.memory.get((mem_page, cell_b), (0, 0))[0] if read_b else 0
        W["uma_read_a"], W["uma_read_b"] = limbs(va), limbs(vb)
        if read_a:
            calls.append(("memory_read", limbs(va), 0))
        if read_b:
            calls.append(("memory_read", limbs(vb), 0))
        tail, ln = draft.mem_tail, draft.mem_len
        if read_a:
            tail, ln = push12(tail, memory_query_encode(ts0, mem_page, cell, 0, 0, va)), ln + 1
        if read_b:
            tail, ln = push12(tail, memory_query_encode(ts0, mem_page, cell_b, 0, 0, vb)), ln + 1
        buf = va.to_bytes(32, "big") + vb.to_bytes(32, "big")
        word = bytearray(buf[unalign:unalign + 32])
        nclean = bytes_to_cleanup if is_fp else 0
        for k in range(nclean):
            word[31 - k] = 0
        read_value = int.from_bytes(word, "big")
        is_write_access = is_hw or is_aw
        exec_write = int(is_write_access and not skip_mem)
        if exec_write:
            wb = bytearray(buf)
            wb[unalign:unalign + 32] = s1.to_bytes(32, "big")
            na, nb = int.from_bytes(wb[:32], "big"), int.from_bytes(wb[32:], "big")
            tail, ln = push12(tail, memory_query_encode(ts3, mem_page, cell, 1, 0, na)), ln + 1
            world.memory[(mem_page, cell)] = (na, 0)
            if unalign:
                tail, ln = push12(tail, memory_query_encode(ts3, mem_page, cell_b, 1, 0, nb)), ln + 1
                world.memory[(mem_page, cell_b)] = (nb, 0)
        incremented_src0 = (s0p, from_limbs([incremented_offset] + s0l[1:]))
        if set_panic:
            pend = 1
        else:
            if is_write_access and inc:
                dst0 = incremented_src0
            elif not is_write_access:
                dst0 = (0, read_value)
            if (not is_write_access) and inc:
                dst1 = incremented_src0
        if access_heap:
            c.heap_bound = new_heap_bound
        if access_aux:
            c.aux_heap_bound = new_aux_bound
        new_ergs = ergs_after
        st.mem_tail, st.mem_len = tail, ln
    elif fam in (FAM["NEAR_CALL"], FAM["FAR_CALL"], FAM["RET"]):   # call_ret.rs + call_ret_impl
        fwd_byte = (s0 >> (8 * D.p("FAR_CALL_FORWARDING_MODE_BYTE_IDX"))) & 0xFF
        use_aux_heap, forward_fat_pointer = fwd_byte == D.p("FORWARD_USE_AUX_HEAP"), fwd_byte == D.p("FORWARD_FAT_POINTER")
        use_heap = not (use_aux_heap or forward_fat_pointer)
        offset, page, start, length = s0l[0:4]
        end_non_inclusive, range_overflow = u32_add(start, length)
        ptr_invalid = (offset != 0 and not forward_fat_pointer) or range_overflow or length < offset
        fp = (0, 0, 0, 0) if ptr_invalid else (offset, page, start, length)
        upper_bound_abi = end_non_inclusive
        readjust = lambda p: (0, p[1], p[2] + p[0], p[3] - p[0])
        new_ctx = old_ctx = None
        apply_ret = fam == FAM["RET"]
        if fam == FAM["NEAR_CALL"]:               # near_call.rs
            old_ctx = copy.deepcopy(dc)
            old_ctx.pc = pc_plus_one
            new_ctx = copy.deepcopy(old_ctx)
            tail = rec.call_tail()
            W["near_call_rollback_queue_tail"] = list(tail)
            calls.append(("rollback_tail_for_call", list(tail)))
            rec.events.append(("call",))
            new_ctx.rq_tail, new_ctx.rq_head, new_ctx.rq_len = list(tail), list(tail), 0
            passed_abi = s0l[0]
            to_pass = preliminary_ergs_left if passed_abi == 0 else passed_abi
            rem, uf = u32_sub(preliminary_ergs_left, to_pass)
            old_ctx.ergs = 0 if uf else rem
            new_ctx.ergs = preliminary_ergs_left if uf else to_pass
            new_ctx.pc, new_ctx.eh, new_ctx.is_local = imm0, imm1, 1
            new_fwd_tail, new_fwd_len = draft.fwd_tail, draft.fwd_len
        elif fam == FAM["FAR_CALL"]:              # far_call.rs
            is_delegate, is_mimic = var("FAR_DELEGATE"), var("FAR_MIMIC")
            old_ctx = copy.deepcopy(dc)
            old_ctx.pc = pc_plus_one
            new_ctx = Ctx()
            new_ctx.heap_bound = new_ctx.aux_heap_bound = D.p("NEW_FRAME_MEMORY_STIPEND")
            caller_for_mimic = draft.regs[D.p("CALL_IMPLICIT_PARAMETER_REG_IDX")][1] & ((1 << 160) - 1)
            destination = s1 & ((1 << 160) - 1)
            is_static_call, is_call_shard = flag("FAR_CALL_STATIC"), flag("FAR_CALL_SHARD")
            abi_shard = (s0 >> (8 * D.p("FAR_CALL_SHARD_ID_BYTE_IDX"))) & 0xFF
            ctor = int(((s0 >> (8 * D.p("FAR_CALL_CONSTRUCTOR_CALL_BYTE_IDX"))) & 0xFF) != 0)
            syscall = int(((s0 >> (8 * D.p("FAR_CALL_SYSTEM_CALL_BYTE_IDX"))) & 0xFF) != 0)
            caller_shard = dc.this_shard
            dest_shard = abi_shard if is_call_shard else caller_shard
            target_is_zkporter = dest_shard != 0
            target_is_kernel = int((destination >> 16) == 0)
            ctor, syscall = int(ctor and dc.is_kernel), int(syscall and target_is_kernel)
            default_page = draft.page_counter
            st.page_counter = draft.page_counter + D.p("NEW_MEMORY_PAGES_PER_FAR_CALL")
            assert st.page_counter <= M32
            # may_be_read_code_hash
            zkporter_ok, default_aa = gctx
            can_read = (not target_is_zkporter) or zkporter_ok
            should_read = int(can_read)
            needs_porter_mask = target_is_zkporter and not zkporter_ok
            dkey = (dest_shard, D.p("DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW"), destination)
            code_hash = world.storage.get(dkey, 0) if should_read else 0
            W["far_call_code_hash_read_value"] = limbs(code_hash)
            if should_read:
                calls.append(("storage_read", limbs(code_hash)))
            q = log_words(D.p("DEPLOYER_SYSTEM_CONTRACT_ADDRESS_LOW"), destination, code_hash, code_hash, 0, D.p("STORAGE_AUX_BYTE"), 0, 0, dest_shard, draft.tx_number, ts1)
            new_fwd_tail, new_fwd_len = draft.fwd_tail, draft.fwd_len
            if should_read:
                enc = log_encode(q)
                rec.events.append(("fwd", enc))
                new_fwd_tail, new_fwd_len = push4(draft.fwd_tail, enc), draft.fwd_len + 1
            bytecode_hash = code_hash
            empty = bytecode_hash == 0
            mask_default_aa = should_read and empty and not target_is_kernel
            if mask_default_aa:
                bytecode_hash = default_aa
            if needs_porter_mask:
                bytecode_hash = 0
            trivial = (empty and not mask_default_aa) or needs_porter_mask or not should_read
            target_page = 0 if trivial else default_page
            top = bytecode_hash >> 224
            version_byte, marker_byte = (top >> 24) & 0xFF, (top >> 16) & 0xFF
            normal_marker, ctor_marker = marker_byte == 0, marker_byte == D.p("CODE_YET_CONSTRUCTED_MARKER")
            code_format_exception = version_byte != D.p("CODE_HASH_VERSION_BYTE") or not (normal_marker or ctor_marker)
            can_call_code = (normal_marker and not ctor) or (ctor_marker and ctor)
            at_rest = (bytecode_hash & ((1 << 224) - 1)) | (((top & 0xFFFF) | (D.p("CODE_AT_REST_MARKER") << 16) | (D.p("CODE_HASH_VERSION_BYTE") << 24)) << 224)
            masked_hash = at_rest if can_call_code else (0 if target_is_kernel else default_aa)
            code_len_words = 0 if code_format_exception else ((masked_hash >> 224) & M16)
            exceptions = code_format_exception or ((not can_call_code) and target_is_kernel) or (forward_fat_pointer and not s0p) or ptr_invalid or range_overflow
            final_fp = readjust(fp) if forward_fat_pointer else (0, heap_page if use_heap else aux_heap_page, fp[2], fp[3])
            if exceptions:
                final_fp = (0, 0, 0, 0)
            upper = 0 if exceptions else upper_bound_abi
            if range_overflow and not forward_fat_pointer:
                upper = M32
            growth = 0
            if use_heap:
                g0, uf = u32_sub(upper, old_ctx.heap_bound)
                growth = 0 if uf else g0
                old_ctx.heap_bound = old_ctx.heap_bound if uf else upper
            if use_aux_heap:
                g0, uf = u32_sub(upper, old_ctx.aux_heap_bound)
                growth = 0 if uf else g0
                old_ctx.aux_heap_bound = old_ctx.aux_heap_bound if uf else upper
            ergs_after_growth, ufg = u32_sub(preliminary_ergs_left, growth)
            if ufg:
                ergs_after_growth = 0
            exception = exceptions or ufg
            should_decommit0 = not exception
            target_page = target_page if should_decommit0 else 0
            cost = D.p("ERGS_PER_CODE_WORD_DECOMMITTMENT") * code_len_words
            after_dec, ufd = u32_sub(ergs_after_growth, cost)
            should_decommit = should_decommit0 and not ufd
            ergs_rem = after_dec if should_decommit else ergs_after_growth
            suggested = 0
            if should_decommit:
                suggested = world.decommitted.get(masked_hash, target_page)
                if masked_hash not in world.decommitted:
                    world.decommitted[masked_hash] = target_page
                    code = world.contracts.get(masked_hash)
                    if code is not None:
                        world.load_code(target_page, code)
            W["far_call_decommit_suggested_page"] = [suggested]
            if should_decommit:
                calls.append(("decommit_page", suggested))
            is_first = int(target_page == suggested)
            if should_decommit and not is_first:
                ergs_rem = ergs_after_growth
            if should_decommit:
                st.dec_tail = push12(draft.dec_tail, decommit_query_encode(masked_hash, suggested, is_first, ts1))
                st.dec_len = draft.dec_len + 1
            code_memory_page = suggested if should_decommit else D.p("UNMAPPED_PAGE")
            pend = int(exception or ufd)
            tail = rec.call_tail()
            W["far_call_rollback_queue_tail"] = list(tail)
            calls.append(("rollback_tail_for_call", list(tail)))
            rec.events.append(("call",))
            new_ctx.rq_tail, new_ctx.rq_head, new_ctx.rq_len = list(tail), list(tail), 0
            max_passable = (ergs_rem // 64) * 63
            leftover = ergs_rem - max_passable
            passed_abi = s0l[6]
            rfm, ufp = u32_sub(max_passable, passed_abi)
            to_pass = max_passable if ufp else passed_abi
            old_ctx.ergs = leftover if ufp else (leftover + rfm) & M32
            new_ctx.ergs, new_ctx.pc, new_ctx.eh = to_pass, 0, imm0
            new_ctx.is_static = int(is_static_call or old_ctx.is_static)
            new_ctx.is_kernel = old_ctx.is_kernel if is_delegate else target_is_kernel
            new_ctx.code_shard, new_ctx.code_address = dest_shard, destination
            new_ctx.this_shard = caller_shard if is_delegate else dest_shard
            new_ctx.this = old_ctx.this if is_delegate else destination
            new_ctx.caller = caller_for_mimic if is_mimic else (old_ctx.caller if is_delegate else old_ctx.this)
            new_ctx.caller_shard = caller_shard
            new_ctx.code_page, new_ctx.base_page = code_memory_page, default_page
            new_ctx.ctx_u128 = old_ctx.ctx_u128 if is_delegate else draft.ctx_u128
            new_ctx.is_local = 0
            # registers (far_call.rs:1006-1071)
            regs = list(st.regs)
            abi0, abi1 = D.p("CALL_SYSTEM_ABI_REGISTERS_BEGIN"), D.p("CALL_SYSTEM_ABI_REGISTERS_END")
            res0, res1, imp = D.p("CALL_RESERVED_RANGE_BEGIN"), D.p("CALL_RESERVED_RANGE_END"), D.p("CALL_IMPLICIT_PARAMETER_REG_IDX")
            for r in range(NREG):
                if abi0 <= r < abi1:
                    regs[r] = (0, regs[r][1] if syscall else 0)
                if res0 <= r < res1 or r == imp:
                    regs[r] = (0, 0)
            regs[0] = (1, from_limbs(list(final_fp) + [0] * 4))
            regs[1] = (0, ctor + 2 * syscall)
            st.regs = regs
            st.ctx_u128 = 0
        else:                                     # ret.rs
            is_ok, is_revert, is_panic = var("RET_OK"), var("RET_REVERT"), var("RET_PANIC")
            is_local = dc.is_local
            r0p = 0 if is_panic else s0p
            popped, prev_sponge = world.callstack.pop() if world.callstack else (Ctx(), [0] * 12)
            W["ret_popped_context"], W["ret_previous_callstack_state"] = popped.flatten(), list(prev_sponge)
            calls.append(("callstack", popped.flatten(), list(prev_sponge)))
            old_ctx = copy.deepcopy(popped)
            new_ctx = copy.deepcopy(popped)
            is_far_return = not is_local
            exc = (forward_fat_pointer and not r0p and is_far_return) or (forward_fat_pointer and fp[1] < dc.base_page) or is_panic
            fpr = (0, 0, 0, 0) if exc else fp
            fpr = readjust(fpr) if forward_fat_pointer else (0, heap_page if use_heap else aux_heap_page, fpr[2], fpr[3])
            upper = 0 if exc else upper_bound_abi
            if range_overflow and not forward_fat_pointer:
                upper = M32
            growth = 0
            if use_heap and is_far_return:
                g0, uf = u32_sub(upper, dc.heap_bound)
                growth = 0 if uf else g0
            if use_aux_heap and is_far_return:
                g0, uf = u32_sub(upper, dc.aux_heap_bound)
                growth = 0 if uf else g0
            ergs_after, ufg = u32_sub(preliminary_ergs_left, growth)
            if ufg:
                ergs_after = 0
            if is_local:
                ergs_after = preliminary_ergs_left
            non_local_panic = (exc or ufg or is_panic) and is_far_return
            final_fp = (0, 0, 0, 0) if non_local_panic else fpr
            new_ctx.ergs = ergs_after + popped.ergs
            assert new_ctx.ergs <= M32
            if is_local:
                new_ctx.heap_bound, new_ctx.aux_heap_bound = dc.heap_bound, dc.aux_heap_bound
            perform_revert = is_revert or is_panic or non_local_panic
            rec.events.append(("ret", bool(perform_revert)))
            if perform_revert:
                if rec.plan:
                    assert list(dc.rq_head) == list(draft.fwd_tail), "reverting frame: rollback head must equal the forward tail"
                new_fwd_tail, new_fwd_len = list(dc.rq_tail), draft.fwd_len + dc.rq_len
            else:
                if rec.plan:
                    assert list(popped.rq_head) == list(dc.rq_tail), "ok return: the frame's declared tail must be the caller's head"
                new_fwd_tail, new_fwd_len = draft.fwd_tail, draft.fwd_len
                new_ctx.rq_head, new_ctx.rq_len = list(dc.rq_head), popped.rq_len + dc.rq_len
            use_label = flag("RET_TO_LABEL") and is_local
            ok_pc = imm0 if use_label else popped.pc
            eh_pc = imm0 if use_label else dc.eh
            new_ctx.pc = eh_pc if perform_revert else ok_pc
            if is_far_return:
                st.regs = [(1, from_limbs(list(final_fp) + [0] * 4))] + [(0, 0)] * (NREG - 1)
                st.ctx_u128 = 0
            ret_panic_flag = int(is_panic or non_local_panic)
        # merge (call_ret.rs:167-330)
        if apply_ret:
            sponge = list(prev_sponge)
        else:
            sponge = list(draft.sponge)
        enc = old_ctx.encode()
        for r in range(4):
            sponge = zko.poseidon2_permute(enc[8 * r:8 * r + 8] + sponge[8:12])
        if apply_ret:
            assert sponge == list(draft.sponge), "popped context does not hash to the current callstack sponge"
            assert draft.depth >= 1
            st.sponge, st.depth = list(prev_sponge), draft.depth - 1
        else:
            world.callstack.append((copy.deepcopy(old_ctx), list(draft.sponge)))
            st.sponge, st.depth = sponge, draft.depth + 1
        st.ctx = new_ctx
        st.fwd_tail, st.fwd_len = list(new_fwd_tail), new_fwd_len
        new_flags = (ret_panic_flag if apply_ret else 0, 0, 0)
        c = st.ctx

    # ---------------- apply state diffs (cycle.rs:160-616)
    if dst0 is not None:
        if dst0_may_go_to_memory and dst0_in_memory:
            st.mem_tail = push12(draft.mem_tail, memory_query_encode(ts3, dst0_page, dst0_index, 1, dst0[0], dst0[1]))
            st.mem_len = draft.mem_len + 1
            world.memory[(dst0_page, dst0_index)] = (dst0[1], dst0[0])
        elif dst0_idx:
            st.regs[dst0_idx - 1] = dst0
    if dst1_idx:
        # dst1 is written unconditionally from the (possibly empty) dot product (cycle.rs:330,346-347,416-432)
        st.regs[dst1_idx - 1] = dst1 if dst1 is not None else (0, 0)
    if new_pc is not None:
        st.ctx.pc = new_pc
    if new_ergs is not None:
        st.ctx.ergs = new_ergs
    if new_flags is not None:
        st.flags = new_flags
    st.pending_exception = pend
    W["_family"] = fam                      # bookkeeping for the tests (not a stream field)
    W["_masked"] = "panic" if mask_into_panic else ("nop" if mask_into_nop else "")
    return st, W


# ------------------------------------------------------------------------------------------------ a whole run
class VmRun:
    """n_instances x limit cycles of one synthetic execution: expected per-cycle states, input streams, commitments"""

    def __init__(self, D: Defs, make_world, n_cycles, rollback_anchor=None, zkporter=0, default_aa=0, start_state=None):
        self.D = D
        self.gctx = (zkporter, default_aa)
        # pass 1: record the log / call / ret events with dummy rollback answers
        rec = Recorder()
        self._simulate(make_world(), rec, n_cycles, [0] * 4, start_state)
        events = [("init", [0] * 4)] + rec.events
        self.plan = plan_rollbacks(events, rollback_anchor)
        rec2 = Recorder(self.plan)
        self.rows, self.states = self._simulate(make_world(), rec2, n_cycles, self.plan["root_tail"], start_state)
        self.rollback_tail_for_block = self.plan["root_tail"]

    def _simulate(self, world, rec, n_cycles, root_tail, start_state):
        D = self.D
        if start_state is None:
            st, empty = initial_bootloader_state(D, 0, [0] * 12, 0, [0] * 12, root_tail)
            world.callstack.append((empty, [0] * 12))
        else:
            st = copy.deepcopy(start_state)
        rows, states = [], [st]
        for _ in range(n_cycles):
            nxt, W = vm_cycle(D, st, world, rec, self.gctx)
            rows.append((st, W))
            states.append(nxt)
            st = nxt
        return rows, states
