"""oracle/keccak_native.py — CPU ORACLE (test infrastructure): native restatement of
keccak256_round_function_entry_point / keccak256_precompile_inner
(/root/reference/src/keccak256_round_function/mod.rs:155-670, 672-794), ByteBuffer
(src/keccak256_round_function/buffer/mod.rs:42-163) and trivial_mapping_function (mod.rs:100-142).
Produces the circuit's input streams, the memory queries the FSM must push, the FSM state before every
cycle and the public input commitment.

[EXT] zkevm_opcode_defs v1.4.1: PRECOMPILE_AUX_BYTE = 3, KECCAK256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS = 0x8010,
PrecompileCallABI::to_u256 limb order (pinned by from_encoding, mod.rs:68-75).
"""
from __future__ import annotations

from . import zko
from .ram_native import mq
from .storage_native import ZERO_QUERY, encode, log_query

PRECOMPILE_AUX_BYTE = 3
KECCAK_ADDRESS = 0x8010
RATE, BUF, READS = 136, 192, 6
OUTER_WORDS, LOOP_WORDS, CARRIED = 474, 507, 423


def bytes_to_u256_words(data: bytes, unalignment: int):
    """the reference test's memory image (mod.rs:968-998): 0xff filler before the input, zeros after it"""
    padded = b"\xff" * unalignment + data
    words = []
    for i in range(0, len(padded), 32):
        words.append(int.from_bytes(padded[i:i + 32].ljust(32, b"\0"), "big"))
    return words


def request(data: bytes, timestamp, input_page, input_offset, output_page, output_offset, address=KECCAK_ADDRESS,
            aux_byte=PRECOMPILE_AUX_BYTE):
    key = input_offset | (len(data) << 32) | (output_offset << 64) | (1 << 96) | (input_page << 128) | (output_page << 160)
    q = log_query(address=address, key=key, aux_byte=aux_byte, rw_flag=1, timestamp=timestamp)
    return dict(query=q, reads=bytes_to_u256_words(data, input_offset % 32))


def empty_fsm():
    return dict(rpc=0, ruw=0, padding=0, completed=0, state=[0] * 25, ts_read=0, ts_write=0, params=[0] * 6, buffer=[0] * BUF,
                filled=0, req=[0] * 9, mem=[0] * 25)


def lane_bytes(lane):
    return [(lane >> (8 * k)) & 0xFF for k in range(8)]


def flatten_fsm(f):
    """reference order: keccak_internal_state[i][j][k] with lane x=i, y=j (input.rs:34)"""
    st = [b for i in range(5) for j in range(5) for b in lane_bytes(f["state"][i + 5 * j])]
    return [f["rpc"], f["ruw"], f["padding"], f["completed"]] + st + [f["ts_read"], f["ts_write"]] + list(f["params"]) + \
        list(f["buffer"]) + [f["filled"]] + list(f["req"]) + list(f["mem"])


def limbs(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def fill_with_bytes(buffer, filled, be, offset, meaningful):
    shifted = list(be[offset:]) + [0] * offset
    if meaningful:
        for idx in range(32):
            src = shifted[idx] if idx < meaningful else 0
            if filled + idx < BUF:
                buffer[filled + idx] = src
    return filled + meaningful


def instance(requests, limit, start_flag=True, fsm_in=None, obs_req=None, obs_mem=None, pending=None):
    if start_flag:
        tail = [0] * 4
        for r in requests:
            tail = zko.queue_tail4_push20(tail, encode(r["query"]))
        obs_req = [0] * 4 + tail + [len(requests)]
        obs_mem = obs_mem or [0] * 25
        fsm_in = empty_fsm()
        f = dict(empty_fsm(), rpc=1, req=list(obs_req), mem=list(obs_mem))
    else:
        f = {k: (list(v) if isinstance(v, list) else v) for k, v in fsm_in.items()}
    outer = [int(start_flag)] + list(obs_req) + list(obs_mem) + flatten_fsm(fsm_in)
    assert len(outer) == OUTER_WORDS
    requests, pending = list(requests), list(pending or [])
    rpc, ruw, padding_round, completed = f["rpc"], f["ruw"], f["padding"], f["completed"]
    state, ts_read, ts_write = list(f["state"]), f["ts_read"], f["ts_write"]
    input_page, byte_offset, byte_length, output_page, output_word_offset, needs_full = f["params"]
    buffer, filled = list(f["buffer"]), f["filled"]
    req_head, req_tail, req_len = f["req"][0:4], f["req"][4:8], f["req"][8]
    mem_head, mem_tail, mem_len = f["mem"][0:12], f["mem"][12:24], f["mem"][24]
    ok = True
    if rpc and req_len == 0:  # can_finish_immediatelly (mod.rs:200-213)
        rpc, ruw, completed = 0, 0, 1
    rows, pushed = [], []

    def push(q):
        nonlocal mem_tail, mem_len
        pushed.append(q)
        mem_tail = zko.queue_full_push(mem_tail, zko.memory_query_encode(q))
        mem_len += 1

    for _ in range(limit):
        carried = [rpc, ruw, padding_round, completed] + [b for lane in state for b in lane_bytes(lane)] + \
            [ts_read, ts_write, input_page, byte_offset, byte_length, output_page, output_word_offset, needs_full] + \
            buffer + [filled] + req_head + [req_len] + mem_tail + [mem_len]
        assert len(carried) == CARRIED
        call = list(ZERO_QUERY)
        if rpc:
            if req_len == 0:
                ok = False
            else:
                r = requests.pop(0)
                call, pending = r["query"], list(r["reads"])
                req_head = zko.queue_tail4_push20(req_head, encode(call))
                req_len -= 1
            if call[29] != PRECOMPILE_AUX_BYTE or call[0:5] != [KECCAK_ADDRESS, 0, 0, 0, 0]:
                ok = False
        key = call[5:13]
        call_length = key[1]
        if rpc:
            byte_offset, byte_length, output_word_offset, input_page, output_page = key[0], key[1], key[2], key[4], key[5]
            needs_full = 1 if call_length % RATE == 0 else 0
            ts_read = call[35]
            ts_write = ts_read + 1
        reset_buffer = rpc or completed
        if rpc and call_length == 0:
            padding_round = 1
        if rpc and call_length != 0:
            ruw = 1
        rpc = 0
        if reset_buffer:
            buffer, filled, state = [0] * BUF, 0, [0] * 25
        values = []
        for _r in range(READS):
            unalignment, aligned = byte_offset % 32, byte_offset // 32
            at_most = 32 - unalignment
            meaningful = byte_length if byte_length < at_most else at_most
            should_read = meaningful != 0 and filled + meaningful <= BUF and ruw
            v = pending.pop(0) if (should_read and pending) else 0
            values.append(v)
            if should_read:
                push(mq(ts_read, input_page, aligned, 0, 0, v))
                byte_offset += meaningful
                byte_length -= meaningful
            filled = fill_with_bytes(buffer, filled, v.to_bytes(32, "big"), unalignment, meaningful if should_read else 0)
        zero_bytes_left = byte_length == 0
        currently_filled = filled
        block = buffer[:RATE]
        buffer = buffer[RATE:] + [0] * RATE
        filled = filled - RATE if filled >= RATE else 0
        buffer_now_empty = filled == 0
        apply_padding = zero_bytes_left and buffer_now_empty and ruw and not needs_full
        if apply_padding:
            if currently_filled < RATE - 1:
                block[currently_filled] = 0x01
            block[RATE - 1] = 0x81 if currently_filled == RATE - 1 else 0x80
        if padding_round:
            block = [0x01] + [0] * (RATE - 2) + [0x80]
        for j in range(RATE):
            state[j // 8] ^= block[j] << (8 * (j % 8))
        state = zko.keccak_f1600(state)
        write_result = apply_padding or padding_round
        if write_result:
            digest = b"".join(state[i].to_bytes(8, "little") for i in range(4))
            push(mq(ts_write, output_page, output_word_offset, 1, 0, int.from_bytes(digest, "big")))
        input_is_empty = req_len == 0
        rpc = 1 if (write_result and not input_is_empty) else 0
        completed = 1 if ((write_result and input_is_empty) or completed) else 0
        padding_round = 1 if (ruw and zero_bytes_left and buffer_now_empty and needs_full) else 0
        ruw = 0 if (rpc or padding_round or completed) else 1
        row = carried + list(call)
        for v in values:
            row += limbs(v)
        rows.append(row)
    if req_len == 0 and req_head != req_tail:
        ok = False
    fsm_out = dict(rpc=rpc, ruw=ruw, padding=padding_round, completed=completed, state=state, ts_read=ts_read, ts_write=ts_write,
                   params=[input_page, byte_offset, byte_length, output_page, output_word_offset, needs_full],
                   buffer=buffer, filled=filled, req=req_head + req_tail + [req_len], mem=mem_head + mem_tail + [mem_len])
    done = completed
    z4 = [0] * 4
    compact = [int(start_flag), done] + zko.commit_encoding(list(obs_req) + list(obs_mem)) + \
        (zko.commit_encoding(fsm_out["mem"]) if done else z4) + \
        (z4 if start_flag else zko.commit_encoding(flatten_fsm(fsm_in))) + \
        (z4 if done else zko.commit_encoding(flatten_fsm(fsm_out)))
    return dict(outer=outer, rows=rows, pushed=pushed, fsm_out=fsm_out, satisfiable=ok, obs_req=obs_req, obs_mem=obs_mem,
                public_input=zko.commit_encoding(compact), rest=(requests, pending), memory_state=fsm_out["mem"])
