"""oracle/sha256_native.py — CPU ORACLE (test infrastructure): native restatement of
sha256_round_function_entry_point / sha256_precompile_inner
(/root/reference/src/sha256_round_function/mod.rs:88-340, 347-468).  Produces the circuit's input streams,
the memory queries the FSM must push, the FSM state before every cycle and the public input commitment.

[EXT] zkevm_opcode_defs v1.4.1: PRECOMPILE_AUX_BYTE = 3, SHA256_ROUND_FUNCTION_PRECOMPILE_FORMAL_ADDRESS = 0x02.
"""
from __future__ import annotations

from . import zko
from .ram_native import mq
from .storage_native import ZERO_QUERY, encode, log_query

IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
PRECOMPILE_AUX_BYTE = 3
SHA256_ADDRESS = 0x02
OUTER_WORDS, LOOP_WORDS, CARRIED = 87, 112, 60


def sha_pad(msg: bytes) -> bytes:
    p = bytearray(msg) + b"\x80"
    while len(p) % 64 != 56:
        p.append(0)
    return bytes(p + (8 * len(msg)).to_bytes(8, "big"))


def request(msg: bytes, timestamp, input_page, input_offset, output_page, output_offset, address=SHA256_ADDRESS,
            aux_byte=PRECOMPILE_AUX_BYTE):
    """one precompile call: the LogQuery (ABI in `key`, mod.rs:62-80) and the memory words it reads"""
    data = sha_pad(msg)
    rounds = len(data) // 64
    key = input_offset | (output_offset << 64) | (input_page << 128) | (output_page << 160) | (rounds << 192)
    q = log_query(address=address, key=key, aux_byte=aux_byte, rw_flag=1, timestamp=timestamp)
    reads = [int.from_bytes(data[32 * i:32 * i + 32], "big") for i in range(2 * rounds)]
    return dict(query=q, reads=reads, rounds=rounds)


def empty_fsm():
    return dict(rpc=0, rwfr=0, completed=0, state=list(IV), ts_read=0, ts_write=0, params=[0] * 5, req=[0] * 9, mem=[0] * 25)


def flatten_fsm(f):
    return [f["rpc"], f["rwfr"], f["completed"]] + list(f["state"]) + [f["ts_read"], f["ts_write"]] + list(f["params"]) + \
        list(f["req"]) + list(f["mem"])


def limbs(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def state_bytes(state):
    return [(w >> (8 * k)) & 0xFF for w in state for k in range(4)]


def instance(requests, limit, start_flag=True, fsm_in=None, obs_req=None, obs_mem=None, pending=None):
    """One circuit instance.  `requests`: what the request queue still holds (list of `request`), `pending`:
    reads left of the request in flight (continuation instances).  Returns the streams, the pushed memory
    queries, the final FSM and the public input; `rest` = (requests, pending) left for the next instance."""
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
    requests = list(requests)
    pending = list(pending or [])
    rpc, rwfr, completed = f["rpc"], f["rwfr"], f["completed"]
    state, ts_read, ts_write = list(f["state"]), f["ts_read"], f["ts_write"]
    input_page, input_offset, output_page, output_offset, num_rounds = f["params"]
    req_head, req_tail, req_len = f["req"][0:4], f["req"][4:8], f["req"][8]
    mem_head, mem_tail, mem_len = f["mem"][0:12], f["mem"][12:24], f["mem"][24]
    ok = True
    # can_finish_immediatelly (mod.rs:120-137)
    can_finish = rpc and req_len == 0
    if can_finish:
        rpc, rwfr, completed = 0, 0, 1
    rows, pushed = [], []
    for _ in range(limit):
        carried = [rpc, rwfr, completed] + state_bytes(state) + [ts_read, ts_write, input_page, input_offset, output_page,
                                                                 output_offset, num_rounds] + req_head + [req_len] + mem_tail + [mem_len]
        assert len(carried) == CARRIED
        call = list(ZERO_QUERY)
        if rpc:
            if req_len == 0:
                ok = False
            else:
                r = requests.pop(0)
                call = r["query"]
                pending = list(r["reads"])
                req_head = zko.queue_tail4_push20(req_head, encode(call))
                req_len -= 1
            if call[29] != PRECOMPILE_AUX_BYTE or call[0:5] != [SHA256_ADDRESS, 0, 0, 0, 0]:
                ok = False
            key = call[5:13]
            input_offset, output_offset, input_page, output_page, num_rounds = key[0], key[2], key[4], key[5], key[6]
            ts_read = call[35]
            ts_write = ts_read + 1
        reset_buffer = rpc or completed
        rwfr = 1 if (rpc or rwfr) else 0
        rpc = 0
        should_read = num_rounds != 0
        values = []
        for _r in range(2):
            v = pending.pop(0) if (should_read and pending) else 0
            values.append(v)
            if should_read:
                q = mq(ts_read, input_page, input_offset, 0, 0, v)
                pushed.append(q)
                mem_tail = zko.queue_full_push(mem_tail, zko.memory_query_encode(q))
                mem_len += 1
            if rwfr:
                input_offset += 1
        if rwfr:
            num_rounds = (num_rounds - 1) % zko.P
        if reset_buffer:
            state = list(IV)
        block = b"".join(v.to_bytes(32, "big") for v in values)
        state = zko.sha256_compress(state, block)
        write_result = rwfr and num_rounds == 0
        if write_result:
            digest = int.from_bytes(b"".join(w.to_bytes(4, "big") for w in state), "big")
            q = mq(ts_write, output_page, output_offset, 1, 0, digest)
            pushed.append(q)
            mem_tail = zko.queue_full_push(mem_tail, zko.memory_query_encode(q))
            mem_len += 1
        input_is_empty = req_len == 0
        nothing_left = write_result and input_is_empty
        rpc = 1 if (write_result and not input_is_empty) else 0
        completed = 1 if (nothing_left or completed) else 0
        rwfr = 0 if (rpc or completed) else 1
        rows.append(carried + list(call) + limbs(values[0]) + limbs(values[1]))
    if req_len == 0 and req_head != req_tail:
        ok = False
    fsm_out = dict(rpc=rpc, rwfr=rwfr, completed=completed, state=state, ts_read=ts_read, ts_write=ts_write,
                   params=[input_page, input_offset, output_page, output_offset, num_rounds],
                   req=req_head + req_tail + [req_len], mem=mem_head + mem_tail + [mem_len])
    done = completed
    obs_out = fsm_out["mem"] if done else [0] * 25
    z4 = [0] * 4
    compact = [int(start_flag), done] + zko.commit_encoding(list(obs_req) + list(obs_mem)) + \
        (zko.commit_encoding(obs_out) if done else z4) + \
        (z4 if start_flag else zko.commit_encoding(flatten_fsm(fsm_in))) + \
        (z4 if done else zko.commit_encoding(flatten_fsm(fsm_out)))
    return dict(outer=outer, rows=rows, pushed=pushed, fsm_out=fsm_out, satisfiable=ok, obs_req=obs_req, obs_mem=obs_mem,
                public_input=zko.commit_encoding(compact), rest=(requests, pending), memory_state=fsm_out["mem"])
