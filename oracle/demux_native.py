"""oracle/demux_native.py — CPU ORACLE (test infrastructure): native restatement of demultiplex_storage_logs_enty_point /
demultiplex_storage_logs_inner (/root/reference/src/demux_log_queue/mod.rs:38-396) and push_with_optimize (:401-442).

[EXT] zkevm_opcode_defs v1.4.1: aux bytes storage/event/l1 message/precompile = 0/1/2/3; precompile formal addresses
keccak256 0x8010, sha256 0x02, ecrecover 0x01."""
from __future__ import annotations

from . import zko
from .storage_native import ZERO_QUERY, encode, fields

NQ = 6
OUTER_WORDS, LOOP_WORDS, CARRIED = 73, 71, 35
ADDR = {3: 0x8010, 4: 0x02, 5: 0x01}


def target_queue(q):
    """index of the output queue (None: no queue), plus whether the cycle is satisfiable"""
    f = fields(q)
    aux, addr = f["aux"], f["address"]
    if aux == 0:
        return (0, True) if f["shard"] == 0 else (None, False)   # porter storage is unreachable (mod.rs:331-338)
    if aux == 1:
        return 1, True
    if aux == 2:
        return 2, True
    if aux == 3:
        for k, a in ADDR.items():
            if addr == [a, 0, 0, 0, 0]:
                return k, True
        return None, True
    return None, False   # not exactly one aux class (mod.rs:383-391)


def empty_fsm():
    return dict(initial=[0] * 9, out=[[0] * 9 for _ in range(NQ)])


def flatten_fsm(f):
    return list(f["initial"]) + [x for q in f["out"] for x in q]


def instance(queries, limit, start_flag=True, fsm_in=None, obs_initial=None):
    """`queries`: what the initial queue still holds at the start of this instance"""
    if start_flag:
        tail = [0] * 4
        for q in queries:
            tail = zko.queue_tail4_push20(tail, encode(q))
        obs_initial = [0] * 4 + tail + [len(queries)]
        fsm_in = empty_fsm()
        f = dict(initial=list(obs_initial), out=[[0] * 9 for _ in range(NQ)])
    else:
        f = dict(initial=list(fsm_in["initial"]), out=[list(q) for q in fsm_in["out"]])
    outer = [int(start_flag)] + list(obs_initial) + flatten_fsm(fsm_in)
    assert len(outer) == OUTER_WORDS
    queries = list(queries)
    head, tail0, length = f["initial"][0:4], f["initial"][4:8], f["initial"][8]
    outs = [dict(head=q[0:4], tail=q[4:8], len=q[8]) for q in f["out"]]
    ok, rows, routed = True, [], [[] for _ in range(NQ)]
    for _ in range(limit):
        carried = head + [length] + [x for o in outs for x in (o["tail"] + [o["len"]])]
        q = list(ZERO_QUERY)
        if length:
            q = queries.pop(0)
            head = zko.queue_tail4_push20(head, encode(q))
            length -= 1
            k, good = target_queue(q)
            ok &= good
            if k is not None:
                outs[k]["tail"] = zko.queue_tail4_push20(outs[k]["tail"], encode(q))
                outs[k]["len"] += 1
                routed[k].append(q)
        rows.append(carried + list(q))
    if length == 0 and head != tail0:
        ok = False
    completed = int(length == 0)
    fsm_out = dict(initial=head + tail0 + [length], out=[o["head"] + o["tail"] + [o["len"]] for o in outs])
    obs_out = [x for q in fsm_out["out"] for x in q] if completed else [0] * (9 * NQ)
    z4 = [0] * 4
    compact = [int(start_flag), completed] + zko.commit_encoding(list(obs_initial)) + \
        (zko.commit_encoding(obs_out) if completed else z4) + \
        (z4 if start_flag else zko.commit_encoding(flatten_fsm(fsm_in))) + \
        (z4 if completed else zko.commit_encoding(flatten_fsm(fsm_out)))
    return dict(outer=outer, rows=rows, fsm_out=fsm_out, completed=completed, satisfiable=ok, routed=routed, rest=queries,
                obs_initial=obs_initial, public_input=zko.commit_encoding(compact))
