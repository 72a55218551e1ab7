"""oracle/ram_native.py — CPU ORACLE (test infrastructure): native restatement of
ram_permutation_entry_point (/root/reference/src/ram_permutation/mod.rs:31-382) computed directly
from the witness, with no constraint system involved.  It yields
  * the input streams the recorded circuit consumes, and
  * every value the circuit must reproduce (final FSM state, 4-element input commitment),
so that recorder + engine are checked against an independent derivation of the same reference code.
"""
from __future__ import annotations

import numpy as np

from . import zko

P = zko.P
BOOTLOADER_HEAP_PAGE = 10  # zkevm_opcode_defs [EXT]
REPS, ENC = 2, 8


def mq(ts, page, index, rw, is_ptr, value):
    """MemoryQueryWitness -> flattened 13 words (src/base_structures/memory_query/mod.rs:52-68)"""
    return [ts, page, index, int(rw), int(is_ptr)] + [(value >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


ZERO_ITEM = [0] * 13


def queue_simulate(items):
    """Full-state queue: returns (states_before_each_push, final_tail). tail' = P([enc, tail[8:]])"""
    tail = [0] * 12
    before = []
    for it in items:
        before.append(tail)
        tail = zko.queue_full_push(tail, zko.memory_query_encode(it))
    return before, tail


def empty_fsm():
    return dict(lhs=[0, 0], rhs=[0, 0], unsorted=[0] * 25, sorted=[0] * 25, prev_sorting_key=[0] * 3,
                prev_full_key=[0] * 2, prev_value=[0] * 8, prev_is_ptr=0, nondet=0)


def flatten_fsm(f):
    return (list(f["lhs"]) + list(f["rhs"]) + list(f["unsorted"]) + list(f["sorted"]) + list(f["prev_sorting_key"]) +
            list(f["prev_full_key"]) + list(f["prev_value"]) + [f["prev_is_ptr"], f["nondet"]])


def instance(unsorted_items, sorted_items, limit, nondet_len, start_flag=True, fsm_in=None,
             obs_unsorted=None, obs_sorted=None, heads=None):
    """One circuit instance processing `unsorted_items`/`sorted_items` (lists of 13-word queries,
    len <= limit).  For a continuation instance pass start_flag=False, the previous instance's
    `fsm_out`, the *global* observable queue states and `heads` = (unsorted_states, sorted_states):
    the queue states before each of this chunk's pops."""
    n = len(unsorted_items)
    assert len(sorted_items) == n and n <= limit
    if start_flag:
        ub, utail = queue_simulate(unsorted_items)
        sb, stail = queue_simulate(sorted_items)
        obs_unsorted = [0] * 12 + utail + [n]
        obs_sorted = [0] * 12 + stail + [n]
        fsm_in = empty_fsm()
        cur_u, cur_s = obs_unsorted, obs_sorted
        ub.append(utail); sb.append(stail)
    else:
        ub, sb = heads
        cur_u, cur_s = fsm_in["unsorted"], fsm_in["sorted"]
    # produce_fs_challenges over the OBSERVABLE tails (mod.rs:111-116)
    fs_in = obs_unsorted[12:24] + [obs_unsorted[24]] + obs_sorted[12:24] + [obs_sorted[24]]
    ch = zko.fs_challenges(fs_in, REPS, ENC + 1)
    lhs = [1, 1] if start_flag else list(fsm_in["lhs"])
    rhs = [1, 1] if start_flag else list(fsm_in["rhs"])
    nondet = 0 if start_flag else fsm_in["nondet"]
    prev_sk, prev_fk = list(fsm_in["prev_sorting_key"]), list(fsm_in["prev_full_key"])
    prev_val, prev_ptr = list(fsm_in["prev_value"]), fsm_in["prev_is_ptr"]
    u_len, s_len = cur_u[24], cur_s[24]
    u_head, s_head = list(cur_u[0:12]), list(cur_s[0:12])
    u_tail, s_tail = list(cur_u[12:24]), list(cur_s[12:24])
    assert u_len == s_len

    loop_rows = []
    ok = True
    for k in range(limit):
        can_pop = u_len != 0
        ui = unsorted_items[k] if k < n else ZERO_ITEM
        si = sorted_items[k] if k < n else ZERO_ITEM
        assert can_pop == (k < n) or not can_pop
        row = [1 if k == 0 else 0] + u_head + [u_len] + s_head + [s_len] + lhs + rhs + [nondet] + prev_sk + prev_fk + prev_val + [prev_ptr] + list(ui) + list(si)
        assert len(row) == 72
        loop_rows.append(row)
        ue, se = zko.memory_query_encode(ui), zko.memory_query_encode(si)
        if can_pop:
            # pop: head' = P([enc, head[8:]])  (symmetry with push)
            u_head = zko.poseidon2_permute(ue + u_head[8:])
            s_head = zko.poseidon2_permute(se + s_head[8:])
            u_len -= 1; s_len -= 1
        ts, page, index, rw, is_ptr = si[0], si[1], si[2], si[3], si[4]
        value = si[5:13]
        if can_pop and ts == 0 and page == BOOTLOADER_HEAP_PAGE and rw and not is_ptr:
            nondet += 1
        # ordering checks (mod.rs:292-357): evaluated natively as assertions
        sk, fk = [ts, index, page], [index, page]
        cur_key = sum(v << (32 * i) for i, v in enumerate(sk))
        prv_key = sum(v << (32 * i) for i, v in enumerate(prev_sk))
        first_of_fresh = start_flag and k == 0
        if can_pop and not first_of_fresh and not (prv_key < cur_key):
            ok = False
        same_cell = fk == prev_fk
        val_zero = all(v == 0 for v in value) and not is_ptr
        val_eq = value == prev_val and prev_ptr == is_ptr
        if first_of_fresh:
            if (not rw) and not val_zero: ok = False
        else:
            if (not same_cell) and (not rw) and not val_zero: ok = False
            if same_cell and (not rw) and not val_eq: ok = False
        prev_sk, prev_fk, prev_val, prev_ptr = sk, fk, list(value), is_ptr
        if can_pop:
            for r in range(REPS):
                lc, rc = ch[r][ENC], ch[r][ENC]
                for i in range(ENC):
                    lc = (lc + ue[i] * ch[r][i]) % P
                    rc = (rc + se[i] * ch[r][i]) % P
                lhs[r] = lhs[r] * lc % P
                rhs[r] = rhs[r] * rc % P
    completed = u_len == 0
    if completed:
        if u_head != u_tail or s_head != s_tail: ok = False  # enforce_consistency
        if lhs != rhs: ok = False
        if nondet != nondet_len: ok = False
    fsm_out = dict(lhs=lhs, rhs=rhs, unsorted=u_head + u_tail + [u_len], sorted=s_head + s_tail + [s_len],
                   prev_sorting_key=prev_sk, prev_full_key=prev_fk, prev_value=prev_val, prev_is_ptr=prev_ptr, nondet=nondet)
    obs_in = list(obs_unsorted) + list(obs_sorted) + [nondet_len]
    c_obs_in = zko.commit_encoding(obs_in)
    c_obs_out = zko.commit_encoding([])
    c_fsm_in = zko.commit_encoding(flatten_fsm(fsm_in))
    c_fsm_out = zko.commit_encoding(flatten_fsm(fsm_out))
    z4 = [0] * 4
    compact = [int(start_flag), int(completed)] + c_obs_in + (c_obs_out if completed else z4) + \
              (z4 if start_flag else c_fsm_in) + (z4 if completed else c_fsm_out)
    commitment = zko.commit_encoding(compact)
    outer_row = [int(start_flag)] + obs_in + flatten_fsm(fsm_in)
    assert len(outer_row) == 121
    return dict(outer=outer_row, loop=loop_rows, fsm_out=fsm_out, completed=completed, commitment=commitment,
                satisfiable=ok, challenges=ch, obs_unsorted=obs_unsorted, obs_sorted=obs_sorted,
                heads=(ub, sb))


def pack_streams(instances, limit):
    """-> (outer_inputs [121, B], loop_inputs [72, B*limit]) lane-minor u64 arrays"""
    B = len(instances)
    outer = np.array([inst["outer"] for inst in instances], dtype=np.uint64).T.copy()
    loop = np.array([row for inst in instances for row in inst["loop"]], dtype=np.uint64).T.copy()
    assert outer.shape == (121, B) and loop.shape == (72, B * limit)
    return outer, loop


def random_ram_witness(rng: np.random.Generator, n_items: int, n_cells: int = 16, write_frac=0.3,
                       nondet_writes: int = 1):
    """SURVEY §8d C1-style synthetic memory trace: random accesses to `n_cells` cells with
    read-after-write consistency; returns (unsorted, sorted, nondet_len)."""
    cells = [(int(rng.integers(11, 1 << 20)), int(rng.integers(0, 1 << 16))) for _ in range(n_cells)]
    cells = list(dict.fromkeys(cells))
    items = []
    for j in range(nondet_writes):  # bootloader heap writes at timestamp 0
        items.append((0, BOOTLOADER_HEAP_PAGE, 600 + j, 1, 0, int(rng.integers(1, 1 << 62))))
    ts = 1
    while len(items) < n_items:
        page, index = cells[int(rng.integers(0, len(cells)))]
        rw = rng.random() < write_frac
        val = int.from_bytes(rng.bytes(32), "little") if rw else None
        items.append((ts, page, index, int(rw), 0, val))
        ts += int(rng.integers(1, 4))
    # resolve read values in (page, index, ts) order
    order = sorted(range(len(items)), key=lambda i: (items[i][1], items[i][2], items[i][0]))
    resolved = [None] * len(items)
    last = {}
    for i in order:
        ts_, page, index, rw, ptr, val = items[i]
        key = (page, index)
        if rw:
            last[key] = val
        else:
            val = last.get(key, 0)
        resolved[i] = mq(ts_, page, index, rw, ptr, val)
    perm = rng.permutation(len(items))
    unsorted = [resolved[i] for i in perm]
    sorted_ = [resolved[i] for i in order]
    return unsorted, sorted_, nondet_writes
