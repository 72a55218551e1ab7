"""oracle/storage_native.py — CPU ORACLE (test infrastructure): native restatement of
sort_and_deduplicate_storage_access_entry_point / _inner
(/root/reference/src/storage_validity_by_grand_product/mod.rs:166-897), LogQuery::encode
(src/base_structures/log_query/mod.rs:121-517) and the 4-element-tail queue rules
(src/main_vm/opcodes/log.rs:508-609).  Produces the circuit's input streams and every value the
recorded circuit must reproduce; `satisfiable` mirrors every enforcement of the reference code.
"""
from __future__ import annotations

import numpy as np

from . import zko

P = zko.P
REPS, ENC, NCH = 2, 20, 21


def u256_limbs(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def addr_limbs(n):
    """UInt160 limbs of Address::from_low_u64_le(n) — [EXT]: the byte->limb convention of boojum's UInt160 is
    not in the tree; any injective map preserves the logic under test, we use the integer n itself."""
    return [(n >> (32 * i)) & 0xFFFFFFFF for i in range(5)]


def log_query(address=0, key=0, read_value=0, written_value=0, rw_flag=0, aux_byte=0, rollback=0, is_service=0, shard_id=0,
              tx_number_in_block=0, timestamp=0):
    """flattened 36 words (mod.rs:60-99)"""
    return addr_limbs(address) + u256_limbs(key) + u256_limbs(read_value) + u256_limbs(written_value) + \
        [aux_byte, int(rw_flag), int(rollback), int(is_service), shard_id, tx_number_in_block, timestamp]


ZERO_QUERY = [0] * 36


def fields(q):
    return dict(address=q[0:5], key=q[5:13], read=q[13:21], written=q[21:29], aux=q[29], rw=q[30], rollback=q[31],
                is_service=q[32], shard=q[33], tx=q[34], ts=q[35])


def encode(q):
    f = fields(q)
    bs = [(l >> (8 * k)) & 0xFF for l in f["key"] for k in range(4)] + [(l >> (8 * k)) & 0xFF for l in f["address"] for k in range(4)]
    base = f["read"] + f["written"] + [f["ts"]]
    v = [base[i] + (bs[3 * i] << 32) + (bs[3 * i + 1] << 40) + (bs[3 * i + 2] << 48) for i in range(17)]
    v.append(f["tx"] + (bs[51] << 32) + (f["aux"] << 40) + (f["shard"] << 48))
    v.append(f["rw"] + 2 * f["is_service"])
    v.append(f["rollback"])
    assert all(x < P for x in v)
    return v


def encode_timestamped(q, ts):
    e = encode(q)
    e[19] = e[19] + (ts << 8)
    return e


def queue4_simulate(encodings):
    """-> (state before each push, final tail)"""
    tail, before = [0] * 4, []
    for e in encodings:
        before.append(tail)
        tail = zko.queue_tail4_push20(tail, e)
    return before, tail


def empty_fsm():
    return dict(lhs=[0, 0], rhs=[0, 0], unsorted=[0] * 9, sorted=[0] * 9, final=[0] * 9, cycle_idx=0, prev_packed_key=[0] * 13,
                prev_key=[0] * 8, prev_address=[0] * 5, prev_timestamp=0, has_read=0, base=[0] * 8, current=[0] * 8, depth=0)


def flatten_fsm(f):
    return (list(f["lhs"]) + list(f["rhs"]) + list(f["unsorted"]) + list(f["sorted"]) + list(f["final"]) + [f["cycle_idx"]] +
            list(f["prev_packed_key"]) + list(f["prev_key"]) + list(f["prev_address"]) + [f["prev_timestamp"], f["has_read"]] +
            list(f["base"]) + list(f["current"]) + [f["depth"]])


def instance(unsorted, sorted_records, limit, shard_id=0, enforce_permutation=True):
    """Fresh (start_flag = true) instance.  unsorted: list of 36-word queries; sorted_records: list of
    (36-word query, timestamp).  len <= limit."""
    n = len(unsorted)
    assert len(sorted_records) == n and n <= limit
    u_enc = [encode(q) for q in unsorted]
    s_enc = [encode_timestamped(q, t) for q, t in sorted_records]
    ub, utail = queue4_simulate(u_enc)
    sb, stail = queue4_simulate(s_enc)
    obs_unsorted = [0] * 4 + utail + [n]
    obs_sorted = [0] * 4 + stail + [n]
    fsm_in = empty_fsm()
    ch = zko.fs_challenges(utail + [n] + stail + [n], REPS, NCH)
    ok = True
    lhs, rhs = [1, 1], [1, 1]
    cycle_idx = 0
    u_head, s_head, u_len, s_len = [0] * 4, [0] * 4, n, n
    f_tail, f_len = [0] * 4, 0
    prev_packed, prev_key, prev_addr, prev_ts = [0] * 13, [0] * 8, [0] * 5, 0
    has_read, base, cur, depth = 0, [0] * 8, [0] * 8, 0
    prev_trivial = 1  # no_work OR is_start
    final_items = []
    rows = []

    def push_final(address, key, base_v, cur_v, should_write):
        nonlocal f_tail, f_len
        q = address + key + base_v + cur_v + [0, int(should_write), 0, 0, shard_id, 0, 0]
        final_items.append(q)
        f_tail = zko.queue_tail4_push20(f_tail, encode(q))
        f_len += 1

    for k in range(limit):
        row = [1 if k == 0 else 0, prev_trivial] + lhs + rhs + [cycle_idx] + u_head + [u_len] + s_head + [s_len] + f_tail + [f_len] + \
            prev_packed + prev_key + prev_addr + [prev_ts, has_read] + base + cur + [depth]
        assert len(row) == 67
        uq = unsorted[k] if k < n else ZERO_QUERY
        sq, sts = sorted_records[k] if k < n else (ZERO_QUERY, 0)
        rows.append(row + list(uq) + list(sq) + [sts])
        original_ts = cycle_idx
        cycle_idx += 1
        should_pop = u_len != 0
        trivial = not should_pop
        ue, se = encode(uq), encode_timestamped(sq, sts)
        if should_pop:
            u_head = zko.queue_tail4_push20(u_head, ue)   # same chain rule as push (symmetry)
            s_head = zko.queue_tail4_push20(s_head, se)
            u_len -= 1; s_len -= 1
        rec = fields(sq)
        if should_pop and rec["shard"] != shard_id:
            ok = False
        if should_pop:
            ext = list(ue); ext[19] = ext[19] + (original_ts << 8)
            for r in range(REPS):
                lc = rc = ch[r][ENC]
                for i in range(ENC):
                    lc = (lc + ext[i] * ch[r][i]) % P
                    rc = (rc + se[i] * ch[r][i]) % P
                lhs[r] = lhs[r] * lc % P
                rhs[r] = rhs[r] * rc % P
        packed = rec["key"] + rec["address"]
        as_int = lambda limbs: sum(v << (32 * i) for i, v in enumerate(limbs))
        keys_equal = packed == prev_packed
        prev_greater = as_int(prev_packed) > as_int(packed)
        if not trivial and prev_greater:
            ok = False
        if keys_equal and not trivial and not (prev_ts < sts):
            ok = False
        # new cell
        if k == 0 and should_pop and keys_equal:  # is_start: first item must open a new cell
            ok = False
        unchanged = cur == base
        issue_read = bool(has_read) or (unchanged and depth != 0)
        should_write = not unchanged
        if (not prev_trivial) and (not keys_equal) and (issue_read or should_write):
            push_final(prev_addr, prev_key, base, cur, should_write)
        new_cell = (not trivial) and (not keys_equal)
        same_cell = (not trivial) and keys_equal
        if new_cell:
            base = list(rec["read"])
            cur = list(rec["written"] if rec["rw"] else rec["read"])
            depth = 1 if rec["rw"] else 0
            has_read = 0 if rec["rw"] else 1
        # same cell (evaluated on the UPDATED state, as the circuit does)
        read_same = same_cell and not rec["rw"]
        write_same = same_cell and rec["rw"]
        w_norb, w_rb = write_same and not rec["rollback"], write_same and rec["rollback"]
        if w_norb:
            depth += 1
        if w_rb:
            depth -= 1
            if depth < 0:
                ok = False  # decrement_unchecked would leave the u32 range
        if (read_same or w_norb) and cur != rec["read"]:
            ok = False
        if w_norb:
            cur = list(rec["written"])
        if w_rb:
            cur = list(rec["read"])
        if depth == 0 and read_same:
            base = list(rec["read"])
            has_read = 1
        prev_addr, prev_key, prev_trivial, prev_ts, prev_packed = list(rec["address"]), list(rec["key"]), int(trivial), sts, packed
    # finalisation (mod.rs:836-880)
    exhausted = u_len == 0
    unchanged = cur == base
    issue_read = bool(has_read) or (unchanged and depth != 0)
    should_write = not unchanged
    if (not prev_trivial) and (issue_read or should_write) and exhausted:
        push_final(prev_addr, prev_key, base, cur, should_write)
    if exhausted:
        has_read = 0
    completed = u_len == 0 and s_len == 0
    if u_len == 0 and u_head != utail: ok = False
    if s_len == 0 and s_head != stail: ok = False
    if (u_len == 0) != (s_len == 0): ok = False
    permutation_ok = lhs == rhs
    if completed and enforce_permutation and not permutation_ok:
        ok = False
    fsm_out = dict(lhs=lhs, rhs=rhs, unsorted=u_head + utail + [u_len], sorted=s_head + stail + [s_len], final=[0] * 4 + f_tail + [f_len],
                   cycle_idx=cycle_idx, prev_packed_key=prev_packed, prev_key=prev_key, prev_address=prev_addr, prev_timestamp=prev_ts,
                   has_read=has_read, base=base, current=cur, depth=depth)
    obs_in = [shard_id] + obs_unsorted + obs_sorted
    obs_out = ([0] * 4 + f_tail + [f_len]) if completed else [0] * 9
    c_obs_in, c_obs_out = zko.commit_encoding(obs_in), zko.commit_encoding(obs_out)
    c_fsm_in, c_fsm_out = zko.commit_encoding(flatten_fsm(fsm_in)), zko.commit_encoding(flatten_fsm(fsm_out))
    z4 = [0] * 4
    compact = [1, int(completed)] + c_obs_in + (c_obs_out if completed else z4) + z4 + (z4 if completed else c_fsm_out)
    commitment = zko.commit_encoding(compact)
    outer = [1] + obs_in + flatten_fsm(fsm_in)
    assert len(outer) == 97 and all(len(r) == 140 for r in rows)
    return dict(outer=outer, loop=rows, fsm_out=fsm_out, completed=completed, commitment=commitment, satisfiable=ok,
                permutation_ok=permutation_ok, final_items=final_items)


def pack_streams(instances, limit):
    B = len(instances)
    outer = np.array([i["outer"] for i in instances], dtype=np.uint64).T.copy()
    loop = np.array([row for i in instances for row in i["loop"]], dtype=np.uint64).T.copy()
    assert outer.shape == (97, B) and loop.shape == (140, B * limit)
    return outer, loop


def random_storage_witness(rng, n_items, n_cells=5, shard_id=0):
    """Synthetic storage log (SURVEY §8d C4 style): reads / writes / rollbacks over `n_cells` slots with correct
    read values; returns (unsorted queries, sorted (query, timestamp) records)."""
    cells = [(int(rng.integers(1, 1 << 40)), int.from_bytes(rng.bytes(32), "little")) for _ in range(n_cells)]
    state = {c: 0 for c in cells}
    history = {c: [] for c in cells}   # stack of previous values for rollbacks
    items = []
    for t in range(n_items):
        c = cells[int(rng.integers(0, len(cells)))]
        r = rng.random()
        if r < 0.35:
            q = dict(rw_flag=0, read_value=state[c], written_value=state[c], rollback=0)
        elif r < 0.85 or not history[c]:
            new = int.from_bytes(rng.bytes(32), "little")
            q = dict(rw_flag=1, read_value=state[c], written_value=new, rollback=0)
            history[c].append(state[c]); state[c] = new
        else:
            old = history[c].pop()
            q = dict(rw_flag=1, read_value=old, written_value=state[c], rollback=1)
            state[c] = old
        items.append((c, t, log_query(address=c[0], key=c[1], shard_id=shard_id, timestamp=1000 + t, tx_number_in_block=t % 7, **q)))
    unsorted = [q for _, _, q in items]
    order = sorted(range(n_items), key=lambda i: ((items[i][0][0] << 256) | items[i][0][1], items[i][1]))
    sorted_records = [(items[i][2], items[i][1]) for i in order]   # extended timestamp = position in the unsorted queue
    return unsorted, sorted_records
