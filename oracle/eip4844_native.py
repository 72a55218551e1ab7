"""oracle/eip4844_native.py — CPU ORACLE (test infrastructure): native restatement of eip_4844_entry_point
(/root/reference/src/eip_4844/mod.rs:107-260) and of the values the reference's own test computes out of circuit
(mod.rs:595-683: linear hash = keccak256(blob), z = last 16 bytes of keccak256(linear_hash ‖ versioned_hash) as a
big-endian integer, y = sum coeff_i z^i with the chunks as coefficients from the highest degree down,
output_hash = keccak256(versioned_hash ‖ z_be16 ‖ y_be32)).  Python big integers play the role of the native
BLS12-381 scalar field."""
from __future__ import annotations

from . import zko

BLS_FR = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
RATE, CHUNK = 136, 31


def shape(n_chunks):
    n_bytes = CHUNK * n_chunks
    n_blocks = n_bytes // RATE + 1
    cpi = -(-n_chunks // n_blocks)
    return n_bytes, n_blocks, cpi


def limbs16(x, n=16):
    return [(x >> (16 * i)) & 0xFFFF for i in range(n)]


def instance(blob: bytes, versioned_hash: bytes, n_chunks: int, linear_hash: bytes | None = None):
    """blob: 31*n_chunks bytes.  Returns outer words, loop rows (carried words included) and expected outputs."""
    n_bytes, n_blocks, cpi = shape(n_chunks)
    assert len(blob) == n_bytes and len(versioned_hash) == 32
    true_linear_hash = zko.keccak256(blob)
    linear_hash = true_linear_hash if linear_hash is None else linear_hash
    z = int.from_bytes(zko.keccak256(linear_hash + versioned_hash)[16:], "big")
    chunks = [int.from_bytes(blob[CHUNK * i:CHUNK * (i + 1)], "little") for i in range(n_chunks)]
    # reference test: coefficients in reversed order, evaluation by powers (mod.rs:640-648)
    y_ref, power = 0, 1
    for coeff in reversed(chunks):
        y_ref = (y_ref + coeff * power) % BLS_FR
        power = power * z % BLS_FR
    output_hash = zko.keccak256(versioned_hash + z.to_bytes(16, "big") + y_ref.to_bytes(32, "big"))

    padded = bytearray(blob) + bytes(RATE - n_bytes % RATE)
    padded[n_bytes] |= 0x01
    padded[-1] |= 0x80
    assert len(padded) == RATE * n_blocks
    state, opening, rows = [0] * 25, [0] * 16, []   # opening: 16 limbs, lazily added (limb-wise) before each reduction
    for t in range(n_blocks):
        st_bytes = [(lane >> (8 * k)) & 0xFF for lane in state for k in range(8)]
        block = list(blob[RATE * t:RATE * (t + 1)].ljust(RATE, b"\0"))       # raw blob bytes; the circuit pads the last block
        cw = list(blob[CHUNK * cpi * t:CHUNK * cpi * (t + 1)].ljust(CHUNK * cpi, b"\0"))
        rows.append(st_bytes + list(opening) + [t] + block + cw)
        for c in range(cpi):
            idx = cpi * t + c
            if idx >= n_chunks:
                continue
            opening = [a + b for a, b in zip(opening, limbs16(chunks[idx]))]   # lazy addition, no carry propagation
            if idx != n_chunks - 1:
                opening = limbs16(sum(l << (16 * i) for i, l in enumerate(opening)) * z % BLS_FR)
        for j in range(RATE):
            state[j // 8] ^= padded[RATE * t + j] << (8 * (j % 8))
        state = zko.keccak_f1600(state)
    assert sum(l << (16 * i) for i, l in enumerate(opening)) % BLS_FR == y_ref
    digest = b"".join(state[i].to_bytes(8, "little") for i in range(4))
    assert digest == true_linear_hash
    obs_out = list(true_linear_hash) + list(output_hash)
    z4 = [0] * 4
    compact = [1, 1] + zko.commit_encoding([]) + zko.commit_encoding(obs_out) + z4 + z4
    return dict(outer=list(versioned_hash) + list(linear_hash), rows=rows, z=z, y=y_ref, linear_hash=true_linear_hash,
                output_hash=output_hash, public_input=zko.commit_encoding(compact), satisfiable=linear_hash == true_linear_hash)
