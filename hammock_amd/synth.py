"""Synthetic peptide sets of BASELINE.md section 4 / SURVEY.md 8(d).

SplitMix64(seed); residue = (next() >> 33) % 20 over ARNDCQEGHILKMFPSTWYV;
with a length range, length = lo + (next() >> 33) % (hi - lo + 1) is drawn
before the residues of each peptide; peptides already seen are discarded and
drawing continues until n DISTINCT peptides exist.  Bit-identical to
hmo_synth() in oracle/hammock_oracle.c (tests/test_oracle.py checks that).
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_stream(seed: int, start: int, count: int) -> np.ndarray:
    """outputs number start .. start+count-1 (0-based) of SplitMix64(seed)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_peptides(seed: int, n: int, len_lo: int, len_hi: int | None = None):
    """-> (residues uint8 [sum len], offsets uint32 [n+1])"""
    if len_hi is None:
        len_hi = len_lo
    if len_lo < 1 or len_hi < len_lo:
        raise ValueError("need 1 <= len_lo <= len_hi")
    if n > sum(20 ** L for L in range(len_lo, min(len_hi, 8) + 1)) and len_hi <= 8:
        raise ValueError(f"only {sum(20 ** L for L in range(len_lo, len_hi + 1))} distinct peptides of length "
                         f"{len_lo}..{len_hi} exist, {n} requested")   # the draw-until-distinct loop would never end
    if len_lo == len_hi:
        L = len_lo
        rows = np.zeros((0, L), dtype=np.uint8)
        pos = 0
        seen = set()
        out = []
        while len(out) < n:
            want = n - len(out)
            z = splitmix64_stream(seed, pos, want * L)
            pos += want * L
            block = ((z >> np.uint64(33)) % np.uint64(20)).astype(np.uint8).reshape(want, L)
            if not seen and len(np.unique(block.view(np.dtype((np.void, L))))) == want:
                out = list(block)  # common case: no duplicates at all
                seen = None
                break
            for r in block:
                key = r.tobytes()
                if key not in seen:
                    seen.add(key)
                    out.append(r)
        res = np.ascontiguousarray(np.stack(out[:n]).reshape(-1)) if n else np.zeros(0, np.uint8)
        off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
        return res, off
    span = len_hi - len_lo + 1
    need = n * (len_hi + 1) + 64
    z = (splitmix64_stream(seed, 0, need) >> np.uint64(33))
    pos, got, seen, parts = 0, 0, set(), []
    while got < n:
        if pos + len_hi + 1 > len(z):
            z = np.concatenate([z, splitmix64_stream(seed, len(z), need) >> np.uint64(33)])
        L = len_lo + int(z[pos] % np.uint64(span))
        pep = (z[pos + 1:pos + 1 + L] % np.uint64(20)).astype(np.uint8)
        pos += 1 + L
        key = pep.tobytes()
        if key in seen:
            continue
        seen.add(key)
        parts.append(pep)
        got += 1
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(p) for p in parts])
    return np.ascontiguousarray(np.concatenate(parts)), off
