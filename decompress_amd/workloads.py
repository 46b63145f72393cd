"""Synthetic workloads of BASELINE.json's configs (seeded, no files needed).

C1  one 64 KiB stream of stored blocks (65535 + 1 bytes)
C2  N x 256 KiB zlib streams, dynamic Huffman (libz level 6)      [headline]
C3  N x 1 MiB uniform printable-ASCII buffers (deflate input)

Plaintext for C2 is Zipf-distributed word text (zlib ratio ~0.38, dynamic
Huffman blocks, full range of match distances) — the reference's corpus does
not travel to the GPU box, so the generator stands in for it.
"""
import zlib
from concurrent.futures import ProcessPoolExecutor

import numpy as np

_VOCAB = 16384
_ZIPF = 1.11  # zlib-6 ratio ~0.38, like English text (book2: 0.34-0.42)


_SEPS = np.frombuffer(b". , \n   ", dtype=np.uint8).reshape(4, 2)  # ". " ", " "\n" " "
_SEP_LEN = np.array([2, 2, 1, 1])
_vocab_cache = None


def _vocab():
    """(chars[V,12] uint8, length[V]) — fixed vocabulary of 2..10-letter words."""
    global _vocab_cache
    if _vocab_cache is None:
        rng = np.random.default_rng(1234)
        lens = rng.integers(2, 11, size=_VOCAB)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        p = 1.0 / np.arange(1, 27)
        p /= p.sum()
        chars = rng.choice(letters, size=(_VOCAB, 12), p=p)
        _vocab_cache = (chars, lens)
    return _vocab_cache


def text(seed, nbytes):
    """Deterministic pseudo-English of exactly nbytes bytes (vectorised)."""
    rng = np.random.default_rng(seed)
    chars, wlen = _vocab()
    nw = nbytes // 3 + 64
    ranks = (rng.zipf(_ZIPF, size=nw) - 1) % _VOCAB
    q = rng.integers(0, 23, size=nw)
    sep = np.where(q == 0, 0, np.where(q == 1, 1, np.where(q == 2, 2, 3)))
    wl = wlen[ranks]
    tl = wl + _SEP_LEN[sep]
    start = np.zeros(nw, dtype=np.int64)
    np.cumsum(tl[:-1], out=start[1:])
    total = int(start[-1] + tl[-1])
    out = np.empty(total, dtype=np.uint8)
    tok = np.repeat(np.arange(nw), tl)          # token index of every output byte
    k = np.arange(total) - start[tok]           # byte index inside the token
    inword = k < wl[tok]
    out[inword] = chars[ranks[tok[inword]], k[inword]]
    ns = ~inword
    out[ns] = _SEPS[sep[tok[ns]], k[ns] - wl[tok[ns]]]
    b = out.tobytes()
    while len(b) < nbytes:
        b += b
    return b[:nbytes]


def ascii_uniform(seed, nbytes):
    """C3 plaintext: bytes uniform over printable ASCII 0x20..0x7e."""
    rng = np.random.default_rng(seed)
    return rng.integers(0x20, 0x7f, size=nbytes, dtype=np.uint8).tobytes()


def stored_stream(payload):
    """C1: raw DEFLATE of stored blocks, 65535 bytes max each (BTYPE=00)."""
    out = bytearray()
    chunks = [payload[i:i + 65535] for i in range(0, len(payload), 65535)] or [b""]
    for k, c in enumerate(chunks):
        last = 1 if k == len(chunks) - 1 else 0
        out += bytes([last]) + len(c).to_bytes(2, "little") + (len(c) ^ 0xffff).to_bytes(2, "little") + c
    return bytes(out)


def _c2_one(args):
    seed, nbytes, level = args
    plain = text(seed, nbytes)
    return zlib.compress(plain, level)


def c2_streams(n, nbytes=256 * 1024, level=6, seed0=0xC2, workers=None, unique=None):
    """n zlib streams of `nbytes` plaintext each.  `unique` < n generates that
    many distinct streams and cycles them (seed = seed0 + i mod unique)."""
    unique = n if unique is None else min(unique, n)
    jobs = [(seed0 + i, nbytes, level) for i in range(unique)]
    if workers == 0 or unique < 8:
        uniq = [_c2_one(j) for j in jobs]
    else:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            uniq = list(ex.map(_c2_one, jobs, chunksize=max(1, unique // 64)))
    return [uniq[i % unique] for i in range(n)]


def pack(streams, align=16):
    """Pack byte strings into one uint8 array; returns (blob, offsets, lengths)."""
    lens = np.array([len(s) for s in streams], dtype=np.int64)
    offs = np.zeros(len(streams), dtype=np.int64)
    if len(streams) > 1:
        np.cumsum(((lens + align - 1) // align * align)[:-1], out=offs[1:])
    total = int(offs[-1] + lens[-1]) if len(streams) else 0
    blob = np.zeros(total + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        blob[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return blob, offs, lens
