"""Workloads of BASELINE.json's configs as SURVEY.md 8(d) defines them (seeded; the only file is the data fixture
tests/golden/corpus.tar.xz = the reference's test/corpus, packed by tests/golden/gen_corpus.py).

C1  one 64 KiB stream of stored blocks (65535 + 1 bytes)
C2  N x 256 KiB zlib streams, dynamic Huffman (libz level 6)      [headline]
    plaintext i: even i = the 256 KiB of the concatenated corpus that start at offset i * 4099 (mod its length),
    odd i = order-2 Markov text over a 64-symbol ASCII alphabet, seed 0xC2 + i (markov_text: zlib-6 ratio 0.40);
    rounds 1-4 used seeded Zipf word text there (text(): still BENCH's `r01_workload` leg)
C3  N x 1 MiB uniform printable-ASCII buffers (deflate input)
C4  the 15 corpus files, file[i mod 15], as gzip members
"""
import io
import lzma
import os
import tarfile
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


_corpus_cache = None


def corpus():
    """the reference's test/corpus as {name: bytes}, in name order (15 files, 3 263 944 bytes)"""
    global _corpus_cache
    if _corpus_cache is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "corpus.tar.xz")
        with open(path, "rb") as f:
            tar = tarfile.open(fileobj=io.BytesIO(lzma.decompress(f.read())))
        _corpus_cache = {m.name: tar.extractfile(m).read() for m in sorted(tar.getmembers(), key=lambda m: m.name)}
    return _corpus_cache


_corpus_cat = None


def corpus_slice(i, nbytes):
    """C2's even streams: nbytes of the concatenated corpus from offset i * 4099 (mod its length), wrapping"""
    global _corpus_cat
    if _corpus_cat is None:
        _corpus_cat = b"".join(corpus().values())
    total = len(_corpus_cat)
    off = (i * 4099) % total
    out = _corpus_cat[off:off + nbytes]
    while len(out) < nbytes:
        out += _corpus_cat[:nbytes - len(out)]
    return out


def c2_plain(i, nbytes, seed0=0xC2):
    """plaintext of C2's stream i"""
    return corpus_slice(i, nbytes) if i % 2 == 0 else markov_text(seed0 + i, nbytes)


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
    i, nbytes, level, seed0 = args
    return zlib.compress(c2_plain(i, nbytes, seed0), level)


def _c2_text_one(args):
    """round 1's C2 stand-in: every stream seeded word text"""
    seed, nbytes, level = args
    return zlib.compress(text(seed, nbytes), level)


def c2_streams(n, nbytes=256 * 1024, level=6, seed0=0xC2, workers=None, unique=None, first=0):
    """n zlib streams of `nbytes` plaintext each: streams first .. first + n - 1 of C2.  `unique` < n generates
    that many distinct streams and cycles them."""
    unique = n if unique is None else min(unique, n)
    jobs = [(first + i, nbytes, level, seed0) for i in range(unique)]
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


# ---- C2's odd streams as SURVEY §8(d) defines them: order-2 Markov ASCII text --------------------------------------
_MARKOV_ALPHABET = (b" \netaoinshrdlcumwfgypbvkjxqz" b"ETAOINSHRDLCUMWFGYPBVKJXQZ" b".,;:'-!?()")
assert len(_MARKOV_ALPHABET) == 64 and len(set(_MARKOV_ALPHABET)) == 64
_MARKOV_THETA = 0.288  # geometric decay of a state's next-symbol probabilities: tuned so that zlib level 6 gives ~0.40
_markov_cache = None


def _markov_table():
    """4096 states (the last two symbols) x 256 slots -> the next STATE for a uniform random byte, as one flat list.
    Every state has its own order of the 64 symbols (fixed seed), the k-th of them with probability ~ theta^k, quantised
    to 1/256."""
    global _markov_cache
    if _markov_cache is None:
        rng = np.random.default_rng(0xC2C2)
        p = _MARKOV_THETA ** np.arange(64)
        p /= p.sum()
        edges = np.minimum(255, np.floor(np.cumsum(p) * 256).astype(np.int64))  # slot r belongs to the first k with r <= edges[k]
        rank_of_slot = np.searchsorted(edges, np.arange(256), side="left")
        perm = np.argsort(rng.random((4096, 64)), axis=1)
        nxt_sym = perm[:, rank_of_slot]                                      # [4096, 256]
        nxt_state = ((np.arange(4096)[:, None] & 63) << 6) | nxt_sym
        _markov_cache = nxt_state.reshape(-1).tolist()
    return _markov_cache


def markov_text(seed, nbytes):
    """nbytes of order-2 Markov text over a 64-symbol ASCII alphabet (one chain per stream, seeded)."""
    flat = _markov_table()
    r = np.random.default_rng(seed).integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()
    st = seed & 4095
    states = [st := flat[(st << 8) | x] for x in r]
    sym = (np.array(states, dtype=np.uint16) & 63).astype(np.uint8)
    return np.frombuffer(_MARKOV_ALPHABET, dtype=np.uint8)[sym].tobytes()
