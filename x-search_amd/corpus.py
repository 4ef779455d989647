"""Deterministic synthetic text corpora (there is no network: no real corpus ships).

Shape follows the reference's integration corpus (test/src/xsearchTest.cpp:17,
60-67 + the decoded test/files/sample.meta): ASCII words, lines
of ~30 bytes, every line '\n'-terminated, the needle planted at ~4.6e-7 per byte,
chunks of `chunk_target` bytes extended to the next '\n'.

Everything is a pure function of (seed, index): the same bytes are produced on
any host, so a test can regenerate an input from its seed instead of storing it.
"""
from __future__ import annotations

import numpy as np

LEXICON = [
    b"the", b"of", b"and", b"to", b"a", b"in", b"that", b"it", b"was", b"I", b"for", b"on", b"you", b"he", b"be",
    b"with", b"as", b"by", b"at", b"have", b"are", b"this", b"not", b"but", b"had", b"his", b"they", b"from", b"she",
    b"which", b"or", b"we", b"an", b"were", b"been", b"their", b"has", b"would", b"what", b"will", b"there", b"if",
    b"can", b"all", b"her", b"said", b"who", b"one", b"so", b"up", b"them", b"some", b"could", b"him", b"into",
    b"time", b"She", b"locked", b"Sher", b"lock", b"Holmes", b"Watson", b"detective", b"street",
]


# The same lexicon without the four decoys that are pieces of the bench needle (`She`, `locked`, `Sher`, `lock`): text
# in which `She[r ]lock` and friends occur only where the needle was planted.  Separates what a kernel costs from
# what this corpus makes it do.
LEXICON_PLAIN = [w for w in LEXICON if w not in (b"She", b"locked", b"Sher", b"lock")]


# ... and without any word that begins with a capital S or H (`She`, `Sher`, `Holmes`): the trigger bytes of
# `Sherlock|Holmes` / `Sher.*mes` then occur only where the needle was planted
LEXICON_NOSH = [w for w in LEXICON if w[:1] not in (b"S", b"H")]


def _mix(seed: int, index: int) -> int:
    # splitmix64 of (seed, index) -> independent stream per block
    z = (seed + 0x9E3779B97F4A7C15 * (index + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def text_block(seed: int, index: int, nbytes: int, needle: bytes = b"Sherlock", needle_rate: float = 4.6e-7 * 6.0,
               words_per_line: float = 6.0, lexicon=None) -> np.ndarray:
    """One '\\n'-terminated block of EXACTLY `nbytes` bytes (uint8 array).

    needle_rate is the probability that a word is replaced by the needle
    (default gives ~4.6e-7 matches per byte with ~5-byte words + separator).
    """
    assert nbytes >= 2
    rng = np.random.Generator(np.random.PCG64(_mix(seed, index)))
    words = LEXICON if lexicon is None else lexicon
    lex = [np.frombuffer(w, dtype=np.uint8) for w in words] + [np.frombuffer(needle, dtype=np.uint8)]
    lens = np.array([len(w) for w in lex], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = np.concatenate(lex)
    mean = float(lens[:-1].mean()) + 1.0
    nwords = int(nbytes / mean * 1.15) + 16
    ids = rng.integers(0, len(words), size=nwords)
    plant = rng.random(nwords) < needle_rate
    ids[plant] = len(lex) - 1
    seps = np.where(rng.random(nwords) < 1.0 / words_per_line, 10, 32).astype(np.uint8)
    wl = lens[ids] + 1
    ends = np.cumsum(wl)
    k = int(np.searchsorted(ends, nbytes, side="left")) + 1  # words fully or partly inside nbytes
    ids, wl, ends, seps = ids[:k], wl[:k], ends[:k], seps[:k]
    total = int(ends[-1])
    assert total >= nbytes
    wstart = ends - wl
    # out[j] = flat[starts[id_k] + (j - wstart_k)] for j inside word k; separator at the word's last slot
    src = np.repeat(starts[ids] - wstart, wl) + np.arange(total, dtype=np.int64)
    sep_pos = ends - 1
    src[sep_pos] = 0
    out = flat[src]
    out[sep_pos] = seps
    out = out[:nbytes].copy()
    # exact size and '\n'-terminated: the cut may fall inside a word; that is fine for a byte scan
    out[nbytes - 1] = 10
    return out


def small_alphabet(seed: int, n: int, alphabet: bytes = b"ab\n", terminate: bool = False) -> np.ndarray:
    """Adversarial small-alphabet text (bordered patterns, dense overlaps, many newlines)."""
    rng = np.random.Generator(np.random.PCG64(_mix(seed, 0xABCDEF)))
    a = np.frombuffer(alphabet, dtype=np.uint8)
    out = a[rng.integers(0, len(a), size=n)] if n else np.zeros(0, dtype=np.uint8)
    out = out.copy()
    if terminate and n:
        out[-1] = 10
    return out


def chunk_table(lengths) -> tuple[np.ndarray, np.ndarray, int]:
    """Pack chunks of the given lengths at 256-byte aligned offsets.
    Returns (offsets, lengths, capacity) -- the shard layout xsg_shard_create expects."""
    lengths = np.asarray(lengths, dtype=np.uint64)
    if lengths.size == 0:
        return np.zeros(0, dtype=np.uint64), lengths, 0
    padded = (lengths + np.uint64(255)) // np.uint64(256) * np.uint64(256)
    # keep at least 256 readable bytes after every chunk
    padded = padded + np.uint64(256)
    offsets = np.concatenate([[np.uint64(0)], np.cumsum(padded)[:-1]]).astype(np.uint64)
    cap = int(padded.sum())
    return offsets, lengths, cap
