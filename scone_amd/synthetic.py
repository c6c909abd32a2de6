"""Seeded synthetic vocabularies and token streams used by bench.py and
the full-size tests.  Host-side numpy only; the table rows themselves are generated on the GPU
(``scone_table_fill_synthetic``)."""

from typing import Tuple

import numpy as np

GPT2_VOCAB = 50257


def zipf_cdf(vocab: int, s: float = 1.1) -> np.ndarray:
    p = np.arange(1, vocab + 1, dtype=np.float64) ** (-s)
    return np.cumsum(p / p.sum())


def zipf_tokens(rng: np.random.Generator, cdf: np.ndarray, size) -> np.ndarray:
    return np.minimum(np.searchsorted(cdf, rng.random(size), side="right"), cdf.shape[0] - 1).astype(np.int64)


def make_keys(n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3, seed: int = 11,
              bigram_share: float = 0.64) -> Tuple[np.ndarray, np.ndarray]:
    """ids 0..vocab-1 are all unigrams; the remaining ids are DISTINCT bigrams / trigrams (about
    64 % / 36 %, the mix measured on the reference's own fit) over Zipf(1.1) tokens, in
    first-drawn order."""
    assert max_n >= 3 and n_rows >= vocab
    rng = np.random.default_rng(seed)
    cdf = zipf_cdf(vocab)
    keys = np.zeros((n_rows, max_n), dtype=np.uint32)
    lens = np.ones(n_rows, dtype=np.uint8)
    keys[:vocab, 0] = np.arange(vocab, dtype=np.uint32)
    rest = n_rows - vocab
    have = np.zeros((0, 3), dtype=np.uint32)      # token+1 per slot, 0 = absent -> distinct rows == distinct keys
    while have.shape[0] < rest:
        m = max(4096, int((rest - have.shape[0]) * 1.6))
        t = zipf_tokens(rng, cdf, (m, 3)).astype(np.uint32) + 1
        t[rng.random(m) < bigram_share, 2] = 0
        allk = np.concatenate([have, t])
        packed = allk[:, 0].astype(np.uint64) | (allk[:, 1].astype(np.uint64) << np.uint64(20)) | \
            (allk[:, 2].astype(np.uint64) << np.uint64(40))
        _, first = np.unique(packed, return_index=True)
        have = allk[np.sort(first)]               # keep first occurrences, in drawing order
    have = have[:rest]
    lens[vocab:] = np.where(have[:, 2] == 0, 2, 3).astype(np.uint8)
    keys[vocab:, :3] = np.where(have > 0, have - 1, 0)
    return keys, lens


def make_keys_structured(n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    """Distinct-by-construction vocabulary for very large tables (1e8 .. 1e9 rows), no de-duplication
    pass: ids 0..vocab-1 are the unigrams, then half bigrams and half trigrams whose tokens are the
    mixed-radix digits of a running counter (multiplied by odd constants mod vocab, so neighbouring
    ids do not share tokens)."""
    assert max_n >= 3 and n_rows >= vocab
    keys = np.zeros((n_rows, max_n), dtype=np.uint32)
    lens = np.ones(n_rows, dtype=np.uint8)
    keys[:vocab, 0] = np.arange(vocab, dtype=np.uint32)
    rest = n_rows - vocab
    nb = rest // 2
    j = np.arange(nb, dtype=np.uint64)
    V = np.uint64(vocab)
    keys[vocab:vocab + nb, 0] = ((j % V) * np.uint64(40503) + np.uint64(17)) % V
    keys[vocab:vocab + nb, 1] = (((j // V) % V) * np.uint64(30011) + np.uint64(5)) % V
    lens[vocab:vocab + nb] = 2
    j = np.arange(rest - nb, dtype=np.uint64)
    keys[vocab + nb:, 0] = ((j % V) * np.uint64(40503) + np.uint64(29)) % V
    keys[vocab + nb:, 1] = (((j // V) % V) * np.uint64(30011) + np.uint64(3)) % V
    keys[vocab + nb:, 2] = (((j // (V * V)) % V) * np.uint64(20011) + np.uint64(11)) % V
    lens[vocab + nb:] = 3
    return keys, lens


def stream_uniform_ids(keys: np.ndarray, lens: np.ndarray, B: int, T: int, seed: int) -> np.ndarray:
    """S_uniform: f-grams with ids uniform in [0, N) laid end to end -- row reads that defeat
    L2 / Infinity-Cache reuse (the roofline run)."""
    rng = np.random.default_rng(seed)
    need = B * T
    out = np.empty(0, dtype=np.int64)
    while out.size < need:
        ids = rng.integers(0, keys.shape[0], size=max(1024, int((need - out.size) / 1.8) + 1024))
        k = keys[ids]
        mask = np.arange(keys.shape[1])[None, :] < lens[ids][:, None]
        out = np.concatenate([out, k[mask].astype(np.int64)])
    return out[:need].reshape(B, T)


def stream_zipf(vocab: int, B: int, T: int, seed: int) -> np.ndarray:
    """S_zipf: iid Zipf(1.1) tokens (the realistic stream of the reference's C1 probe)."""
    rng = np.random.default_rng(seed)
    return zipf_tokens(rng, zipf_cdf(vocab), (B, T))
