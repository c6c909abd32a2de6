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


def make_keys_torch(n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3, seed: int = 11, bigram_share: float = 0.64,
                    device="cuda") -> Tuple[np.ndarray, np.ndarray]:
    """:func:`make_keys` drawn with torch on ``device`` (same law -- all unigrams, then DISTINCT bigrams / trigrams over
    Zipf(1.1) tokens in first-drawn order --, torch's generator instead of numpy's: not the same vocabulary).  The
    de-duplication passes over 1e7 keys take seconds on the GPU instead of a minute and a half on the host, which is what
    lets bench.py carry config C3 in its default run.  Returns host arrays like :func:`make_keys`."""
    import torch
    assert max_n >= 3 and n_rows >= vocab
    g = torch.Generator(device=device).manual_seed(int(seed))
    cdf = torch.from_numpy(zipf_cdf(vocab)).to(device)
    rest = n_rows - vocab
    have = torch.zeros(0, dtype=torch.int64, device=device)        # packed (token + 1) x 3, 20 bits each; 0 = absent
    while have.numel() < rest:
        m = max(4096, int((rest - have.numel()) * 1.6))
        t = torch.clamp(torch.searchsorted(cdf, torch.rand((m, 3), generator=g, device=device, dtype=torch.float64), right=True),
                        max=vocab - 1) + 1
        t[torch.rand(m, generator=g, device=device) < bigram_share, 2] = 0
        allk = torch.cat([have, t[:, 0] | (t[:, 1] << 20) | (t[:, 2] << 40)])
        # first occurrences in drawing order: stable sort by key, keep the first of every run, restore positions
        order = torch.argsort(allk, stable=True)
        sk = allk[order]
        first = torch.ones_like(sk, dtype=torch.bool)
        first[1:] = sk[1:] != sk[:-1]
        have = allk[torch.sort(order[first]).values]
    have = have[:rest]
    k3 = torch.stack([have & 0xFFFFF, (have >> 20) & 0xFFFFF, (have >> 40) & 0xFFFFF], dim=1)
    keys = np.zeros((n_rows, max_n), dtype=np.uint32)
    lens = np.ones(n_rows, dtype=np.uint8)
    keys[:vocab, 0] = np.arange(vocab, dtype=np.uint32)
    k3 = k3.cpu().numpy()
    lens[vocab:] = np.where(k3[:, 2] == 0, 2, 3).astype(np.uint8)
    keys[vocab:, :3] = np.where(k3 > 0, k3 - 1, 0)
    return keys, lens


STRUCTURED_MULTIPLIERS = (40503, 30011, 20011)


def check_structured_vocab(vocab: int) -> None:
    """The structured generator maps a digit x to (x * m + c) % vocab: a bijection of the digits -- distinct keys by
    construction -- only when every multiplier is coprime to the vocabulary (40503 = 3 * 23 * 587)."""
    import math
    if not 3 <= int(vocab) <= 262144:
        raise ValueError("structured vocabulary: vocab must be in [3, 262144] (the direct unigram table's size)")
    for m in STRUCTURED_MULTIPLIERS:
        if math.gcd(int(vocab), m) != 1:
            raise ValueError(f"structured vocabulary: vocab {vocab} shares a factor with the generator's multiplier {m}: "
                             "the keys would not be distinct (choose a vocab coprime to 40503 * 30011 * 20011)")


def structured_keys_for_ids(ids: np.ndarray, n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    """Rows ``ids`` of :func:`make_keys_structured` in closed form (any subset of a 1e9-row vocabulary without
    materialising it): ids 0..vocab-1 are the unigrams, then half bigrams and half trigrams whose tokens are the
    mixed-radix digits of a running counter (multiplied by odd constants mod vocab, so neighbouring ids do not share
    tokens)."""
    assert max_n >= 3 and n_rows >= vocab
    check_structured_vocab(vocab)
    ids = np.asarray(ids, dtype=np.int64)
    keys = np.zeros((ids.shape[0], max_n), dtype=np.uint32)
    lens = np.ones(ids.shape[0], dtype=np.uint8)
    V = np.uint64(vocab)
    rest = n_rows - vocab
    nb = rest // 2
    uni = ids < vocab
    bi = (~uni) & (ids < vocab + nb)
    tri = ids >= vocab + nb
    keys[uni, 0] = ids[uni].astype(np.uint32)
    j = (ids[bi] - vocab).astype(np.uint64)
    keys[bi, 0] = ((j % V) * np.uint64(40503) + np.uint64(17)) % V
    keys[bi, 1] = (((j // V) % V) * np.uint64(30011) + np.uint64(5)) % V
    lens[bi] = 2
    j = (ids[tri] - vocab - nb).astype(np.uint64)
    keys[tri, 0] = ((j % V) * np.uint64(40503) + np.uint64(29)) % V
    keys[tri, 1] = (((j // V) % V) * np.uint64(30011) + np.uint64(3)) % V
    keys[tri, 2] = (((j // (V * V)) % V) * np.uint64(20011) + np.uint64(11)) % V
    lens[tri] = 3
    return keys, lens


def make_keys_structured(n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3) -> Tuple[np.ndarray, np.ndarray]:
    """Distinct-by-construction vocabulary for very large tables (1e8 .. 1e9 rows), no de-duplication pass."""
    out_k = np.zeros((n_rows, max_n), dtype=np.uint32)
    out_l = np.ones(n_rows, dtype=np.uint8)
    step = 1 << 24
    for a in range(0, n_rows, step):
        b = min(a + step, n_rows)
        out_k[a:b], out_l[a:b] = structured_keys_for_ids(np.arange(a, b, dtype=np.int64), n_rows, vocab, max_n)
    return out_k, out_l


class StructuredVocab:
    """The structured vocabulary as an object a (sharded) cache can index WITHOUT host key arrays: the index of a
    1e9-key table is built from keys generated on the GPU in chunks (``torch`` arithmetic + ``index_build_device``),
    and the keys of any ids are available in closed form for laying out token streams.  Duck-types the part of
    ``NGramExtractor`` that ``ShardedEmbeddingCache`` / ``EmbeddingCache.from_synthetic`` use (``max_n``, ``len()``,
    ``build_index``)."""

    def __init__(self, n_rows: int, vocab: int = GPT2_VOCAB, max_n: int = 3) -> None:
        assert max_n == 3 and n_rows >= vocab
        check_structured_vocab(vocab)
        self.n_rows, self.vocab, self.max_n = int(n_rows), int(vocab), int(max_n)

    def __len__(self) -> int:
        return self.n_rows

    def keys_for(self, ids: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        return structured_keys_for_ids(ids, self.n_rows, self.vocab, self.max_n)

    def keys_for_torch(self, ids):
        """:meth:`keys_for` in torch arithmetic on the device of ``ids`` (int64): ``(keys [n, 3] int64, lens [n] uint8)``."""
        import torch
        V, nb = self.vocab, (self.n_rows - self.vocab) // 2
        keys = torch.zeros((ids.numel(), 3), dtype=torch.int64, device=ids.device)
        lens = torch.ones(ids.numel(), dtype=torch.uint8, device=ids.device)
        uni = ids < V
        keys[:, 0] = torch.where(uni, ids, keys[:, 0])
        bi = (~uni) & (ids < V + nb)
        j = ids - V
        k0 = ((j % V) * 40503 + 17) % V
        k1 = (((j // V) % V) * 30011 + 5) % V
        keys[:, 0] = torch.where(bi, k0, keys[:, 0])
        keys[:, 1] = torch.where(bi, k1, keys[:, 1])
        lens = torch.where(bi, torch.full_like(lens, 2), lens)
        tri = ids >= V + nb
        j = ids - V - nb
        k0 = ((j % V) * 40503 + 29) % V
        k1 = (((j // V) % V) * 30011 + 3) % V
        k2 = (((j // (V * V)) % V) * 20011 + 11) % V
        keys[:, 0] = torch.where(tri, k0, keys[:, 0])
        keys[:, 1] = torch.where(tri, k1, keys[:, 1])
        keys[:, 2] = torch.where(tri, k2, keys[:, 2])
        lens = torch.where(tri, torch.full_like(lens, 3), lens)
        return keys, lens

    def build_index(self, table, chunk: int = 1 << 25) -> None:
        import torch
        dev = table.device
        n = self.n_rows
        for a in range(0, n, chunk):
            b = min(a + chunk, n)
            keys, lens = self.keys_for_torch(torch.arange(a, b, dtype=torch.int64, device=dev))
            table.index_build_device(keys.to(torch.int32).contiguous(), lens.contiguous(), id0=a)
            torch.cuda.synchronize(dev)        # the chunk's temporaries are freed before the next one is built
            del keys, lens


def stream_uniform_ids(keys, lens, B: int, T: int, seed: int) -> np.ndarray:
    """S_uniform: f-grams with ids uniform in [0, N) laid end to end -- row reads that defeat
    L2 / Infinity-Cache reuse (the roofline run).  ``keys`` is the dense key array, or a :class:`StructuredVocab`
    (``lens`` ignored) whose keys are computed for the drawn ids only."""
    rng = np.random.default_rng(seed)
    need = B * T
    n_rows = len(keys) if isinstance(keys, StructuredVocab) else keys.shape[0]
    out = np.empty(0, dtype=np.int64)
    while out.size < need:
        ids = rng.integers(0, n_rows, size=max(1024, int((need - out.size) / 1.8) + 1024))
        if isinstance(keys, StructuredVocab):
            k, l = keys.keys_for(ids)
        else:
            k, l = keys[ids], lens[ids]
        mask = np.arange(k.shape[1])[None, :] < l[:, None]
        out = np.concatenate([out, k[mask].astype(np.int64)])
    return out[:need].reshape(B, T)


def stream_zipf_ids(keys, lens, B: int, T: int, seed: int, s: float = 1.1) -> np.ndarray:
    """S_zipf_ids: f-grams laid end to end like :func:`stream_uniform_ids`, their ids drawn from a Zipf-like law over
    ``[0, N)`` (bounded power law, exponent ``s``) instead of uniformly.  f-gram ids are FREQUENCY-ordered (the reference
    assigns them in ``Counter.most_common`` order, ``n_gram_extractor.py:91-99``), so this is what a table sorted by
    frequency sees from real text: the head of the table takes most references, the tail is long and its rows recur
    within a batch -- the stream on which a hot head in HBM and a de-duplicating prefetch pay."""
    rng = np.random.default_rng(seed)
    need = B * T
    n_rows = len(keys) if isinstance(keys, StructuredVocab) else keys.shape[0]
    out = np.empty(0, dtype=np.int64)
    e = 1.0 - s
    while out.size < need:
        u = rng.random(max(1024, int((need - out.size) / 1.5) + 1024))
        x = ((float(n_rows + 1) ** e - 1.0) * u + 1.0) ** (1.0 / e)          # inverse CDF of the bounded power law on [1, N + 1)
        ids = np.minimum(x.astype(np.int64) - 1, n_rows - 1)
        if isinstance(keys, StructuredVocab):
            k, l = keys.keys_for(ids)
        else:
            k, l = keys[ids], lens[ids]
        mask = np.arange(k.shape[1])[None, :] < l[:, None]
        out = np.concatenate([out, k[mask].astype(np.int64)])
    return out[:need].reshape(B, T)


SCRAMBLE_MULT = 61_803_399      # odd, not a multiple of 5: a bijection of [0, N) for the 10^k-row tables of BASELINE.json


def stream_zipf_ids_torch(vocab: "StructuredVocab", B: int, T: int, seed: int, s: float = 1.1, device="cuda",
                          scramble: bool = False, shift: int = 0):
    """:func:`stream_zipf_ids` for a :class:`StructuredVocab`, drawn on the GPU (same law, torch's generator instead of
    numpy's: not the same batch): a serving loop's worth of DIFFERENT batches -- what a cache of cold rows must be measured
    on -- costs milliseconds each instead of a quarter of a second of host time.  int32 ``[B, T]`` on ``device``.

    ``scramble`` / ``shift``: the popularity RANK r is served by row ``(r * SCRAMBLE_MULT + shift) % N`` instead of row r --
    the same law over a table whose order is NOT the traffic's frequency order (a table built on one corpus and served on
    another; ``shift`` growing from batch to batch: a hot set that moves).  This is the stream on which a static hot head
    cannot work and a cache of cold rows has to."""
    import math
    import torch
    if scramble and math.gcd(SCRAMBLE_MULT, len(vocab)) != 1:
        raise ValueError("scramble needs a row count coprime to SCRAMBLE_MULT")
    g = torch.Generator(device=device).manual_seed(int(seed))
    need, n_rows, e = B * T, len(vocab), 1.0 - s
    parts, have = [], 0
    while have < need:
        u = torch.rand(max(1024, int((need - have) / 1.5) + 1024), generator=g, device=device, dtype=torch.float64)
        x = ((float(n_rows + 1) ** e - 1.0) * u + 1.0) ** (1.0 / e)          # inverse CDF of the bounded power law on [1, N + 1)
        ids = torch.clamp(x.to(torch.int64) - 1, max=n_rows - 1)
        if scramble:
            ids = (ids * SCRAMBLE_MULT + int(shift)) % n_rows
        elif shift:
            ids = (ids + int(shift)) % n_rows
        k, l = vocab.keys_for_torch(ids)
        t = k[torch.arange(3, device=k.device)[None, :] < l[:, None]]
        parts.append(t)
        have += t.numel()
    return torch.cat(parts)[:need].reshape(B, T).to(torch.int32)


def stream_zipf(vocab: int, B: int, T: int, seed: int) -> np.ndarray:
    """S_zipf: iid Zipf(1.1) tokens (the realistic stream of the reference's C1 probe)."""
    rng = np.random.default_rng(seed)
    return zipf_tokens(rng, zipf_cdf(vocab), (B, T))
