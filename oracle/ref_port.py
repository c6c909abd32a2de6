"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's f-gram lookup path.

Nothing under ``scone_amd/`` may import this module.  The only allowed users are
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; there it is the *checker* (or the timed CPU baseline), never the
product.

Each function restates one piece of llmsresearch/scone (paths relative to the
reference checkout) and cites the lines it follows:

* ``fit``                     scone/tokenization/n_gram_extractor.py:72-104
* ``get_token_f_grams``       scone/tokenization/n_gram_extractor.py:106-126
* ``RefCache``                scone/inference/embedding_cache.py:29-181
* ``aggregate``               scone/inference/engine.py:234-266
* ``combine``                 scone/models/language_model.py:234-254

Pinning: ``tests/golden/make_golden.py`` imports the real reference in the build
container and stores its outputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those
fixtures (ids bit-exact, fp32 vectors bit-exact).

The second half of the file holds numpy-vectorised equivalents (``match_hits``,
``hits_to_csr``, ``embed_numpy``) used for larger parity cases, and the numpy
statement of *this repo's* table formats (INT8 / INT4 quantisers).  The
reference has no quantised table (its cache is always fp32,
embedding_cache.py:86,134,139), so the quantisation format is "parity unpinned"
by the reference: only lookup / reduce semantics are pinned, and the oracle is
always run on the *dequantised* table.
"""

from collections import Counter
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------
# Line-for-line restatement (pure Python + torch, as the reference runs it)
# --------------------------------------------------------------------------

def fit(tokenized_texts: Sequence[Sequence[int]], max_n: int, min_freq: int,
        max_f_grams: int) -> List[Tuple[int, ...]]:
    """n_gram_extractor.py:72-104.  Returns the f-grams in id order.

    Counter.most_common: count-descending, ties in first-insertion order
    (n_gram_extractor.py:91-99).  Per text, n-grams are inserted n = 1..max_n,
    start ascending (extract_all_n_grams, :59-70).
    """
    counter: Counter = Counter()
    for token_ids in tokenized_texts:
        all_n_grams = []
        for n in range(1, min(max_n + 1, len(token_ids) + 1)):
            all_n_grams.extend(
                tuple(token_ids[i:i + n]) for i in range(len(token_ids) - n + 1))
        counter.update(all_n_grams)
    return [g for g, c in counter.most_common(max_f_grams) if c >= min_freq]


def get_token_f_grams(f_grams: Set[Tuple[int, ...]], max_n: int,
                      token_ids: Sequence[int]) -> Dict[int, List[Tuple[int, ...]]]:
    """n_gram_extractor.py:106-126: every position gets every f-gram covering it.

    Order per position: n ascending, then window start ascending; duplicates kept.
    """
    token_f_grams: Dict[int, List[Tuple[int, ...]]] = {i: [] for i in range(len(token_ids))}
    for n in range(1, min(max_n + 1, len(token_ids) + 1)):
        for i in range(len(token_ids) - n + 1):
            n_gram = tuple(token_ids[i:i + n])
            if n_gram in f_grams:
                for j in range(i, i + n):
                    token_f_grams[j].append(n_gram)
    return token_f_grams


class RefCache:
    """embedding_cache.py:29-181 (in-memory dict variant and [N,d] fp32 array variant)."""

    def __init__(self, f_gram_to_id: Dict[Tuple[int, ...], int], max_n: int,
                 embedding_dim: int, use_memory_map: bool = False) -> None:
        self.f_gram_to_id = f_gram_to_id
        self.f_grams = set(f_gram_to_id.keys())
        self.max_n = max_n
        self.embedding_dim = embedding_dim
        self.use_memory_map = use_memory_map
        self.embeddings: Dict[int, np.ndarray] = {}
        self.memory_mapped_embeddings: Optional[np.ndarray] = None

    def cache_embeddings(self, f_gram_ids: Sequence[int], embeddings: torch.Tensor) -> None:
        """embedding_cache.py:56-111."""
        if self.use_memory_map:
            if self.memory_mapped_embeddings is None:
                # :76-91 -- [len(f_grams), d] fp32, zero-filled
                self.memory_mapped_embeddings = np.zeros(
                    (len(self.f_grams), self.embedding_dim), dtype=np.float32)
            for f_gram_id, embedding in zip(f_gram_ids, embeddings):
                self.memory_mapped_embeddings[f_gram_id] = embedding.cpu().numpy()
        else:
            for f_gram_id, embedding in zip(f_gram_ids, embeddings):
                self.embeddings[f_gram_id] = embedding.cpu().numpy()

    def get_embeddings(self, f_gram_ids: Sequence[int], device: Optional[torch.device] = None) -> torch.Tensor:
        """embedding_cache.py:113-147."""
        if self.use_memory_map:
            if self.memory_mapped_embeddings is None:
                raise ValueError("Memory-mapped embeddings not initialized")
            embeddings = torch.tensor(self.memory_mapped_embeddings[list(f_gram_ids)],
                                      dtype=torch.float32)
        else:
            embeddings = torch.stack([
                torch.tensor(self.embeddings[f_gram_id], dtype=torch.float32)
                for f_gram_id in f_gram_ids
            ])
        if device is not None:                                       # :144-145
            embeddings = embeddings.to(device)
        return embeddings

    def get_token_embeddings(self, token_ids: Sequence[int]) -> Dict[int, torch.Tensor]:
        """embedding_cache.py:149-181: positions with no f-gram are omitted (:169-170)."""
        token_f_grams = get_token_f_grams(self.f_grams, self.max_n, token_ids)
        token_embeddings = {}
        for pos, f_grams in token_f_grams.items():
            if not f_grams:
                continue
            f_gram_ids = [self.f_gram_to_id[g] for g in f_grams]
            token_embeddings[pos] = self.get_embeddings(f_gram_ids)
        return token_embeddings


def aggregate(cache: RefCache, token_ids: Sequence[int], hidden_size: int,
              half: bool = False, device: torch.device = torch.device("cpu")) -> torch.Tensor:
    """engine.py:234-266: mean over the K_t rows, zeros where K_t = 0, optional .half().
    ``device`` plays engine.device (the CPU here), so the per-position ``.to(device)`` of
    embedding_cache.py:144-145 is executed as in the reference."""
    token_f_grams = get_token_f_grams(cache.f_grams, cache.max_n, token_ids)
    token_embeddings = {}
    for pos, f_grams in token_f_grams.items():
        if not f_grams:
            continue
        f_gram_ids = [cache.f_gram_to_id[g] for g in f_grams]
        embeddings = cache.get_embeddings(f_gram_ids, device)      # engine.py:247
        token_embeddings[pos] = embeddings.mean(dim=0)            # engine.py:250
    f_gram_embeddings = torch.zeros((1, len(token_ids), hidden_size), device=device)  # :253-256
    for pos, embedding in token_embeddings.items():
        f_gram_embeddings[0, pos] = embedding                        # :258-259
    if half:
        f_gram_embeddings = f_gram_embeddings.half()                # :265-266
    return f_gram_embeddings


def combine(input_ids: torch.Tensor, f_gram_embeddings: Optional[torch.Tensor],
            wte: torch.Tensor, wpe: torch.Tensor,
            proj_weight: Optional[torch.Tensor] = None,
            position_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """language_model.py:234-254: proj (bias-free Linear) -> + wte(ids) -> + wpe(pos)."""
    if f_gram_embeddings is not None and proj_weight is not None:
        f_gram_embeddings = torch.nn.functional.linear(f_gram_embeddings, proj_weight)  # :235-236
    base = torch.nn.functional.embedding(input_ids, wte)                                # :239
    combined = base + f_gram_embeddings if f_gram_embeddings is not None else base       # :242-245
    if position_ids is None:                                                             # :248-251
        position_ids = torch.arange(0, input_ids.size(1), dtype=torch.long).unsqueeze(0)
    return combined + torch.nn.functional.embedding(position_ids, wpe)                   # :253-254


def match_csr_python(f_gram_to_id: Dict[Tuple[int, ...], int], max_n: int,
                     token_ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Per-position id lists (get_token_f_grams + the id map of embedding_cache.py:173)
    flattened as CSR: offsets[T+1], ids[sum K]."""
    tfg = get_token_f_grams(set(f_gram_to_id.keys()), max_n, token_ids)
    offsets = np.zeros(len(token_ids) + 1, dtype=np.int64)
    ids: List[int] = []
    for pos in range(len(token_ids)):
        ids.extend(f_gram_to_id[g] for g in tfg[pos])
        offsets[pos + 1] = len(ids)
    return offsets, np.asarray(ids, dtype=np.int64)


# --------------------------------------------------------------------------
# The PAPER's lookup (Algorithm 2 in assets/algorithm.png, shown at README.md:28-32).  The
# reference CODE does not implement it (it uses the covering / mean / add scheme above), so this
# restatement is pinned to the published algorithm only: "parity unpinned" by reference code.
#   for i = 1..m: j <- smallest j' < i s.t. (sigma_j', ..., sigma_i) in V_f-gram, else i
#                 e_i <- T(sigma_i) if j == i else F(sigma_j, ..., sigma_i)
# V_f-gram holds n-grams of length 2..n only, so unigram keys never match.
# --------------------------------------------------------------------------

def paper_lookup(f_gram_to_id: Dict[Tuple[int, ...], int], max_n: int, token_ids: Sequence[int]) -> List[int]:
    """Per position: id of the longest f-gram (length 2..max_n) ending there, or -1."""
    out = []
    for i in range(len(token_ids)):
        hit = -1
        for j in range(max(0, i - max_n + 1), i):              # smallest j' first = longest f-gram
            g = tuple(int(x) for x in token_ids[j:i + 1])
            if g in f_gram_to_id:
                hit = f_gram_to_id[g]
                break
        out.append(hit)
    return out


def paper_embed(f_gram_to_id: Dict[Tuple[int, ...], int], max_n: int, tok: np.ndarray, table_f32: np.ndarray,
                wte: Optional[np.ndarray] = None, wpe: Optional[np.ndarray] = None) -> np.ndarray:
    """e_i = F(f-gram) if one ends at i else T(sigma_i); then + position embedding (the model adds it
    to whatever the embedding layer returns, language_model.py:253-254).  fp32, [B, T, d]."""
    B, T = tok.shape
    d = table_f32.shape[1]
    out = np.zeros((B, T, d), dtype=np.float32)
    for b in range(B):
        ids = paper_lookup(f_gram_to_id, max_n, tok[b].tolist())
        for i, fid in enumerate(ids):
            if fid >= 0:
                e = table_f32[fid].astype(np.float32)
            elif wte is not None:
                e = wte[tok[b, i]].astype(np.float32)
            else:
                e = np.zeros(d, dtype=np.float32)
            # same fp32 association as the kernel: (base + f_gram) + pos with the unused term = 0
            out[b, i] = (np.float32(0) + e) + (wpe[i].astype(np.float32) if wpe is not None else np.float32(0))
    return out


# --------------------------------------------------------------------------
# numpy-vectorised equivalents (same results; for larger parity cases)
# --------------------------------------------------------------------------

def keys_from_dict(f_gram_to_id: Dict[Tuple[int, ...], int], max_n: int):
    """Dense (keys[N,max_n] uint32, lens[N] uint8) arrays, row number == id."""
    n = len(f_gram_to_id)
    keys = np.zeros((n, max_n), dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint8)
    for g, i in f_gram_to_id.items():
        keys[i, :len(g)] = g
        lens[i] = len(g)
    return keys, lens


def _key_dict(keys: np.ndarray, lens: np.ndarray, id0: int = 0):
    d: Dict[Tuple[int, ...], int] = {}
    for i in range(keys.shape[0]):
        g = tuple(int(x) for x in keys[i, :lens[i]])
        d.setdefault(g, id0 + i)        # first id wins on duplicate keys (repo convention)
    return d


def match_hits(keys: np.ndarray, lens: np.ndarray, tok: np.ndarray, max_n: int) -> np.ndarray:
    """hits[n-1, b, i] = id of the f-gram tok[b, i:i+n] or -1 (window must fit in T).

    Same membership test as n_gram_extractor.py:119-121, one entry per (n, start).
    """
    B, T = tok.shape
    kd = None                        # python dict of all keys: only built if the packed-u64 path cannot be used
    hits = np.full((max_n, B, T), -1, dtype=np.int32)
    # per-length sorted key arrays -> searchsorted on packed python-int-free keys
    for n in range(1, max_n + 1):
        if T < n:
            continue
        sel = np.nonzero(lens == n)[0]
        if sel.size == 0:
            continue
        # pack n tokens (< 2**21 each for n<=3, < 2**16 for n=4 -> fits u64) else fall back
        kn = keys[sel, :n].astype(np.uint64)
        bits = 64 // n
        if int(kn.max(initial=0)) >= (1 << bits) or int(tok.max(initial=0)) >= (1 << bits) or tok.min(initial=0) < 0:
            if kd is None:
                kd = _key_dict(keys, lens)
            for b in range(B):
                for i in range(T - n + 1):
                    hits[n - 1, b, i] = kd.get(tuple(int(x) for x in tok[b, i:i + n]), -1)
            continue
        packed = np.zeros(sel.size, dtype=np.uint64)
        for k in range(n):
            packed |= kn[:, k] << np.uint64(bits * k)
        order = np.argsort(packed, kind="stable")
        sp = packed[order]
        sid = sel[order]
        # first id wins on duplicate keys: stable sort keeps ascending id among equals
        win = np.zeros((B, T - n + 1), dtype=np.uint64)
        for k in range(n):
            win |= tok[:, k:T - n + 1 + k].astype(np.uint64) << np.uint64(bits * k)
        pos = np.searchsorted(sp, win, side="left")
        posc = np.minimum(pos, sp.size - 1)
        found = sp[posc] == win
        hits[n - 1, :, :T - n + 1] = np.where(found, sid[posc], -1).astype(np.int32)
    return hits


def match_hits_structured(n_rows: int, tok: np.ndarray, max_n: int = 3, vocab: int = 50257) -> np.ndarray:
    """:func:`match_hits` for the STRUCTURED synthetic vocabulary (scone_amd/synthetic.py structured_keys_for_ids) without
    its key arrays: the generator is a bijection from ids to keys, inverted here in closed form, so membership of a window
    and its id (n_gram_extractor.py:119-121) cost a few modular multiplications -- what lets bench.py spot-check a 1e8-row
    table against this oracle.  ids 0..vocab-1: the unigrams; then nb = (n_rows - vocab) // 2 bigrams
    ((a * 40503 + 17) % V, (b * 30011 + 5) % V) with j = a + b V < nb; then the trigrams
    ((a * 40503 + 29) % V, (b * 30011 + 3) % V, (c * 20011 + 11) % V) with j = a + b V + c V^2 < n_rows - vocab - nb.
    Pinned by tests/test_host_logic.py::test_structured_vocabulary_oracle_equals_the_materialised_one against match_hits on
    the materialised keys (match_hits itself is pinned to the reference's outputs by tests/test_oracle_golden.py)."""
    assert max_n == 3
    V = int(vocab)
    B, T = tok.shape
    hits = np.full((max_n, B, T), -1, dtype=np.int32)
    t = tok.astype(np.int64)
    ok = (t >= 0) & (t < V)
    hits[0] = np.where(ok, t, -1).astype(np.int32)                  # every token of the vocabulary is a unigram f-gram
    rest = n_rows - V
    nb = rest // 2
    ntri = rest - nb
    i1, i2, i3 = pow(40503, -1, V), pow(30011, -1, V), pow(20011, -1, V)
    if T >= 2:
        a = ((t[:, :-1] - 17) % V) * i1 % V
        b = ((t[:, 1:] - 5) % V) * i2 % V
        j = a + b * V
        good = ok[:, :-1] & ok[:, 1:] & (j < nb)
        hits[1, :, :T - 1] = np.where(good, V + j, -1).astype(np.int32)
    if T >= 3:
        a = ((t[:, :-2] - 29) % V) * i1 % V
        b = ((t[:, 1:-1] - 3) % V) * i2 % V
        c = ((t[:, 2:] - 11) % V) * i3 % V
        j = a + b * V + c * V * V
        good = ok[:, :-2] & ok[:, 1:-1] & ok[:, 2:] & (j < ntri)
        hits[2, :, :T - 2] = np.where(good, V + nb + j, -1).astype(np.int32)
    return hits


def hits_to_csr(hits: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Expand (n,start) hits into the per-position id lists of
    n_gram_extractor.py:117-124 (n ascending, start ascending, duplicates kept);
    sequences are independent (no cross-sequence windows)."""
    max_n, B, T = hits.shape
    cols = []
    for n in range(1, max_n + 1):
        for s in range(n - 1, -1, -1):           # start i = j - s, ascending i
            c = np.full((B, T), -1, dtype=np.int64)
            if T - s > 0:
                c[:, s:] = hits[n - 1, :, :T - s]
            cols.append(c)
    cand = np.stack(cols, axis=-1).reshape(B * T, len(cols))   # [B*T, max_n(max_n+1)/2]
    valid = cand >= 0
    counts = valid.sum(axis=1)
    offsets = np.zeros(B * T + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets, cand[valid]


def embed_numpy(table_f32: np.ndarray, offsets: np.ndarray, ids: np.ndarray,
                reduce: str = "mean") -> np.ndarray:
    """Sequential fp32 sum in list order, then / K (engine.py:250); zeros where K = 0."""
    ntok = offsets.shape[0] - 1
    d = table_f32.shape[1]
    out = np.zeros((ntok, d), dtype=np.float32)
    counts = np.diff(offsets)
    kmax = int(counts.max(initial=0))
    for k in range(kmax):
        m = counts > k
        out[m] = out[m] + table_f32[ids[offsets[:-1][m] + k]]
    if reduce == "mean":
        nz = counts > 0
        out[nz] = out[nz] / counts[nz, None].astype(np.float32)
    return out


# --------------------------------------------------------------------------
# This repo's table formats (numpy statement; "parity unpinned" by the reference)
# --------------------------------------------------------------------------

def quantize_i8(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-row symmetric INT8: scale = fp16(absmax / 127); q = clamp(rint(x / scale), +-127)."""
    rows = np.asarray(rows, dtype=np.float32)
    absmax = np.abs(rows).max(axis=1) if rows.shape[1] else np.zeros(rows.shape[0], np.float32)
    scale = (absmax.astype(np.float32) / np.float32(127.0)).astype(np.float16)
    sf = scale.astype(np.float32)
    safe = np.where(sf > 0, sf, np.float32(1.0))
    q = np.rint(rows / safe[:, None])
    q = np.where(sf[:, None] > 0, np.clip(q, -127, 127), 0).astype(np.int8)
    return q, scale


def dequantize_i8(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    return q.astype(np.float32) * scale.astype(np.float32)[:, None]


I4_GROUP = 128


def quantize_i4(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Group-wise (128 along d) symmetric INT4: scale = fp16(absmax / 7);
    q = clamp(rint(x / scale), +-7), stored offset-binary (q + 8), two per byte,
    element 2k in the low nibble."""
    rows = np.asarray(rows, dtype=np.float32)
    n, d = rows.shape
    assert d % I4_GROUP == 0
    g = rows.reshape(n, d // I4_GROUP, I4_GROUP)
    absmax = np.abs(g).max(axis=2)
    scale = (absmax / np.float32(7.0)).astype(np.float16)
    sf = scale.astype(np.float32)
    safe = np.where(sf > 0, sf, np.float32(1.0))
    q = np.rint(g / safe[:, :, None])
    q = np.where(sf[:, :, None] > 0, np.clip(q, -7, 7), 0).astype(np.int32).reshape(n, d)
    u = (q + 8).astype(np.uint8)
    packed = (u[:, 0::2] | (u[:, 1::2] << 4)).astype(np.uint8)
    return packed, scale


def dequantize_i4(packed: np.ndarray, scale: np.ndarray) -> np.ndarray:
    n = packed.shape[0]
    d = packed.shape[1] * 2
    q = np.empty((n, d), dtype=np.float32)
    q[:, 0::2] = (packed & 0xF).astype(np.float32) - 8.0
    q[:, 1::2] = (packed >> 4).astype(np.float32) - 8.0
    sf = np.repeat(scale.astype(np.float32), I4_GROUP, axis=1)
    return q * sf


# --------------------------------------------------------------------------
# Synthetic table / key generators shared by tests and bench (counter-based, so
# any row can be recomputed on the host without materialising the table)
# --------------------------------------------------------------------------

def hash32(x: np.ndarray) -> np.ndarray:
    """32-bit finaliser (lowbias32); identical to scone_hash32 in csrc/scone_common.h."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def synth_rows_i8(seed: int, row_ids: np.ndarray, d: int) -> np.ndarray:
    """Row i, 4-byte word w = hash32(seed ^ hash32(i_lo + 0x9E3779B9*i_hi) + w); bytes
    little-endian -> int8.  Same as the synthetic fill kernel in csrc."""
    row_ids = np.asarray(row_ids, dtype=np.uint64)
    lo = (row_ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (row_ids >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        base = hash32(lo + np.uint32(0x9E3779B9) * hi) ^ np.uint32(seed)
        w = np.arange(d // 4, dtype=np.uint32)
        words = hash32(base[:, None] + w[None, :])
    return words.view(np.uint8).reshape(len(row_ids), d).view(np.int8)


def synth_rows_i4(seed: int, row_ids: np.ndarray, d: int, base_scale: float) -> Tuple[np.ndarray, np.ndarray]:
    """INT4 form of the synthetic fill (k_fill_synth<I4> in csrc/scone_table.hip): payload word w of row i =
    hash32(row_base(i) + w) for w < d/8 (eight offset-binary nibbles per word, element 8w+b in bits 4b..4b+3),
    group scale g = synth_scale_f16(counter = i * (d/128) + g).  Returns (packed [n, d/2] uint8, scales [n, d/128] f16)."""
    row_ids = np.asarray(row_ids, dtype=np.uint64)
    lo = (row_ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (row_ids >> np.uint64(32)).astype(np.uint32)
    ng = d // I4_GROUP
    with np.errstate(over="ignore"):
        base = hash32(lo + np.uint32(0x9E3779B9) * hi) ^ np.uint32(seed)
        words = hash32(base[:, None] + np.arange(d // 8, dtype=np.uint32)[None, :])
    packed = np.ascontiguousarray(words).view(np.uint8).reshape(len(row_ids), d // 2)
    counters = row_ids[:, None] * np.uint64(ng) + np.arange(ng, dtype=np.uint64)[None, :]
    scales = synth_scale_f16(seed, counters.reshape(-1), base_scale).reshape(len(row_ids), ng)
    return packed, scales


def synth_scale_f16(seed: int, row_ids: np.ndarray, base_scale: float) -> np.ndarray:
    """fp16 scale in [0.5, 1.5) * base_scale, from hash32(seed + 0x51ED27 ^ row)."""
    row_ids = np.asarray(row_ids, dtype=np.uint64)
    lo = (row_ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (row_ids >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = hash32((hash32(lo + np.uint32(0x9E3779B9) * hi) ^ np.uint32(seed)) + np.uint32(0x51ED27))
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(base_scale) * (np.float32(0.5) + u)).astype(np.float16)
