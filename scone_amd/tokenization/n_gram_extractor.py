"""f-gram vocabulary + per-token n-gram match, MI355X-native.

Mirrors ``scone/tokenization/n_gram_extractor.py`` of the reference: same class
name, constructor, attributes (``f_grams``, ``f_gram_to_id``, ``id_to_f_gram``),
``fit`` / ``get_token_f_grams`` / ``save`` / ``load`` signatures and file format.
The hot method, :meth:`get_token_f_grams` (reference lines 106-126), runs on the
GPU: tokens are matched against an exact-key hash index in HBM by
``scone_match_csr`` (include/scone_hip.h).  There is no CPU fallback for it.
"""

from collections import Counter
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np


class NGramExtractor:
    """Extracts n-grams and identifies frequent n-grams (f-grams).

    Attributes (as in the reference, n_gram_extractor.py:18-24):
        max_n, min_freq, max_f_grams, f_grams, f_gram_to_id, id_to_f_gram.
    """

    def __init__(self, max_n: int = 3, min_freq: int = 100, max_f_grams: int = 10_000_000) -> None:
        if not 1 <= int(max_n) <= 4:
            raise ValueError("scone_amd supports max_n in 1..4 (reference configs use 3 and 4)")
        self.max_n = int(max_n)
        self.min_freq = min_freq
        self.max_f_grams = max_f_grams
        self._f_gram_to_id: Optional[Dict[Tuple[int, ...], int]] = {}
        self._id_to_f_gram: Optional[Dict[int, Tuple[int, ...]]] = {}
        self._f_grams: Optional[Set[Tuple[int, ...]]] = set()
        self._keys: Optional[np.ndarray] = None   # [N, max_n] uint32, row == id
        self._lens: Optional[np.ndarray] = None   # [N] uint8
        self._index = None                        # device index (hip_backend.SconeTable, dim = 0)
        self._index_sig = None

    # ------------------------------------------------------------------ views
    # The three dict/set attributes are materialised lazily when the extractor was
    # created from arrays (large synthetic vocabularies never need them).
    @property
    def f_gram_to_id(self) -> Dict[Tuple[int, ...], int]:
        if self._f_gram_to_id is None:
            d: Dict[Tuple[int, ...], int] = {}
            for i in range(self._lens.shape[0]):
                d.setdefault(tuple(int(x) for x in self._keys[i, :self._lens[i]]), i)
            self._f_gram_to_id = d
        return self._f_gram_to_id

    @f_gram_to_id.setter
    def f_gram_to_id(self, value: Dict[Tuple[int, ...], int]) -> None:
        self._f_gram_to_id = value
        self._keys = self._lens = None
        self._index = None

    @property
    def id_to_f_gram(self) -> Dict[int, Tuple[int, ...]]:
        if self._id_to_f_gram is None:
            self._id_to_f_gram = {v: k for k, v in self.f_gram_to_id.items()}
        return self._id_to_f_gram

    @id_to_f_gram.setter
    def id_to_f_gram(self, value) -> None:
        self._id_to_f_gram = value

    @property
    def f_grams(self) -> Set[Tuple[int, ...]]:
        if self._f_grams is None:
            self._f_grams = set(self.f_gram_to_id.keys())
        return self._f_grams

    @f_grams.setter
    def f_grams(self, value) -> None:
        self._f_grams = value

    def __len__(self) -> int:
        if self._lens is not None and self._f_gram_to_id is None:
            return int(self._lens.shape[0])
        return len(self.f_gram_to_id)

    @property
    def num_f_grams(self) -> int:
        return len(self)

    # ------------------------------------------------------------------ host utilities
    def extract_n_grams(self, token_ids: List[int], n: int) -> List[Tuple[int, ...]]:
        """n_gram_extractor.py:46-57."""
        return [tuple(token_ids[i:i + n]) for i in range(len(token_ids) - n + 1)]

    def extract_all_n_grams(self, token_ids: List[int]) -> List[Tuple[int, ...]]:
        """n_gram_extractor.py:59-70."""
        all_n_grams = []
        for n in range(1, min(self.max_n + 1, len(token_ids) + 1)):
            all_n_grams.extend(self.extract_n_grams(token_ids, n))
        return all_n_grams

    def fit(self, tokenized_texts: Iterable[Sequence[int]], verbose: bool = True) -> "NGramExtractor":
        """Identify f-grams from a corpus (n_gram_extractor.py:72-104).

        Host-side (vocabulary construction is off the lookup path):
        count-descending ``Counter.most_common`` order, ties in first-seen order,
        ids dense from 0.
        """
        counter: Counter = Counter()
        for token_ids in tokenized_texts:
            counter.update(self.extract_all_n_grams(list(token_ids)))
        frequent = [g for g, c in counter.most_common(self.max_f_grams) if c >= self.min_freq]
        self._set_from_list(frequent)
        if verbose:
            print(f"Extracted {len(self.f_grams)} f-grams")
        return self

    def fit_gpu(self, tokenized_texts: Iterable[Sequence[int]], verbose: bool = True, device=None) -> "NGramExtractor":
        """Same result as :meth:`fit` (identical f-grams and ids), computed on the GPU by ``scone_fit``:
        hash-table counting of every n-gram + two stable radix sorts (count descending, first-seen
        ascending).  Needs a GPU; the vocabulary stays in array form (see :meth:`from_arrays`)."""
        import torch
        from scone_amd.hip_backend import fit_gpu
        texts = [np.asarray(t, dtype=np.int64) for t in tokenized_texts]
        lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
        offsets = np.zeros(len(texts) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        flat = np.concatenate(texts) if texts else np.zeros(0, dtype=np.int64)
        if flat.size and (flat.min() < 0 or flat.max() > 2**31 - 2):
            raise ValueError("fit_gpu: token ids must be in [0, 2**31 - 2]")
        keys, klens, counts, _ = fit_gpu(torch.from_numpy(flat), torch.from_numpy(offsets), self.max_n, self.min_freq,
                                         self.max_f_grams, device=device)
        self._keys, self._lens = np.ascontiguousarray(keys), np.ascontiguousarray(klens)
        self._f_gram_to_id = self._id_to_f_gram = self._f_grams = None
        self._index = None
        self.counts = counts
        if verbose:
            print(f"Extracted {len(self)} f-grams")
        return self

    def _set_from_list(self, grams: List[Tuple[int, ...]]) -> None:
        self._f_grams = set(grams)
        self._f_gram_to_id = {g: i for i, g in enumerate(grams)}
        self._id_to_f_gram = {i: g for i, g in enumerate(grams)}
        self._keys = self._lens = None
        self._index = None

    @classmethod
    def from_arrays(cls, keys: np.ndarray, lens: np.ndarray, max_n: Optional[int] = None, min_freq: int = 1,
                    max_f_grams: Optional[int] = None) -> "NGramExtractor":
        """Vocabulary given as dense arrays: ``keys[N, max_n]`` token ids, ``lens[N]``; row == id.
        On duplicate keys the smallest id wins (the index's rule)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        ex = cls(max_n=max_n or keys.shape[1], min_freq=min_freq, max_f_grams=max_f_grams or max(len(lens), 1))
        if keys.shape != (lens.shape[0], ex.max_n):
            raise ValueError("keys must be [N, max_n]")
        ex._keys, ex._lens = keys, lens
        ex._f_gram_to_id = ex._id_to_f_gram = ex._f_grams = None
        return ex

    def key_arrays(self) -> Tuple[np.ndarray, np.ndarray]:
        """Dense ``(keys[N, max_n] uint32, lens[N] uint8)`` with row == id; N = max id + 1."""
        if self._keys is None:
            d = self.f_gram_to_id
            n = (max(d.values()) + 1) if d else 0
            keys = np.zeros((n, self.max_n), dtype=np.uint32)
            lens = np.zeros(n, dtype=np.uint8)
            for g, i in d.items():
                if not 1 <= len(g) <= self.max_n:
                    raise ValueError(f"f-gram {g} longer than max_n={self.max_n}")
                keys[i, :len(g)] = g
                lens[i] = len(g)
            self._keys, self._lens = keys, lens
        return self._keys, self._lens

    # ------------------------------------------------------------------ device index
    def build_index(self, table) -> None:
        """Insert this vocabulary into ``table``'s device index (ids = row numbers)."""
        keys, lens = self.key_arrays()
        if (lens == 0).any():
            # ids with no f-gram (sparse id space): insert only the real keys, chunked by runs
            idx = np.nonzero(lens)[0]
            start = 0
            while start < idx.size:
                end = start
                while end + 1 < idx.size and idx[end + 1] == idx[end] + 1:
                    end += 1
                a, b = idx[start], idx[end] + 1
                table.index_build(keys[a:b], lens[a:b], id0=int(a))
                start = end + 1
        else:
            table.index_build(keys, lens, id0=0)

    def device_index(self, device=None):
        """Index-only device handle for :meth:`get_token_f_grams`; rebuilt when the vocabulary changes."""
        from scone_amd.hip_backend import SconeTable
        sig = (id(self._f_gram_to_id), len(self), id(self._keys))
        if self._index is None or self._index_sig != sig:
            n = self.key_arrays()[1].shape[0]
            t = SconeTable(self.max_n, n, dim=0, device=device)
            self.build_index(t)
            self._index, self._index_sig = t, sig
        return self._index

    # ------------------------------------------------------------------ hot path
    def get_token_f_grams(self, token_ids: List[int]) -> Dict[int, List[Tuple[int, ...]]]:
        """All f-grams containing each token (n_gram_extractor.py:106-126), matched on the GPU.

        Per position: n ascending, then window start ascending, duplicates kept.
        """
        import torch
        result: Dict[int, List[Tuple[int, ...]]] = {i: [] for i in range(len(token_ids))}
        if len(token_ids) == 0:
            return result
        index = self.device_index()
        tok = torch.as_tensor(np.asarray(token_ids, dtype=np.int64).clip(-1, 2**31 - 1), dtype=torch.int32)
        offsets, ids = index.match_csr(tok)
        offsets = offsets.cpu().tolist()                      # plain Python ints: list slicing beats numpy scalar indexing
        grams = list(map(self.id_to_f_gram.__getitem__, ids.cpu().tolist()))
        for pos in range(len(token_ids)):
            result[pos] = grams[offsets[pos]:offsets[pos + 1]]
        return result

    def get_token_f_grams_batch(self, input_ids) -> List[Dict[int, List[Tuple[int, ...]]]]:
        """:meth:`get_token_f_grams` for every row of ``input_ids [B, T]`` with ONE GPU match
        (the reference loops over sequences, f_gram_tokenizer.py:121-123)."""
        import torch
        ids = torch.as_tensor(input_ids)
        if ids.dim() != 2:
            raise ValueError("input_ids must be [B, T]")
        B, T = ids.shape
        if B == 0 or T == 0:
            return [{} for _ in range(B)]
        index = self.device_index()
        offsets, flat = index.match_csr(ids.clamp(-1, 2**31 - 1).to(torch.int32))
        offsets = offsets.cpu().tolist()
        grams = list(map(self.id_to_f_gram.__getitem__, flat.cpu().tolist()))
        out = []
        for b in range(B):
            base = b * T
            out.append({pos: grams[offsets[base + pos]:offsets[base + pos + 1]] for pos in range(T)})
        return out

    # ------------------------------------------------------------------ persistence
    def save(self, path: str) -> None:
        """Same on-disk format as the reference (n_gram_extractor.py:128-141)."""
        data = {
            "max_n": self.max_n,
            "min_freq": self.min_freq,
            "max_f_grams": self.max_f_grams,
            "f_gram_to_id": {",".join(map(str, k)): v for k, v in self.f_gram_to_id.items()},
        }
        np.save(path, data, allow_pickle=True)

    @classmethod
    def load(cls, path: str) -> "NGramExtractor":
        """Reads files written by the reference's ``NGramExtractor.save`` (:143-165)."""
        data = np.load(path, allow_pickle=True).item()
        extractor = cls(max_n=data["max_n"], min_freq=data["min_freq"], max_f_grams=data["max_f_grams"])
        extractor.f_gram_to_id = {tuple(map(int, k.split(","))): v for k, v in data["f_gram_to_id"].items()}
        extractor.id_to_f_gram = {v: k for k, v in extractor.f_gram_to_id.items()}
        extractor.f_grams = set(extractor.f_gram_to_id.keys())
        return extractor
