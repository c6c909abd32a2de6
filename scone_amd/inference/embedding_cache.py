"""f-gram embedding table + lookup, MI355X-native.

Mirrors ``scone/inference/embedding_cache.py`` of the reference: same class name,
constructor, attributes and method signatures (``cache_embeddings``,
``get_embeddings``, ``get_token_embeddings``, ``save``, ``load``), the same
exceptions and the same on-disk formats.  Host attributes (``embeddings`` dict /
``memory_mapped_embeddings``) are kept as the reference keeps them; every *lookup*
is served from a device copy of the table (fp32 / fp16 / INT8 / INT4 rows in HBM
or in pinned host memory) by the kernels behind ``include/scone_hip.h``.  There is
no CPU fallback for lookups.

Additive API (not in the reference): :meth:`embed_tokens` -- the fused
match -> gather -> dequantise -> mean -> ``+ wte + wpe`` path, :meth:`match`,
:meth:`to_device`, :meth:`from_synthetic`.
"""

import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from scone_amd.tokenization.n_gram_extractor import NGramExtractor


class EmbeddingCache:
    """Cache for f-gram embeddings (reference: embedding_cache.py:13-54).

    Extra keyword-only arguments select the device representation:
        table_format: "fp32" (reference-exact), "fp16", "int8", "int4".
        placement:    "hbm" or "pinned_host" (rows >= hot_rows stay in host DRAM, read over PCIe).
        hot_rows:     with "pinned_host", the head of the table (ids are frequency-ordered) kept in HBM.
        stage_tokens: with "pinned_host": 0 = rows are read in place over PCIe by the lookup kernel; > 0 =
                      prefetch through an HBM cache of cold rows (chunks of about this many tokens; the cold rows a
                      chunk references that are not cached are copied host -> HBM once, on side streams, while the
                      previous chunk is reduced; cached rows stay resident across chunks, batches and calls --
                      the counterpart of the page cache under the reference's memory-mapped table,
                      embedding_cache.py:76-91, 132-135).
        cache_rows:   row slots of that cache (0 = what the chunk pipeline needs: 42 x stage_tokens for max_n = 3).
        lookup_mode:  "cover" = the reference code (all covering f-grams, mean, added to wte);
                      "longest_suffix" = the paper's Algorithm 2 (longest f-gram of length >= 2 ending
                      at the token replaces wte; causal) -- affects embed_tokens only.
        device:       HIP device for the table (default: current device).
        keep_host_copy: keep the reference's host dict / memmap (needed by ``save``).
    """

    def __init__(self, n_gram_extractor: NGramExtractor, embedding_dim: int, cache_dir: Optional[str] = None,
                 use_memory_map: bool = False, *, table_format: str = "fp32", placement: str = "hbm",
                 device=None, keep_host_copy: bool = True, hot_rows: int = 0, lookup_mode: str = "cover",
                 stage_tokens: int = 0, cache_rows: int = 0) -> None:
        self.n_gram_extractor = n_gram_extractor
        self.embedding_dim = embedding_dim
        self.cache_dir = cache_dir
        self.use_memory_map = use_memory_map

        self.embeddings: Dict[int, np.ndarray] = {}
        self.memory_mapped_embeddings: Optional[np.ndarray] = None

        self.table_format = table_format
        self.placement = placement
        self.hot_rows = int(hot_rows)          # placement='pinned_host': rows [0, hot_rows) stay in HBM
        self.keep_host_copy = keep_host_copy
        self.lookup_mode = lookup_mode        # 'cover' (reference code) or 'longest_suffix' (paper, Algorithm 2)
        self.stage_tokens = int(stage_tokens)  # placement='pinned_host': > 0 = prefetch through an HBM cache of cold rows, in chunks
        self.cache_rows = int(cache_rows)      # ... its row slots (0 = the pipeline's minimum)
        self._device = device
        self._table = None           # hip_backend.SconeTable
        self._dirty = True           # host rows changed since the last upload
        self._present: Optional[np.ndarray] = None   # which ids have a row (in-memory variant, no host copy)

        if self.cache_dir is not None and not os.path.exists(self.cache_dir):
            os.makedirs(self.cache_dir)

    # ------------------------------------------------------------------ ingestion (a8)
    def cache_embeddings(self, f_gram_ids: Union[Sequence[int], Dict[int, torch.Tensor]],
                         embeddings: Optional[torch.Tensor] = None, verbose: bool = True) -> None:
        """Cache embeddings for f-grams (embedding_cache.py:56-111).

        Also accepts the ``{id: tensor}`` form that the reference's own callers pass
        (precompute_embeddings.py:138, simple_example.py:111).
        """
        if isinstance(f_gram_ids, dict):
            items = f_gram_ids
            f_gram_ids = list(items.keys())
            embeddings = (torch.stack([torch.as_tensor(items[i], dtype=torch.float32).reshape(-1)
                                       for i in f_gram_ids])
                          if f_gram_ids else torch.zeros(0, self.embedding_dim))
        if embeddings is None:
            raise TypeError("cache_embeddings() missing required argument: 'embeddings'")
        ids = np.asarray(list(f_gram_ids), dtype=np.int64)
        rows = embeddings[:len(ids)] if len(embeddings) > len(ids) else embeddings
        ids = ids[:len(rows)]                                     # zip() semantics of the reference
        if self.use_memory_map:
            if self.memory_mapped_embeddings is None:
                if self.cache_dir is None:
                    raise ValueError("Cache directory must be provided for memory mapping")
                mmap_path = os.path.join(self.cache_dir, "embeddings.npy")
                shape = (len(self.n_gram_extractor.f_grams), self.embedding_dim)
                # raw row-major fp32 [N, d], no header, zero-filled (embedding_cache.py:84-91)
                mode = "r+" if os.path.exists(mmap_path) and os.path.getsize(mmap_path) == shape[0] * shape[1] * 4 else "w+"
                self.memory_mapped_embeddings = np.memmap(mmap_path, dtype=np.float32, mode=mode, shape=shape)
                if mode == "w+":
                    self.memory_mapped_embeddings[:] = 0.0
            if len(ids):
                self.memory_mapped_embeddings[ids] = rows.detach().to("cpu", torch.float32).numpy()
            if hasattr(self.memory_mapped_embeddings, "flush"):
                self.memory_mapped_embeddings.flush()
        elif self.keep_host_copy:
            host = rows.detach().to("cpu", torch.float32).numpy()
            for k, f_gram_id in enumerate(ids.tolist()):
                self.embeddings[f_gram_id] = host[k]
        else:
            # device-only ingestion: rows go straight into the device table (quantised on the GPU)
            table = self._device_only_table()
            if len(ids):
                if int(ids.min()) < 0 or int(ids.max()) >= table.n_rows:
                    raise IndexError(f"f-gram id outside the table (size {table.n_rows})")
                table.store_f32(rows, ids=torch.from_numpy(ids))
                self._present[ids] = True
            return
        self._dirty = True

    # ------------------------------------------------------------------ device table
    def _n_rows(self) -> int:
        n = len(self.n_gram_extractor)
        if self.use_memory_map and self.memory_mapped_embeddings is not None:
            n = max(n, self.memory_mapped_embeddings.shape[0])
        elif self.embeddings:
            n = max(n, max(self.embeddings.keys()) + 1)
        if self._present is not None:
            n = max(n, self._present.shape[0])
        return n

    def _make_table(self, n_rows: int):
        from scone_amd.hip_backend import SconeTable
        table = SconeTable(self.n_gram_extractor.max_n, n_rows, dim=self.embedding_dim,
                           table_format=self.table_format, placement=self.placement, device=self._device,
                           hot_rows=self.hot_rows, lookup_mode=self.lookup_mode,
                           stage_tokens=self.stage_tokens, cache_rows=self.cache_rows)
        self.n_gram_extractor.build_index(table)
        return table

    def _device_only_table(self):
        if self._table is None:
            n = len(self.n_gram_extractor)
            self._table = self._make_table(n)
            self._present = np.zeros(n, dtype=bool)
            self._dirty = False
        return self._table

    def to_device(self, device=None, table_format: Optional[str] = None, placement: Optional[str] = None):
        """(Re)build the device table + index from the host rows; returns the backend handle."""
        device_only = not self.keep_host_copy and not self.use_memory_map
        if device_only:
            if self._table is not None and ((table_format not in (None, self.table_format))
                                            or (placement not in (None, self.placement))):
                raise RuntimeError("a device-only cache (keep_host_copy=False) cannot be re-laid out")
            if device is not None and self._table is None:
                self._device = device
            if table_format is not None and self._table is None:
                self.table_format = table_format
            if placement is not None and self._table is None:
                self.placement = placement
            return self._device_only_table()
        if device is not None:
            self._device = device
        if table_format is not None and table_format != self.table_format:
            self.table_format, self._dirty = table_format, True
        if placement is not None and placement != self.placement:
            self.placement, self._dirty = placement, True
        if self._table is not None and not self._dirty:
            return self._table
        table = self._make_table(self._n_rows())
        chunk = max(1, (64 << 20) // (4 * self.embedding_dim))
        if self.use_memory_map:
            mm = self.memory_mapped_embeddings
            if mm is not None:
                for a in range(0, mm.shape[0], chunk):
                    table.store_f32(torch.from_numpy(np.ascontiguousarray(mm[a:a + chunk])), row0=a)
        elif self.embeddings:
            ids = np.fromiter(self.embeddings.keys(), dtype=np.int64, count=len(self.embeddings))
            for a in range(0, len(ids), chunk):
                part = ids[a:a + chunk]
                rows = np.stack([np.asarray(self.embeddings[int(i)], dtype=np.float32).reshape(-1) for i in part])
                table.store_f32(torch.from_numpy(rows), ids=torch.from_numpy(part))
        self._table = table
        self._dirty = False
        return table

    @property
    def table(self):
        """The device handle (built on first use)."""
        return self.to_device()

    @classmethod
    def from_synthetic(cls, n_gram_extractor: NGramExtractor, embedding_dim: int, *, table_format: str = "int8",
                       seed: int = 7, base_scale: float = 0.02 / 127, placement: str = "hbm", device=None,
                       n_rows: Optional[int] = None, hot_rows: int = 0, lookup_mode: str = "cover",
                       stage_tokens: int = 0, cache_rows: int = 0) -> "EmbeddingCache":
        """Cache whose device table is generated on the GPU by the counter-based hash of
        ``scone_table_fill_synthetic`` (bench / full-size tests; nothing materialised on the host)."""
        cache = cls(n_gram_extractor, embedding_dim, table_format=table_format, placement=placement, device=device,
                    keep_host_copy=False, hot_rows=hot_rows, lookup_mode=lookup_mode, stage_tokens=stage_tokens,
                    cache_rows=cache_rows)
        n = int(n_rows if n_rows is not None else len(n_gram_extractor))
        table = cache._make_table(n)
        table.fill_synthetic(seed, base_scale)
        cache._table, cache._dirty = table, False
        cache._present = np.ones(n, dtype=bool)
        return cache

    # ------------------------------------------------------------------ lookups (a4, a5)
    def _validate_ids(self, f_gram_ids: Sequence[int]) -> np.ndarray:
        ids = np.asarray(list(f_gram_ids), dtype=np.int64).reshape(-1)
        if self.use_memory_map:
            if self.memory_mapped_embeddings is None:
                raise ValueError("Memory-mapped embeddings not initialized")
            n = self.memory_mapped_embeddings.shape[0]
            bad = (ids < -n) | (ids >= n)
            if bad.any():
                raise IndexError(f"index {int(ids[bad][0])} is out of bounds for axis 0 with size {n}")
            ids = np.where(ids < 0, ids + n, ids)                 # numpy fancy-index semantics
        elif self.keep_host_copy:
            for i in ids.tolist():
                if i not in self.embeddings:
                    raise KeyError(i)
        else:
            n = 0 if self._present is None else self._present.shape[0]
            ok = (ids >= 0) & (ids < n)
            if n:
                ok &= self._present[np.clip(ids, 0, n - 1)]
            if not ok.all():
                raise KeyError(int(ids[~ok][0]))
        return ids

    def get_embeddings(self, f_gram_ids: List[int], device: Optional[torch.device] = None) -> torch.Tensor:
        """Rows for f-gram ids as fp32 ``[K, d]`` (embedding_cache.py:113-147); a fresh tensor every call.

        ``device=None`` returns a CPU tensor as the reference does."""
        ids = self._validate_ids(f_gram_ids)
        if ids.size == 0 and not self.use_memory_map:
            raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) in the reference
        table = self.to_device()
        out = table.gather_rows(torch.from_numpy(ids))
        return out.cpu() if device is None else out.to(device)

    def get_token_embeddings(self, token_ids: List[int],
                             device: Optional[torch.device] = None) -> Dict[int, torch.Tensor]:
        """``{position: Tensor[K_pos, d]}`` for every position covered by at least one f-gram
        (embedding_cache.py:149-181); positions with K = 0 are omitted."""
        if len(token_ids) == 0:
            return {}
        table = self.to_device()
        tok = torch.as_tensor(np.asarray(token_ids, dtype=np.int64).clip(-1, 2**31 - 1), dtype=torch.int32)
        offsets, ids = table.match_csr(tok)
        off = offsets.cpu().numpy()
        ids_host = ids.cpu().numpy()
        if not self.use_memory_map and ids_host.size:
            self._validate_ids(np.unique(ids_host))               # KeyError for an f-gram without a cached row
        rows = table.gather_rows(ids.to(torch.int64)) if ids_host.size else None
        if rows is not None:
            rows = rows.cpu() if device is None else rows.to(device)
        if rows is None:
            return {}
        # disjoint slices of a tensor made for this call (fresh, as in the reference); one split call instead of T slicings
        counts = np.diff(off)
        parts = rows.split(counts.tolist())
        return {pos: parts[pos] for pos in np.nonzero(counts)[0].tolist()}

    # ------------------------------------------------------------------ additive fused API
    def match(self, input_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-position f-gram id lists as CSR over the flattened ``B*T`` positions."""
        return self.to_device().match_csr(torch.as_tensor(input_ids))

    def embed_tokens(self, input_ids: torch.Tensor, *, reduce: str = "mean", wte: Optional[torch.Tensor] = None,
                     wpe: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                     out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None,
                     check: bool = False, base: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused lookup for ``input_ids [B, T]`` -> ``[B, T, d]``:

            out[b, t] = (wte[input_ids[b, t]] + reduce_k row(f_gram_k)) + wpe[position_ids[b, t]]

        i.e. get_token_f_grams + id map + get_embeddings + ``mean(dim=0)`` + zero-fill
        (engine.py:234-266) and, when ``wte``/``wpe`` are given, the combine of
        ``SconeLanguageModel.forward`` (language_model.py:239-254) with the bias-free
        projection folded into the table.  ``check=True`` synchronises and raises
        ``IndexError`` for token / position ids outside ``wte`` / ``wpe``.

        ``base [B, T, d]`` instead of ``wte`` / ``wpe``: a dense tensor the caller has already computed (the
        ``inputs_embeds`` of language_model.py:239-243, say) -- ``out = base + reduce_k row(f_gram_k)``, one rounding to
        ``out_dtype`` (default: ``base``'s dtype); the match produces CSR lists and ``scone_gather_reduce`` consumes them.
        """
        if base is not None:
            if wte is not None or wpe is not None or position_ids is not None or out is not None:
                raise ValueError("base= replaces wte / wpe / position_ids (and takes no out=)")
            table = self.to_device()
            tok = torch.as_tensor(input_ids)
            if tok.dim() == 1:
                tok = tok.unsqueeze(0)
            B, T = tok.shape
            if tuple(base.shape) != (B, T, self.embedding_dim):
                raise ValueError(f"base must be [{B}, {T}, {self.embedding_dim}]")
            if out_dtype is None:
                out_dtype = base.dtype
            offsets, ids = table.match_csr(tok)
            return table.gather_reduce(offsets, ids, reduce, base=base.reshape(B * T, self.embedding_dim),
                                       out_dtype=out_dtype).view(B, T, self.embedding_dim)
        table = self.to_device()
        result = table.embed(torch.as_tensor(input_ids), wte=wte, wpe=wpe, position_ids=position_ids, reduce=reduce,
                             out_dtype=out_dtype, out=out)
        if check and table.status() & 1:
            raise IndexError("index out of range in self")
        return result

    def prefetch_tokens(self, input_ids: torch.Tensor, tokens_ready: bool = False) -> None:
        """Tables in pinned host DRAM with ``stage_tokens > 0``: start fetching the cold rows of the NEXT batch now
        (``scone_embed_prefetch``: its first chunks are matched, placed in the HBM cache and copied on side streams behind the
        current stream).  ``input_ids`` must be the very int32 device tensor ``[B, T]`` the later :meth:`embed_tokens` gets,
        unchanged in between.  A serving loop calls it right after ``embed_tokens`` of the current batch -- with
        ``tokens_ready=True`` when the next tokens are complete (uploaded earlier): the prefetch then runs BESIDE the lookup
        just queued instead of behind it -- and never pays the pipeline's fill; other tables: a no-op (matching the next batch
        ahead on a side stream was measured slower than match-then-gather on one stream: profiles/r05b).  New here (the
        reference's memmap faults rows in on first use, embedding_cache.py:132-135)."""
        tok = torch.as_tensor(input_ids)
        if not (tok.dim() == 2 and tok.dtype == torch.int32 and tok.is_cuda and tok.is_contiguous()):
            raise ValueError("prefetch_tokens needs the contiguous int32 device tensor [B, T] that embed_tokens will get")
        self.to_device().embed_prefetch(tok, tokens_ready)

    def alloc_output(self, input_ids: torch.Tensor, *, wte: Optional[torch.Tensor] = None, wpe: Optional[torch.Tensor] = None,
                     out_dtype: Optional[torch.dtype] = None, candidates: int = 8, trials: int = 5):
        """An output buffer ``[B, T, d]`` for a loop that re-uses it (a server with a static batch shape; ``bench.py``), chosen by
        MEASUREMENT: the lookup kernel's time follows the physical placement of the buffer it writes -- 0.616 ... 0.657 ms over
        five 1.6-GB allocations of one process on the headline workload, the same whatever the table, stable per allocation,
        not a matter of alignment or of the offset inside an allocation (``profiles/r06m``) -- and nothing the library or
        the caller can ask the driver for decides it; the first few allocations of a process tend to be the slow ones.  So:
        ``candidates`` allocations, ``trials`` timed lookups of
        ``input_ids`` into each (HIP events around the kernel), a re-match of the three fastest, the winner is kept, the others go
        back to the driver.  Returns
        ``(out, report)``, ``report`` = the kernel milliseconds of every candidate and the index kept.  Costs
        ``candidates * (trials + 1)`` lookups and, for a moment, ``candidates`` buffers."""
        table = self.to_device()
        tok = torch.as_tensor(input_ids)
        if tok.dim() == 1:
            tok = tok.unsqueeze(0)
        B, T = tok.shape
        if out_dtype is None:
            out_dtype = wte.dtype if wte is not None else (wpe.dtype if wpe is not None else torch.float32)
        n = max(1, int(candidates))
        bufs = []
        for _ in range(n):
            try:
                bufs.append(torch.empty((B, T, self.embedding_dim), dtype=out_dtype, device=table.device))
            except torch.cuda.OutOfMemoryError:      # fewer candidates than asked for: choose among those that fit
                if not bufs:
                    raise
                break
        n = len(bufs)
        if n == 1:
            return bufs[0], {"candidates": 1, "kernel_ms": [None], "kept": 0}
        for o in bufs:
            table.embed(tok, wte=wte, wpe=wpe, out=o)                    # untimed: first touch of EVERY buffer before any is timed
        torch.cuda.synchronize(table.device)                             # (nothing of the allocations' own work is left in flight)
        times = []
        for o in bufs:
            table.profile_enable(True)
            table.profile_read(reset=True)
            for _ in range(max(1, trials)):
                table.embed(tok, wte=wte, wpe=wpe, out=o)
            k, ms = table.profile_read(reset=True)
            table.profile_enable(False)
            times.append(ms / max(k, 1))
        # second look at the three fastest, interleaved (round-robin) so that drift hits them alike: the minimum of `candidates`
        # noisy averages is biased low, and the buffer that wins a re-match is the one a long loop will see
        finalists = sorted(range(n), key=lambda i: times[i])[:min(3, n)]
        again = {i: 0.0 for i in finalists}
        for _ in range(max(1, trials)):
            for i in finalists:
                table.profile_enable(True)
                table.profile_read(reset=True)
                table.embed(tok, wte=wte, wpe=wpe, out=bufs[i])
                k, ms = table.profile_read(reset=True)
                table.profile_enable(False)
                again[i] += ms / max(k, 1)
        again = {i: v / max(1, trials) for i, v in again.items()}
        kept = min(finalists, key=lambda i: again[i])
        out = bufs[kept]
        del bufs, o
        torch.cuda.empty_cache()                                        # the rejected blocks go back to the driver, not into torch's cache
        return out, {"candidates": n, "kernel_ms": [float(t) for t in times], "finalists_kernel_ms": {int(i): float(v) for i, v in again.items()},
                     "kept": int(kept)}

    # ------------------------------------------------------------------ native shard format
    NATIVE_MAGIC = "scone_amd.table.v1"          # round 1-2: one uncompressed .npz, whole arrays in host memory
    NATIVE_MAGIC_V2 = "scone_amd.table.v2"       # round 3: one memory-mapped .npy, written and read in chunks

    @staticmethod
    def _native_layout(sections):
        """Byte offsets of the sections of a v2 file inside the payload of its ``.npy``: ``[u64 header length | json
        header | pad to 4096 | section | pad | section ...]``; returns (offset of every section, total bytes)."""
        off, pos = {}, 0
        for name, (dtype, shape) in sections.items():
            pos = (pos + 4095) // 4096 * 4096
            off[name] = pos
            pos += int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        return off, (pos + 4095) // 4096 * 4096

    def save_native(self, path: str, chunk_rows: int = 1 << 18, with_index: bool = False) -> None:
        """Write the device table as it is stored (quantised rows + scales) together with the f-gram keys, so
        :meth:`load_native` restores it without re-quantising -- the rows THIS handle owns only (a shard of a 1e9-row table
        writes its 125M rows, not a 528 GB array) and, with ``with_index=True``, the built device index (hash slots,
        unigram table, presence bitmap) so that a load copies it back instead of re-inserting every key (1e9 keys: ~16 s).

        ONE file, a valid ``.npy`` of bytes whose payload is ``[header | keys | lens | rows | scales | index blobs]``
        (4096-aligned sections, described by the json header), created as a memory map and filled ``chunk_rows`` rows at
        a time straight from the device: the host never holds more than one chunk beside the file's own page cache, whatever
        the shard's size (a C5 shard: 66 GB of rows + a 34 GB index).  A vocabulary without key arrays (the synthetic
        ``StructuredVocab``) is recorded by its parameters."""
        import json
        table = self.to_device()
        n, a, b = table.n_rows, table.row_begin, table.row_end
        spr = table.scales_per_row()
        ex = self.n_gram_extractor
        meta = {"magic": self.NATIVE_MAGIC_V2, "table_format": self.table_format, "embedding_dim": self.embedding_dim,
                "max_n": ex.max_n, "n_rows": n, "row_begin": a, "row_end": b, "rows_are_local": True}
        sections = {}
        keys = lens = None
        if hasattr(ex, "key_arrays"):
            keys, lens = ex.key_arrays()
            sections["keys"] = ("uint32", tuple(keys.shape))
            sections["lens"] = ("uint8", tuple(lens.shape))
        else:                                             # closed-form vocabulary: its parameters are the keys
            meta["vocabulary"] = {"kind": type(ex).__name__, "n_rows": len(ex), "vocab": getattr(ex, "vocab", None)}
        sections["rows"] = ("uint8", (b - a, table.payload_bytes()))
        sections["scales"] = ("float16", (b - a, spr))
        if with_index:
            sb, ub, bb = table.index_blob_sizes()
            sections["index_slots"] = ("uint8", (sb,))
            sections["index_uni"] = ("int32", (ub // 4,))
            sections["index_bloom"] = ("uint8", (bb,))
            meta["index_capacity"] = sb // 16
        for _ in range(2):                                # (the header's own length moves the first section once)
            meta["sections"] = {k: {"dtype": v[0], "shape": list(v[1])} for k, v in sections.items()}
            hdr = json.dumps(meta).encode()
            hdr_room = (8 + len(hdr) + 4095 + 64) // 4096 * 4096      # 64 B of slack for "index_keys" below
            off, body = self._native_layout(sections)
            meta["section_offsets"] = {k: hdr_room + v for k, v in off.items()}
        if not str(path).endswith(".npy"):
            path = str(path) + ".npy"
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(hdr_room + body,))

        def section(name):
            dtype, shape = sections[name]
            o = meta["section_offsets"][name]
            nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
            return mm[o:o + nbytes].view(dtype).reshape(shape)

        if keys is not None:
            section("keys")[:] = keys
            section("lens")[:] = lens
        rows, scales = section("rows"), section("scales")
        for r0 in range(a, b, chunk_rows):
            m = min(chunk_rows, b - r0)
            table.download(r0, m, rows=rows[r0 - a:r0 - a + m], scales=scales[r0 - a:r0 - a + m] if spr else None)
        if with_index:
            _, _, _, n_keys, cap = table.index_export(out=(section("index_slots"), section("index_uni"), section("index_bloom")))
            meta["index_keys"] = int(n_keys)
        hdr = json.dumps(meta).encode()
        assert 8 + len(hdr) <= hdr_room
        mm[:8] = np.frombuffer(np.uint64(len(hdr)).tobytes(), dtype=np.uint8)
        mm[8:8 + len(hdr)] = np.frombuffer(hdr, dtype=np.uint8)
        mm.flush()
        del mm

    @classmethod
    def load_native(cls, path: str, *, placement: str = "hbm", device=None, hot_rows: int = 0,
                    chunk_rows: int = 1 << 18) -> "EmbeddingCache":
        """Restore a cache written by :meth:`save_native` (extractor included; the index from the file when it is there).
        The file is memory-mapped and uploaded ``chunk_rows`` rows at a time (v1 ``.npz`` files of rounds 1-2 are still read,
        whole)."""
        import json
        from scone_amd.hip_backend import SconeTable
        path = str(path)
        if not os.path.exists(path):
            path = path + (".npy" if os.path.exists(path + ".npy") else ".npz")
        if path.endswith(".npz"):
            return cls._load_native_v1(path, placement=placement, device=device, hot_rows=hot_rows)
        mm = np.load(path, mmap_mode="r")
        if mm.dtype != np.uint8 or mm.ndim != 1 or mm.shape[0] < 4096:
            raise ValueError("not a scone_amd native table file")
        hlen = int(np.frombuffer(bytes(mm[:8]), dtype=np.uint64)[0])
        if hlen <= 0 or 8 + hlen > mm.shape[0]:
            raise ValueError("not a scone_amd native table file")
        try:
            meta = json.loads(bytes(mm[8:8 + hlen]).decode())
        except ValueError:
            meta = {}
        if meta.get("magic") != cls.NATIVE_MAGIC_V2:
            raise ValueError("not a scone_amd native table file")

        def section(name):
            sec = meta["sections"][name]
            o = meta["section_offsets"][name]
            nbytes = int(np.prod(sec["shape"], dtype=np.int64)) * np.dtype(sec["dtype"]).itemsize
            return mm[o:o + nbytes].view(sec["dtype"]).reshape(sec["shape"])

        if "keys" in meta["sections"]:
            ex = NGramExtractor.from_arrays(np.array(section("keys")), np.array(section("lens")), max_n=meta["max_n"])
        else:
            voc = meta["vocabulary"]
            if voc["kind"] != "StructuredVocab":
                raise ValueError(f"native table file without keys for an unknown vocabulary kind {voc['kind']!r}")
            from scone_amd.synthetic import StructuredVocab
            ex = StructuredVocab(voc["n_rows"], voc["vocab"], meta["max_n"])
        cache = cls(ex, meta["embedding_dim"], table_format=meta["table_format"], placement=placement, device=device,
                    keep_host_copy=False, hot_rows=hot_rows)
        a, b = meta["row_begin"], meta["row_end"]
        if "index_slots" in meta["sections"]:
            table = SconeTable(ex.max_n, meta["n_rows"], dim=cache.embedding_dim, table_format=cache.table_format,
                               placement=placement, device=device, hot_rows=hot_rows, lookup_mode=cache.lookup_mode,
                               index_capacity=meta["index_capacity"], row_begin=a, row_end=b)
            table.index_import(section("index_slots"), section("index_uni"), section("index_bloom"), meta["index_keys"])
        else:
            table = SconeTable(ex.max_n, meta["n_rows"], dim=cache.embedding_dim, table_format=cache.table_format,
                               placement=placement, device=device, hot_rows=hot_rows, lookup_mode=cache.lookup_mode,
                               row_begin=a, row_end=b)
            ex.build_index(table)
        rows, scales = section("rows"), section("scales")
        for r0 in range(a, b, chunk_rows):
            m = min(chunk_rows, b - r0)
            table.upload(rows[r0 - a:r0 - a + m], scales[r0 - a:r0 - a + m] if scales.shape[1] else None, row0=r0)
        cache._table, cache._dirty = table, False
        cache._present = np.ones(meta["n_rows"], dtype=bool)
        del mm
        return cache

    @classmethod
    def _load_native_v1(cls, path: str, *, placement: str = "hbm", device=None, hot_rows: int = 0) -> "EmbeddingCache":
        import json
        from scone_amd.hip_backend import SconeTable
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        if meta.get("magic") != cls.NATIVE_MAGIC:
            raise ValueError("not a scone_amd native table file")
        ex = NGramExtractor.from_arrays(z["keys"], z["lens"], max_n=meta["max_n"])
        cache = cls(ex, meta["embedding_dim"], table_format=meta["table_format"], placement=placement, device=device,
                    keep_host_copy=False, hot_rows=hot_rows)
        a, b = meta["row_begin"], meta["row_end"]
        if "index_slots" in z.files:
            table = SconeTable(ex.max_n, meta["n_rows"], dim=cache.embedding_dim, table_format=cache.table_format,
                               placement=placement, device=device, hot_rows=hot_rows, lookup_mode=cache.lookup_mode,
                               index_capacity=meta["index_capacity"], row_begin=a, row_end=b)
            table.index_import(z["index_slots"], z["index_uni"], z["index_bloom"], meta["index_keys"])
        elif (a, b) == (0, meta["n_rows"]):
            table = cache._make_table(meta["n_rows"])
        else:
            table = SconeTable(ex.max_n, meta["n_rows"], dim=cache.embedding_dim, table_format=cache.table_format,
                               placement=placement, device=device, hot_rows=hot_rows, lookup_mode=cache.lookup_mode,
                               row_begin=a, row_end=b)
            ex.build_index(table)
        rows, scales = z["rows"], z["scales"]
        off = a if meta.get("rows_are_local") else 0          # files of the first layout hold all n_rows rows
        chunk = 1 << 18
        for r0 in range(a, b, chunk):
            m = min(chunk, b - r0)
            table.upload(rows[r0 - off:r0 - off + m], scales[r0 - off:r0 - off + m] if scales.shape[1] else None, row0=r0)
        cache._table, cache._dirty = table, False
        cache._present = np.ones(meta["n_rows"], dtype=bool)
        return cache

    # ------------------------------------------------------------------ persistence (a9)
    def save(self, path: str) -> None:
        """Same on-disk format as the reference (embedding_cache.py:183-203)."""
        if self.use_memory_map:
            np.save(path, {"use_memory_map": True, "cache_dir": self.cache_dir,
                           "embedding_dim": self.embedding_dim})
        else:
            if not self.keep_host_copy:
                raise RuntimeError("save() needs the host copy of the rows (keep_host_copy=True)")
            np.save(path, {"use_memory_map": False, "embeddings": self.embeddings,
                           "embedding_dim": self.embedding_dim})

    @classmethod
    def load(cls, path: str, n_gram_extractor: NGramExtractor, cache_dir: Optional[str] = None,
             use_memory_map: Optional[bool] = None, **device_kwargs) -> "EmbeddingCache":
        """Load a cache written by the reference or by :meth:`save` (embedding_cache.py:205-243).

        ``cache_dir`` / ``use_memory_map`` are accepted because the reference's own callers
        pass them (engine.py:180, tests/test_embedding_cache.py:136); the file decides.
        """
        data = np.load(path, allow_pickle=True).item()
        if data["use_memory_map"]:
            cache = cls(n_gram_extractor=n_gram_extractor, embedding_dim=data["embedding_dim"],
                        cache_dir=data["cache_dir"], use_memory_map=True, **device_kwargs)
            mmap_path = os.path.join(data["cache_dir"], "embeddings.npy")
            try:
                cache.memory_mapped_embeddings = np.load(mmap_path, mmap_mode="r")
            except ValueError:
                # the reference writes the file raw (np.memmap, no .npy header): [N, d] fp32
                n = os.path.getsize(mmap_path) // (4 * data["embedding_dim"])
                cache.memory_mapped_embeddings = np.memmap(mmap_path, dtype=np.float32, mode="r",
                                                           shape=(n, data["embedding_dim"]))
        else:
            cache = cls(n_gram_extractor=n_gram_extractor, embedding_dim=data["embedding_dim"],
                        use_memory_map=False, **device_kwargs)
            cache.embeddings = data["embeddings"]
        cache._dirty = True
        return cache
