"""Thin object wrapper over the C ABI: one :class:`SconeTable` = one ``scone_handle``
(device f-gram index + table shard).  Tensors cross the boundary as raw pointers
(``tensor.data_ptr()``); all work is enqueued on torch's current HIP stream."""

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L

_FMT = {"fp32": L.FMT_F32, "float32": L.FMT_F32, "f32": L.FMT_F32,
        "fp16": L.FMT_F16, "float16": L.FMT_F16, "f16": L.FMT_F16,
        "int8": L.FMT_I8, "i8": L.FMT_I8, "int4": L.FMT_I4, "i4": L.FMT_I4}
_PLACE = {"hbm": L.PLACE_HBM, "pinned_host": L.PLACE_PINNED_HOST}
_REDUCE = {"mean": L.REDUCE_MEAN, "sum": L.REDUCE_SUM}
_MODE = {"cover": L.MODE_COVER, "longest_suffix": L.MODE_LONGEST_SUFFIX}
_DT = {torch.float32: L.DT_F32, torch.float16: L.DT_F16, torch.bfloat16: L.DT_BF16}

I4_GROUP = 128


def format_code(fmt) -> int:
    if isinstance(fmt, int):
        return fmt
    try:
        return _FMT[str(fmt).lower()]
    except KeyError:
        raise ValueError(f"unknown table format {fmt!r} (fp32, fp16, int8, int4)") from None


def row_bytes(fmt: int, d: int) -> int:
    """Algorithmic bytes per table row."""
    return {L.FMT_F32: 4 * d, L.FMT_F16: 2 * d, L.FMT_I8: d + 2,
            L.FMT_I4: d // 2 + 2 * (d // I4_GROUP)}[fmt]


class SconeError(RuntimeError):
    pass


def _raise(code: int, msg: str):
    text = f"{msg} [{L.lib().scone_strerror(code).decode()}]"
    if code == L.EINVAL:
        raise ValueError(text)
    if code == L.ERANGE:
        raise IndexError(text)
    if code == L.ENOMEM:
        raise MemoryError(text)
    raise SconeError(text)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("scone_amd: the f-gram lookup path needs an MI355X (no HIP device visible); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def fit_gpu(tokens: torch.Tensor, text_offsets: torch.Tensor, max_n: int, min_freq: int, max_f_grams: int,
            device: Optional[torch.device] = None):
    """``scone_fit``: f-gram vocabulary of a tokenised corpus, built on the GPU.
    Returns ``(keys [S, max_n] uint32, lens [S] uint8, counts [S] uint32, n_distinct)`` as numpy arrays;
    row r is f-gram id r (the reference's ``Counter.most_common`` order)."""
    dev = torch.device(device) if device is not None else require_gpu()
    require_gpu()
    dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
    tok = tokens.to(device=dev, dtype=torch.int32).contiguous()
    off = text_offsets.to(device=dev, dtype=torch.int64).contiguous()
    n_tok, n_texts = tok.numel(), off.numel() - 1
    cap = int(min(max_f_grams, max(1, n_tok * max_n)))
    keys = torch.zeros((cap, max_n), dtype=torch.int32, device=dev)
    lens = torch.zeros(cap, dtype=torch.uint8, device=dev)
    counts = torch.zeros(cap, dtype=torch.int32, device=dev)
    n_out, n_distinct = C.c_uint64(0), C.c_uint64(0)
    with torch.cuda.device(dev):
        rc = L.lib().scone_fit(dev.index, _ptr(tok), n_tok, _ptr(off), n_texts, int(max_n), int(max(min_freq, 0)),
                               int(max_f_grams), _ptr(keys), _ptr(lens), _ptr(counts), cap, C.byref(n_out),
                               C.byref(n_distinct), _stream())
    if rc != L.OK:
        _raise(rc, "scone_fit failed (negative or too large token ids, or out of memory)")
    n = n_out.value
    return (keys[:n].cpu().numpy().view(np.uint32), lens[:n].cpu().numpy(), counts[:n].cpu().numpy().view(np.uint32),
            n_distinct.value)


class SconeTable:
    """Device-resident f-gram index (+ optional table shard)."""

    def __init__(self, max_n: int, n_rows: int, dim: int = 0, table_format="fp32", placement: str = "hbm",
                 device: Optional[torch.device] = None, row_begin: int = 0, row_end: Optional[int] = None,
                 index_capacity: int = 0, hot_rows: int = 0, lookup_mode: str = "cover",
                 stage_tokens: int = 0, cache_rows: int = 0) -> None:
        self._h = None
        lib = L.lib()
        dev = torch.device(device) if device is not None else require_gpu()
        if dev.type != "cuda":
            raise RuntimeError("scone_amd: SconeTable lives on a HIP device; got " + str(dev))
        require_gpu()
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.max_n, self.n_rows, self.dim = int(max_n), int(n_rows), int(dim)
        self.fmt = format_code(table_format)
        self.row_begin = int(row_begin)
        self.row_end = int(n_rows if row_end is None else row_end)
        cfg = L.SconeCfg(C.sizeof(L.SconeCfg), self.device.index, self.max_n, self.dim, self.fmt,
                         _PLACE[placement], self.n_rows, self.row_begin, self.row_end, int(index_capacity),
                         int(hot_rows), _MODE[lookup_mode], int(stage_tokens), int(cache_rows))
        h = C.c_void_p()
        rc = lib.scone_create(C.byref(cfg), C.byref(h))
        if rc != L.OK:
            _raise(rc, "scone_create: " + lib.scone_last_error(None).decode())
        self._h = h
        self._embed_fn = lib.scone_embed

    # -- plumbing ---------------------------------------------------------------
    def _check(self, rc: int, who: str) -> None:
        if rc != L.OK:
            _raise(rc, f"{who}: " + L.lib().scone_last_error(self._h).decode())

    def close(self) -> None:
        if self._h is not None:
            L.lib().scone_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self) -> int:
        bits = C.c_uint32(0)
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_status(self._h, C.byref(bits), _stream()), "scone_status")
        return bits.value

    def stage_counters(self) -> dict:
        """The cache of cold rows of a pinned-host table with ``stage_tokens > 0``: ``cache_rows`` (slots), ``rows_copied``
        (host -> HBM since the cache was created: every miss crosses PCIe once), ``chunks``, ``chunk_tokens`` (synchronises)."""
        v = [C.c_uint64(0) for _ in range(4)]
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_stage_counters(self._h, *[C.byref(x) for x in v]), "scone_stage_counters")
        return dict(zip(("cache_rows", "rows_copied", "chunks", "chunk_tokens"), (x.value for x in v)))

    # -- index ------------------------------------------------------------------
    def index_build(self, keys: np.ndarray, lens: np.ndarray, id0: int = 0) -> None:
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        n = int(lens.shape[0])
        if n == 0:
            return
        if keys.shape != (n, self.max_n):
            raise ValueError(f"keys must be [{n}, {self.max_n}] uint32, got {keys.shape}")
        rc = L.lib().scone_index_build(self._h, keys.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                       n, int(id0))
        self._check(rc, "scone_index_build")

    def index_build_device(self, keys: torch.Tensor, lens: torch.Tensor, id0: int = 0) -> None:
        assert keys.is_cuda and lens.is_cuda and keys.is_contiguous() and lens.is_contiguous()
        assert keys.dtype in (torch.int32, torch.uint32) and lens.dtype == torch.uint8
        n = int(lens.shape[0])
        with torch.cuda.device(self.device):
            rc = L.lib().scone_index_build_device(self._h, _ptr(keys), _ptr(lens), n, int(id0), _stream())
        self._check(rc, "scone_index_build_device")

    def index_stats(self) -> Tuple[int, int, int]:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(L.lib().scone_index_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "scone_index_stats")
        return a.value, b.value, c.value

    def index_blob_sizes(self) -> Tuple[int, int, int]:
        """Bytes of the three blobs of the built index: hash slots, unigram table, presence bitmap."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(L.lib().scone_index_blob_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)), "scone_index_blob_sizes")
        return a.value, b.value, c.value

    def index_export(self, out=None):
        """The built index as host arrays ``(slots uint8, uni int32, bloom uint8, n_keys, capacity)``.  ``out = (slots, uni,
        bloom)``: write into these arrays (e.g. sections of a memory-mapped file) instead of allocating."""
        a, b, c = (C.c_uint64(x) for x in self.index_blob_sizes())
        if out is None:
            slots = np.empty(a.value, dtype=np.uint8)
            uni = np.empty(b.value // 4, dtype=np.int32)
            bloom = np.empty(c.value, dtype=np.uint8)
        else:
            slots, uni, bloom = out
            assert (slots.nbytes, uni.nbytes, bloom.nbytes) == (a.value, b.value, c.value)
            assert all(x.flags["C_CONTIGUOUS"] and x.flags["WRITEABLE"] for x in out)
        n = C.c_uint64(0)
        rc = L.lib().scone_index_export(self._h, slots.ctypes.data_as(C.c_void_p), uni.ctypes.data_as(C.c_void_p),
                                        bloom.ctypes.data_as(C.c_void_p), C.byref(n))
        self._check(rc, "scone_index_export")
        return slots, uni, bloom, n.value, a.value // 16

    def index_import(self, slots: np.ndarray, uni: np.ndarray, bloom: np.ndarray, n_keys: int) -> None:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(L.lib().scone_index_blob_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)), "scone_index_blob_sizes")
        slots, uni, bloom = (np.ascontiguousarray(x) for x in (slots, uni, bloom))
        if (slots.nbytes, uni.nbytes, bloom.nbytes) != (a.value, b.value, c.value):
            raise ValueError("index blobs do not fit this handle (create it with the same max_n and index_capacity)")
        rc = L.lib().scone_index_import(self._h, slots.ctypes.data_as(C.c_void_p), uni.ctypes.data_as(C.c_void_p),
                                        bloom.ctypes.data_as(C.c_void_p), int(n_keys))
        self._check(rc, "scone_index_import")

    # -- table ------------------------------------------------------------------
    def upload(self, rows, scales=None, row0: int = 0) -> None:
        """Raw rows already in the table format (numpy host arrays or device tensors)."""
        is_dev = isinstance(rows, torch.Tensor)
        if is_dev:
            assert rows.is_cuda and rows.is_contiguous()
            nrows, rp = rows.shape[0], _ptr(rows)
            sp = _ptr(scales) if scales is not None else None
        else:
            rows = np.ascontiguousarray(rows)
            nrows, rp = rows.shape[0], rows.ctypes.data_as(C.c_void_p)
            if scales is not None:
                scales = np.ascontiguousarray(scales, dtype=np.float16)
            sp = scales.ctypes.data_as(C.c_void_p) if scales is not None else None
        with torch.cuda.device(self.device):
            rc = L.lib().scone_table_upload(self._h, rp, sp, int(row0), int(nrows), int(is_dev), _stream())
        self._check(rc, "scone_table_upload")

    def payload_bytes(self) -> int:
        return {L.FMT_F32: 4 * self.dim, L.FMT_F16: 2 * self.dim, L.FMT_I8: self.dim, L.FMT_I4: self.dim // 2}[self.fmt]

    def scales_per_row(self) -> int:
        return {L.FMT_F32: 0, L.FMT_F16: 0, L.FMT_I8: 1, L.FMT_I4: self.dim // I4_GROUP}[self.fmt]

    def download(self, row0: int, nrows: int, rows: Optional[np.ndarray] = None,
                 scales: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Raw payload rows ``uint8 [nrows, payload_bytes]`` and fp16 scales of global rows ``row0 ..`` -- into ``rows`` /
        ``scales`` when given (contiguous host arrays of those shapes, e.g. slices of a memory-mapped file)."""
        spr = self.scales_per_row()
        if rows is None:
            rows = np.empty((nrows, self.payload_bytes()), dtype=np.uint8)
        if scales is None and spr:
            scales = np.empty((nrows, spr), dtype=np.float16)
        assert rows.shape == (nrows, self.payload_bytes()) and rows.dtype == np.uint8 and rows.flags["C_CONTIGUOUS"]
        assert not spr or (scales.shape == (nrows, spr) and scales.dtype == np.float16 and scales.flags["C_CONTIGUOUS"])
        if not spr:
            scales = None
        with torch.cuda.device(self.device):
            rc = L.lib().scone_table_download(self._h, rows.ctypes.data_as(C.c_void_p),
                                              scales.ctypes.data_as(C.c_void_p) if spr else None, int(row0), int(nrows),
                                              0, _stream())
        self._check(rc, "scone_table_download")
        return rows, scales

    def store_f32(self, rows: torch.Tensor, row0: int = 0, ids: Optional[torch.Tensor] = None) -> None:
        rows = rows.to(device=self.device, dtype=torch.float32).contiguous()
        if rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {tuple(rows.shape)}")
        with torch.cuda.device(self.device):
            if ids is None:
                rc = L.lib().scone_table_store_f32(self._h, _ptr(rows), int(row0), rows.shape[0], _stream())
            else:
                ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
                rc = L.lib().scone_table_store_f32_ids(self._h, _ptr(rows), _ptr(ids), rows.shape[0], _stream())
            # `rows`/`ids` may be temporaries: keep them alive until the kernel has read them
            torch.cuda.current_stream().synchronize()
        self._check(rc, "scone_table_store_f32")

    def fill_synthetic(self, seed: int, base_scale: float) -> None:
        with torch.cuda.device(self.device):
            rc = L.lib().scone_table_fill_synthetic(self._h, int(seed) & 0xFFFFFFFF, float(base_scale), _stream())
        self._check(rc, "scone_table_fill_synthetic")

    def gather_rows(self, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty((ids.numel(), self.dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_table_gather_rows(self._h, _ptr(ids), ids.numel(), _ptr(out), _stream())
        self._check(rc, "scone_table_gather_rows")
        return out

    # -- hot path ---------------------------------------------------------------
    def _tok(self, tok: torch.Tensor) -> torch.Tensor:
        if tok.dim() == 1:
            tok = tok.unsqueeze(0)
        if tok.dim() != 2:
            raise ValueError("token ids must be [T] or [B, T]")
        return tok.to(device=self.device, dtype=torch.int32).contiguous()

    def match(self, tok: torch.Tensor) -> torch.Tensor:
        """hits[max_n, B, T] int32: id of the window tok[b, i:i+n] or -1."""
        tok = self._tok(tok)
        B, T = tok.shape
        hits = torch.empty((self.max_n, B, T), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_match(self._h, _ptr(tok), B, T, _ptr(hits), _stream())
        self._check(rc, "scone_match")
        return hits

    def match_csr(self, tok: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-position id lists (reference order, duplicates kept) as CSR (offsets[B*T+1], ids)."""
        tok = self._tok(tok)
        B, T = tok.shape
        ncand = self.max_n * (self.max_n + 1) // 2
        offsets = torch.empty(B * T + 1, dtype=torch.int32, device=self.device)
        ids = torch.empty(max(B * T * ncand, 1), dtype=torch.int32, device=self.device)
        total = C.c_int64(0)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_match_csr(self._h, _ptr(tok), B, T, _ptr(offsets), _ptr(ids), ids.numel(),
                                         C.byref(total), _stream())
        self._check(rc, "scone_match_csr")
        return offsets, ids[:total.value]

    def gather_reduce(self, offsets: torch.Tensor, ids: torch.Tensor, reduce: str = "mean",
                      base: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        offsets = offsets.to(device=self.device, dtype=torch.int32).contiguous()
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        ntok = offsets.numel() - 1
        out = torch.empty((ntok, self.dim), dtype=out_dtype, device=self.device)
        if base is not None:
            base = base.to(device=self.device, dtype=out_dtype).contiguous()
            assert base.shape == out.shape
        with torch.cuda.device(self.device):
            rc = L.lib().scone_gather_reduce(self._h, _ptr(offsets), _ptr(ids), ntok, _ptr(base), _REDUCE[reduce],
                                             _ptr(out), _DT[out_dtype], _stream())
        self._check(rc, "scone_gather_reduce")
        return out

    def embed(self, tok: torch.Tensor, wte: Optional[torch.Tensor] = None, wpe: Optional[torch.Tensor] = None,
              position_ids: Optional[torch.Tensor] = None, reduce: str = "mean",
              out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused match + gather + dequantise + reduce (+ wte[tok] + wpe[pos]) -> [B, T, d]."""
        if not (tok.dim() == 2 and tok.dtype == torch.int32 and tok.is_cuda and tok.is_contiguous()):
            tok = self._tok(tok)          # decode-size calls are host-bound: skip conversions that are no-ops
        B, T = tok.shape
        if out_dtype is None:
            out_dtype = wte.dtype if wte is not None else (wpe.dtype if wpe is not None else torch.float32)
        for name, w in (("wte", wte), ("wpe", wpe)):
            if w is not None:
                if not (w.is_cuda and w.is_contiguous() and w.dtype == out_dtype and w.dim() == 2
                        and w.shape[1] == self.dim):
                    raise ValueError(f"{name} must be a contiguous [*, {self.dim}] {out_dtype} tensor on {self.device}")
        if position_ids is not None:
            position_ids = position_ids.to(device=self.device, dtype=torch.int32).expand(B, T).contiguous()
        if out is None:
            out = torch.empty((B, T, self.dim), dtype=out_dtype, device=self.device)
        else:
            assert out.is_cuda and out.is_contiguous() and out.dtype == out_dtype and out.numel() == B * T * self.dim
        # the library selects its device itself (hipSetDevice); torch's current stream of THAT device is the launch stream
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._embed_fn(self._h, tok.data_ptr(), B, T, None if wte is None else wte.data_ptr(),
                            0 if wte is None else wte.shape[0], None if wpe is None else wpe.data_ptr(),
                            0 if wpe is None else wpe.shape[0], None if position_ids is None else position_ids.data_ptr(),
                            _REDUCE[reduce], out.data_ptr(), _DT[out_dtype], stream)
        if rc != L.OK:
            self._check(rc, "scone_embed")
        return out

    def embed_prefetch(self, tok: torch.Tensor, tokens_ready: bool = False) -> None:
        """``scone_embed_prefetch``: start the pinned-host prefetch pipeline for ``tok`` (int32 ``[B, T]`` on the device: pass the
        very tensor the later :meth:`embed` gets) behind the current stream -- or, ``tokens_ready=True``, right away (the tokens
        are complete; the prefetch then runs beside a lookup queued just before).  A no-op for other tables."""
        assert tok.dim() == 2 and tok.dtype == torch.int32 and tok.is_cuda and tok.is_contiguous()
        B, T = tok.shape
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(L.lib().scone_embed_prefetch(self._h, tok.data_ptr(), B, T, int(bool(tokens_ready)), stream), "scone_embed_prefetch")
        self._prefetch_keepalive = tok

    def reserve(self, max_tokens: int) -> None:
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_reserve(self._h, int(max_tokens)), "scone_reserve")

    def set_cu_reserve(self, n_reserved: int) -> None:
        """Leave ``n_reserved`` compute units (a multiple of 8; 0 = off) free of this handle's large-batch lookup kernels, for
        the transport kernels of other streams (``scone_set_cu_reserve``)."""
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_set_cu_reserve(self._h, int(n_reserved)), "scone_set_cu_reserve")

    def lookup_stream(self) -> Optional["torch.cuda.Stream"]:
        """The handle's CU-masked stream as a torch stream (None without a reserve): a loop queued on it -- ``with
        torch.cuda.stream(table.lookup_stream()):`` -- has its lookups launched there directly, without the two cross-stream
        events of the transparent form.  The handle uses it until the next :meth:`set_cu_reserve`; the stream object itself is
        never destroyed by the library (PyTorch's allocator keeps the streams a tensor was recorded on), only retired and re-used."""
        p = C.c_void_p()
        self._check(L.lib().scone_lookup_stream(self._h, C.byref(p)), "scone_lookup_stream")
        return torch.cuda.ExternalStream(p.value, device=self.device) if p.value else None

    def streams_overlap(self, a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
        """Do kernels queued on ``b`` start while a kernel queued before them on ``a`` is still running (i.e. do the two streams
        sit on different hardware queues)?  Synchronises both streams (``scone_streams_overlap``)."""
        v = C.c_int32(0)
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_streams_overlap(self._h, a.cuda_stream, b.cuda_stream, C.byref(v)), "scone_streams_overlap")
        return bool(v.value)

    def pick_side_stream(self, candidates: int = 6) -> "torch.cuda.Stream":
        """A new stream whose kernels run BESIDE those of the current stream: the first of ``candidates`` fresh streams that
        passes :meth:`streams_overlap` against the current stream (HIP spreads streams over four hardware queues; a side
        stream that shares the current stream's queue overlaps with nothing).  The first candidate if none passes."""
        cur = torch.cuda.current_stream(self.device)
        made = [torch.cuda.Stream(device=self.device) for _ in range(max(1, candidates))]
        for s in made:
            if self.streams_overlap(cur, s):
                return s
        return made[0]

    def cu_reserve(self) -> Tuple[int, int]:
        """(compute units reserved, compute units of the device)."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(L.lib().scone_get_cu_reserve(self._h, C.byref(a), C.byref(b)), "scone_get_cu_reserve")
        return a.value, b.value

    def profile_enable(self, enable: bool = True) -> None:
        self._check(L.lib().scone_profile_enable(self._h, int(enable)), "scone_profile_enable")

    def profile_read(self, reset: bool = True) -> Tuple[int, float]:
        """(launches, total milliseconds) of the gather/reduce kernel since the last reset."""
        n, ms = C.c_uint64(0), C.c_double(0.0)
        self._check(L.lib().scone_profile_read(self._h, C.byref(n), C.byref(ms), int(reset)), "scone_profile_read")
        return n.value, ms.value

    def profile_samples(self) -> np.ndarray:
        """Milliseconds of every timed launch since the last reset (launch order)."""
        n = C.c_uint64(0)
        self._check(L.lib().scone_profile_samples(self._h, None, 0, C.byref(n)), "scone_profile_samples")
        out = np.zeros(n.value, dtype=np.float32)
        if n.value:
            self._check(L.lib().scone_profile_samples(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n.value,
                                                      C.byref(n)), "scone_profile_samples")
        return out[:n.value]

    def embed_partial(self, tok: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        tok = self._tok(tok)
        B, T = tok.shape
        partial = torch.empty((B * T, self.dim), dtype=torch.float32, device=self.device)
        counts = torch.empty(B * T, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_embed_partial(self._h, _ptr(tok), B, T, _ptr(partial), _ptr(counts), _stream())
        self._check(rc, "scone_embed_partial")
        return partial, counts

    # -- row exchange between shards (scone_shard_*) ---------------------------------
    def shard_set_head(self, n_head: int) -> None:
        """Keep global rows ``[0, n_head)`` on this shard as well (replicated head: never sent between shards)."""
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_shard_set_head(self._h, int(n_head)), "scone_shard_set_head")
        self.n_head = min(int(n_head), self.n_rows)

    def shard_head_store_f32(self, rows: torch.Tensor, row0: int = 0) -> None:
        rows = rows.to(device=self.device, dtype=torch.float32).contiguous()
        if rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {tuple(rows.shape)}")
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_head_store_f32(self._h, _ptr(rows), int(row0), rows.shape[0], _stream())
            torch.cuda.current_stream().synchronize()          # `rows` may be a temporary
        self._check(rc, "scone_shard_head_store_f32")

    def shard_record_bytes(self) -> int:
        n = C.c_uint64(0)
        self._check(L.lib().scone_shard_record_bytes(self._h, C.byref(n)), "scone_shard_record_bytes")
        return n.value

    # -- all-gather form: one record per distinct row (scone_shard_gather_*) -------------
    def shard_gather_plan(self, tok: torch.Tensor) -> int:
        """Number of records this shard contributes: the distinct rows it owns (outside the replicated head) that
        the batch references (synchronises)."""
        tok = self._tok(tok)
        B, T = tok.shape
        n = C.c_uint64(0)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_plan(self._h, _ptr(tok), B, T, C.byref(n), _stream())
        self._check(rc, "scone_shard_gather_plan")
        self._shard_keepalive = (tok,)
        return n.value

    def shard_gather_pack(self, n_records: int) -> torch.Tensor:
        """uint8 ``[n_records, record_bytes]``: ``[row payload | scales | row id]`` per claimed row."""
        buf = torch.empty((int(n_records), self.shard_record_bytes()), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_pack(self._h, _ptr(buf), _stream())
        self._check(rc, "scone_shard_gather_pack")
        return buf

    def shard_select_slot(self, slot: int) -> None:
        """Plan slot 0 .. 3 for the scone_shard_gather_* calls that follow (host-side switch): the receiver-side state of a
        planned batch exists once per slot, so batches b + 1, b + 2 can be planned and exchanged while batch b is being reduced."""
        self._check(L.lib().scone_shard_select_slot(self._h, int(slot)), "scone_shard_select_slot")

    def shard_gather_plan_chunks(self, tok: torch.Tensor, n_chunks: int, dedup_across_chunks: bool = True) -> list:
        """Chunked plan: ``ends[c]`` = records this shard contributes to chunks ``0..c`` of the batch (chunk c = sequences
        ``[c * ceil(B / n_chunks), ...)``).  ``dedup_across_chunks=True`` (all-gather form): a row claimed by an earlier chunk
        is not claimed again; ``False`` (slice exchange, ``n_chunks = world``): every chunk lists each distinct row it
        references (synchronises)."""
        tok = self._tok(tok)
        B, T = tok.shape
        ends = (C.c_uint64 * 64)()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_plan_chunks(self._h, _ptr(tok), B, T, int(n_chunks), int(bool(dedup_across_chunks)),
                                                        ends, _stream())
        self._check(rc, "scone_shard_gather_plan_chunks")
        self._shard_keepalive = (tok,)
        return [int(e) for e in ends[:n_chunks]]

    def ell_width(self) -> int:
        """int32 words of one token's list record (8 for max_n <= 3, 16 for max_n = 4)."""
        n = C.c_uint32(0)
        self._check(L.lib().scone_ell_width(self._h, C.byref(n)), "scone_ell_width")
        return n.value

    def shard_gather_match(self, tok: torch.Tensor, seq_begin: int, seq_end: int, out_ell: torch.Tensor) -> None:
        """List records of sequences ``[seq_begin, seq_end)`` of the batch (matched against ALL rows) into
        ``out_ell[:(seq_end - seq_begin) * T]`` (int32 ``[., ell_width()]``): this rank's share of a plan whose match is
        sharded over the ranks."""
        tok = self._tok(tok)
        B, T = tok.shape
        n = (seq_end - seq_begin) * T
        assert out_ell.is_cuda and out_ell.is_contiguous() and out_ell.dtype == torch.int32 and out_ell.numel() >= n * self.ell_width()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_match(self._h, _ptr(tok), B, T, int(seq_begin), int(seq_end), _ptr(out_ell), _stream())
        self._check(rc, "scone_shard_gather_match")
        self._shard_keepalive = (tok, out_ell)

    def shard_gather_plan_ell(self, ell: torch.Tensor, B: int, T: int, n_chunks: int, dedup_across_chunks: bool = True) -> list:
        """The claim passes of :meth:`shard_gather_plan_chunks` over list records the caller gathered (``ell`` int32
        ``[>= B * T, ell_width()]``, token order); ``ell`` is borrowed by the plan slot until the batch has been reduced
        (synchronises)."""
        assert ell.is_cuda and ell.is_contiguous() and ell.dtype == torch.int32 and ell.numel() >= B * T * self.ell_width()
        ends = (C.c_uint64 * 64)()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_plan_ell(self._h, _ptr(ell), int(B), int(T), int(n_chunks),
                                                     int(bool(dedup_across_chunks)), ends, _stream())
        self._check(rc, "scone_shard_gather_plan_ell")
        self._shard_keepalive = (ell,)
        return [int(e) for e in ends[:n_chunks]]

    def shard_gather_pack_range(self, first: int, count: int, out: torch.Tensor) -> None:
        """Records ``[first, first + count)`` of the plan into ``out[:count]``; the rest of ``out`` becomes padding."""
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.uint8 and out.shape[1] == self.shard_record_bytes()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_pack_range(self._h, int(first), int(count), int(out.shape[0] - count), _ptr(out),
                                                       _stream())
        self._check(rc, "scone_shard_gather_pack_range")

    def shard_gather_add_records(self, records: torch.Tensor, record0: int, n_records: int) -> None:
        """Receiver: records ``[record0, record0 + n_records)`` of the gathered buffer ``records [n_total, record_bytes]``
        join the row map (``record0 == 0`` starts a new exchange)."""
        assert records.is_cuda and records.is_contiguous() and records.dtype == torch.uint8
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_add_records(self._h, _ptr(records), int(record0), int(n_records),
                                                        records.shape[0], _stream())
        self._check(rc, "scone_shard_gather_add_records")

    def shard_gather_remap_range(self, seq_begin: int, seq_end: int) -> None:
        """Rewrite the lists of sequences ``[seq_begin, seq_end)`` of the planned batch to record numbers on the current
        stream (their rows must have been added); :meth:`shard_gather_embed_range` then only launches the lookup."""
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_remap_range(self._h, int(seq_begin), int(seq_end), _stream())
        self._check(rc, "scone_shard_gather_remap_range")

    def shard_gather_embed_range(self, tok: torch.Tensor, seq_begin: int, seq_end: int, records: torch.Tensor,
                                 out: torch.Tensor, wte: Optional[torch.Tensor] = None, wpe: Optional[torch.Tensor] = None,
                                 position_ids: Optional[torch.Tensor] = None, reduce: str = "mean",
                                 out_is_slice: bool = False) -> None:
        """Sequences ``[seq_begin, seq_end)`` of the planned batch into their place in ``out [B*T, d]`` -- or, with
        ``out_is_slice``, into ``out [>= (seq_end - seq_begin) * T, d]`` whose first row is sequence ``seq_begin``."""
        tok = self._tok(tok)
        B, T = tok.shape
        n_out = (seq_end - seq_begin) * T if out_is_slice else B * T
        assert out.is_cuda and out.is_contiguous() and out.numel() >= n_out * self.dim
        if position_ids is not None:                      # (a host / int64 tensor handed to the kernel as it is would be read as garbage)
            position_ids = position_ids.to(device=self.device, dtype=torch.int32).expand(B, T).contiguous()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_embed_range(self._h, _ptr(tok), B, T, int(seq_begin), int(seq_end), _ptr(records),
                                                        records.shape[0], _ptr(wte), 0 if wte is None else wte.shape[0],
                                                        _ptr(wpe), 0 if wpe is None else wpe.shape[0], _ptr(position_ids),
                                                        _REDUCE[reduce], _ptr(out), int(seq_begin) * T if out_is_slice else 0,
                                                        _DT[out.dtype], _stream())
        self._shard_keepalive = (records, tok, position_ids, wte, wpe, out)
        self._check(rc, "scone_shard_gather_embed_range")

    # -- all-gather form with columns on the wire (scone_shard_cols_*) ---------------------------------
    @staticmethod
    def cols_frag_slots(count: int) -> int:
        """u64 slots of the sender's hash fragment for ``count`` rows (both ends derive it from the exchanged counts)."""
        n = C.c_uint64(0)
        L.lib().scone_shard_cols_frag_slots(int(count), C.byref(n))
        return n.value

    def shard_cols_pack(self, first: int, count: int, rows_out: torch.Tensor, scales_out: Optional[torch.Tensor],
                        frag_out: torch.Tensor) -> None:
        """Records ``[first, first + count)`` of the plan as columns: payload rows ``uint8 [count, payload_bytes]``, scales
        ``uint8 [count, scale bytes]`` (None for fp32 / fp16 tables) and the hash fragment ``int64 [slots]`` (cleared and
        filled: row id -> position)."""
        assert rows_out.is_cuda and rows_out.is_contiguous() and rows_out.numel() >= count * self.payload_bytes()
        assert frag_out.is_cuda and frag_out.is_contiguous() and frag_out.dtype == torch.int64
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_cols_pack(self._h, int(first), int(count), _ptr(rows_out), _ptr(scales_out), _ptr(frag_out),
                                               frag_out.numel(), _stream())
        self._check(rc, "scone_shard_cols_pack")

    def shard_gather_plan_async(self, tok: torch.Tensor) -> None:
        """The one-chunk plan without the host round trip: match + claim pass are enqueued, the count stays on the device
        (pack with :meth:`shard_cols_pack_cap`)."""
        tok = self._tok(tok)
        B, T = tok.shape
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_plan_async(self._h, _ptr(tok), B, T, _stream())
        self._check(rc, "scone_shard_gather_plan_async")
        self._shard_keepalive = (tok,)

    def shard_gather_plan_ell_async(self, ell: torch.Tensor, B: int, T: int) -> None:
        """:meth:`shard_gather_plan_ell` (one chunk) without the host round trip."""
        assert ell.is_cuda and ell.is_contiguous() and ell.dtype == torch.int32 and ell.numel() >= B * T * self.ell_width()
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_plan_ell_async(self._h, _ptr(ell), int(B), int(T), _stream())
        self._check(rc, "scone_shard_gather_plan_ell_async")
        self._shard_keepalive = (ell,)

    def shard_cols_pack_cap(self, cap_rows: int, rows_out: torch.Tensor, scales_out: Optional[torch.Tensor], frag_out: torch.Tensor,
                            header_out: torch.Tensor) -> None:
        """The pack of a sync-free plan: up to ``cap_rows`` claimed rows as columns, the count read on the device;
        ``header_out`` (int64 ``[2]``, device) = (rows claimed, 1 if more than ``cap_rows``)."""
        assert rows_out.is_cuda and rows_out.is_contiguous() and rows_out.numel() >= cap_rows * self.payload_bytes()
        assert frag_out.is_cuda and frag_out.is_contiguous() and frag_out.dtype == torch.int64
        assert header_out.is_cuda and header_out.is_contiguous() and header_out.dtype == torch.int64 and header_out.numel() >= 2
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_cols_pack_cap(self._h, int(cap_rows), _ptr(rows_out), _ptr(scales_out), _ptr(frag_out),
                                                   frag_out.numel(), _ptr(header_out), _stream())
        self._check(rc, "scone_shard_cols_pack_cap")

    def shard_cols_build_frag(self, ids: torch.Tensor, frag_out: torch.Tensor) -> None:
        """The fragment of an arbitrary id list (position = index in the list)."""
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        assert frag_out.is_cuda and frag_out.is_contiguous() and frag_out.dtype == torch.int64
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_cols_build_frag(self._h, _ptr(ids), ids.numel(), _ptr(frag_out), frag_out.numel(), _stream())
            torch.cuda.current_stream().synchronize()              # `ids` may be a temporary
        self._check(rc, "scone_shard_cols_build_frag")

    def scale_bytes(self) -> int:
        return 2 * self.scales_per_row()

    def shard_head_version(self) -> int:
        """Counter bumped by every change of the replicated head (buffers that start with the head's scales are refilled)."""
        v = C.c_uint64(0)
        self._check(L.lib().scone_shard_head_version(self._h, C.byref(v)), "scone_shard_head_version")
        return v.value

    def shard_head_scales_into(self, scales_full: torch.Tensor) -> None:
        """The replicated head's scales into the front of a ``[n_head + capacity, scale bytes]`` buffer."""
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_head_scales(self._h, _ptr(scales_full), _stream())
        self._check(rc, "scone_shard_head_scales")

    def shard_cols_embed(self, tok: torch.Tensor, seq_begin: int, seq_end: int, rows: torch.Tensor, n_total: int,
                         scales_full: Optional[torch.Tensor], frags: torch.Tensor, frag_off, frag_slots, rec_base,
                         out: torch.Tensor, wte: Optional[torch.Tensor] = None, wpe: Optional[torch.Tensor] = None,
                         position_ids: Optional[torch.Tensor] = None, reduce: str = "mean", row_lo=None) -> None:
        """Sequences ``[seq_begin, seq_end)`` of the planned batch into their place in ``out [B*T, d]`` out of ``[replicated
        head | rows]`` (payload stride), lists resolved through the owners' fragments.  ``row_lo``: the owners' row ranges,
        ``len(frag_off) + 1`` ascending values from 0 to ``n_rows`` (None: the floor partition of ``distributed.shard_range``)."""
        tok = self._tok(tok)
        B, T = tok.shape
        W = len(frag_off)
        assert out.is_cuda and out.is_contiguous() and out.numel() >= B * T * self.dim
        if position_ids is not None:
            position_ids = position_ids.to(device=self.device, dtype=torch.int32).expand(B, T).contiguous()
        arr = lambda v: (C.c_uint64 * 65)(*[int(x) for x in v])
        if row_lo is not None and len(row_lo) != W + 1:
            raise ValueError("row_lo needs len(frag_off) + 1 entries")
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_cols_embed(self._h, _ptr(tok), B, T, int(seq_begin), int(seq_end), _ptr(rows), int(n_total),
                                                _ptr(scales_full), _ptr(frags), frags.numel(), arr(frag_off), arr(frag_slots),
                                                arr(rec_base), None if row_lo is None else arr(row_lo), W,
                                                _ptr(wte), 0 if wte is None else wte.shape[0], _ptr(wpe),
                                                0 if wpe is None else wpe.shape[0], _ptr(position_ids), _REDUCE[reduce], _ptr(out),
                                                0, _DT[out.dtype], _stream())
        self._shard_keepalive = (rows, scales_full, frags, tok, position_ids, wte, wpe, out)
        self._check(rc, "scone_shard_cols_embed")

    # -- peer-mapped buffers, interprocess events, copy-engine pushes (scone_ipc_*: the "sdma" transport) ------------
    def ipc_alloc(self, nbytes: int) -> Tuple[int, bytes]:
        """Device memory of this handle's device + its 64-byte interprocess handle: ``(pointer, handle)``."""
        p, hb = C.c_void_p(), C.create_string_buffer(64)
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_ipc_alloc(self._h, int(nbytes), C.byref(p), hb), "scone_ipc_alloc")
        return p.value, hb.raw

    def ipc_free(self, ptr: int) -> None:
        self._check(L.lib().scone_ipc_free(self._h, C.c_void_p(ptr)), "scone_ipc_free")

    def ipc_open(self, handle: bytes) -> int:
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_ipc_open(self._h, C.c_char_p(handle), C.byref(p)), "scone_ipc_open")
        return p.value

    def ipc_close(self, ptr: int) -> None:
        self._check(L.lib().scone_ipc_close(self._h, C.c_void_p(ptr)), "scone_ipc_close")

    def ipc_event_create(self) -> Tuple[int, bytes]:
        e, hb = C.c_void_p(), C.create_string_buffer(64)
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_ipc_event_create(self._h, C.byref(e), hb), "scone_ipc_event_create")
        return e.value, hb.raw

    def ipc_event_open(self, handle: bytes) -> int:
        e = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(L.lib().scone_ipc_event_open(self._h, C.c_char_p(handle), C.byref(e)), "scone_ipc_event_open")
        return e.value

    def ipc_event_destroy(self, event: int) -> None:
        self._check(L.lib().scone_ipc_event_destroy(self._h, C.c_void_p(event)), "scone_ipc_event_destroy")

    # (the stream is torch's current stream of THE TABLE'S device, whatever device is current in the caller)
    def ipc_event_record(self, event: int) -> None:
        with torch.cuda.device(self.device):
            rc = L.lib().scone_ipc_event_record(self._h, C.c_void_p(event), _stream())
        self._check(rc, "scone_ipc_event_record")

    def ipc_event_wait(self, event: int) -> None:
        with torch.cuda.device(self.device):
            rc = L.lib().scone_ipc_event_wait(self._h, C.c_void_p(event), _stream())
        self._check(rc, "scone_ipc_event_wait")

    def ipc_push(self, dst_ptr: int, src_ptr: int, nbytes: int, copy_engine: bool = True) -> None:
        """``nbytes`` from ``src_ptr`` (this device) to ``dst_ptr`` (possibly a peer's mapped buffer) on the current stream."""
        with torch.cuda.device(self.device):
            rc = L.lib().scone_ipc_push(self._h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), int(nbytes), int(bool(copy_engine)), _stream())
        self._check(rc, "scone_ipc_push")

    def ipc_tensor(self, ptr: int, nbytes: int) -> torch.Tensor:
        """A uint8 tensor over raw device memory of this handle's device (no ownership: keep the allocation alive)."""
        class _Raw:
            pass
        raw = _Raw()
        raw.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        with torch.cuda.device(self.device):
            return torch.as_tensor(raw, device=self.device)

    def shard_gather_embed(self, tok: torch.Tensor, records: torch.Tensor, wte: Optional[torch.Tensor] = None,
                           wpe: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                           reduce: str = "mean", out_dtype: torch.dtype = torch.float32,
                           out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The whole batch ``[B*T, d]`` out of ``[replicated head | records of every shard]``."""
        tok = self._tok(tok)
        B, T = tok.shape
        if position_ids is not None:
            position_ids = position_ids.to(device=self.device, dtype=torch.int32).expand(B, T).contiguous()
        if out is None:
            out = torch.empty((B * T, self.dim), dtype=out_dtype, device=self.device)
        assert records.is_cuda and records.is_contiguous() and records.dtype == torch.uint8
        with torch.cuda.device(self.device):
            rc = L.lib().scone_shard_gather_embed(self._h, _ptr(tok), B, T, _ptr(records), records.shape[0], _ptr(wte),
                                                  0 if wte is None else wte.shape[0], _ptr(wpe),
                                                  0 if wpe is None else wpe.shape[0], _ptr(position_ids), _REDUCE[reduce],
                                                  _ptr(out), _DT[out_dtype], _stream())
        self._shard_keepalive = (records, tok, position_ids, wte, wpe)       # read in place after the call returns
        self._check(rc, "scone_shard_gather_embed")
        return out

    def finalize(self, sums: torch.Tensor, counts: torch.Tensor, tok: torch.Tensor, tok_begin: int, tok_end: int,
                 wte: Optional[torch.Tensor] = None, wpe: Optional[torch.Tensor] = None,
                 position_ids: Optional[torch.Tensor] = None, reduce: str = "mean",
                 out_dtype: torch.dtype = torch.float32, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        tok = self._tok(tok)
        B, T = tok.shape
        n = tok_end - tok_begin
        sums = sums.to(device=self.device, dtype=torch.float32).contiguous()
        counts = counts.to(device=self.device, dtype=torch.int32).contiguous()
        assert sums.shape == (n, self.dim) and counts.shape == (n,)
        if position_ids is not None:
            position_ids = position_ids.to(device=self.device, dtype=torch.int32).expand(B, T).contiguous()
        if out is None:
            out = torch.empty((n, self.dim), dtype=out_dtype, device=self.device)
        else:
            assert out.is_cuda and out.is_contiguous() and out.dtype == out_dtype and out.shape == (n, self.dim)
        with torch.cuda.device(self.device):
            rc = L.lib().scone_finalize(self._h, _ptr(sums), _ptr(counts), _ptr(tok), B, T, int(tok_begin),
                                        int(tok_end), _ptr(wte), 0 if wte is None else wte.shape[0], _ptr(wpe),
                                        0 if wpe is None else wpe.shape[0], _ptr(position_ids), _REDUCE[reduce],
                                        _ptr(out), _DT[out_dtype], _stream())
        self._check(rc, "scone_finalize")
        return out
