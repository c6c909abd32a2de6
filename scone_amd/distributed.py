"""Row-sharded f-gram tables across GPUs (one process per GPU, ``torch.distributed``;
backend "nccl" is RCCL over xGMI on ROCm).

The reference keeps its table in one process (``embedding_cache.py:49-50``); sharding is new
here and only needed when the table does not fit one GPU's 288 GB (BASELINE config C5:
1e9 rows INT4 d=1024 = 528 GB).  Layout:

* rows: contiguous id ranges, rank r owns ``[r*N/W, (r+1)*N/W)``;
* index (f-gram -> id): replicated, so every rank matches the full batch locally and knows
  every token's full hit count K_t -- no id exchange;
* rank r finalises slice r of the batch (whole sequences);
* exchange ``"rows"`` (default for slices; one record per DISTINCT row and destination, built from the chunked-gather
  primitives -- ``_embed_row_exchange_dedup``; the first form, one record per reference, was removed in round 4): what
  crosses xGMI are the QUANTISED ROWS a slice needs from the other shards -- ``scone_shard_gather_plan_chunks`` with chunk q =
  slice q (both ends derive what is sent from the replicated tokens and index: no request round) ->
  ``scone_shard_gather_pack_range`` -> ONE ``all_to_all_single`` of records -> ``scone_shard_gather_add_records`` /
  ``_embed_range`` (the ordinary fused kernel reads the records in place).  An INT4 d=1024 row is 544 B on the wire where an
  fp32 partial sum is 4096 B per token AND rank: for the C5 workload about 25 MB per rank and step instead of 3.7 GB, spread
  over all 7 xGMI links by the all-to-all, and the result is bit-identical to the unsharded table (same reduction order).
  This is the form for DATA-PARALLEL consumers (``gather_output=False``): the only one whose per-rank work falls with the
  world size;
* ``replicated_rows=H``: the head of the table, global rows ``[0, H)``, is kept on every rank and never sent.
  f-gram ids are frequency-ordered (``Counter.most_common``), so the head holds every unigram and the most
  frequent f-grams -- about half of all row references on the C5-shaped workload, all of which would otherwise
  leave the one rank that owns them (measured on one GPU with 8 shards of a 100M-row table,
  ``tools/shard_emulate.py``: rank 0 sends 550 MB per 1M-token step, the others 52-75 MB);
* exchange ``"gather_rows"``: when EVERY rank needs the whole ``[B, T, d]`` output (the north-star's all-gather of the
  aggregated vectors), gathering the 2 KB fp16 vector of every token costs four times the bytes of gathering the
  quantised ROWS the batch references, and every DISTINCT row crosses once however many tokens reference it:
  ``scone_shard_gather_plan`` claims the distinct rows this rank owns that the batch references, ``scone_shard_gather_pack``
  writes one record ``[row | scales | row id]`` per claimed row, the records of all ranks are gathered -- as exact
  point-to-point ranges (``gather_transport="p2p"``: one ``batch_isend_irecv``, each xGMI link carries one peer's records) or
  through ``all_gather_into_tensor`` with every contribution padded to the largest (``"all_gather"``) -- and
  ``scone_shard_gather_embed`` indexes them by row id and reduces the whole batch out of
  ``[replicated head | gathered records]`` -- bit-identical again, ~0.25 GB into every rank per 1M-token step instead
  of ~1.9 GB;
* split-phase form of ``"gather_rows"`` for a serving loop: ``gather_rows_begin`` (plan, packs, transfers -- on a side stream,
  on another of the handle's plan slots) / ``gather_rows_finish`` (the reduction, on the caller's stream): two batches in
  flight (``plan_slots=3``: three), the next batch's exchange hidden behind the current batch's reduction;
* round 3, both on by default for batches of >= 65,536 tokens.  ``shard_match``: in the all-gather form every rank needs the id
  lists of the WHOLE batch, and matching all of it against a 1e9-key index was the largest helper kernel of the step; index and
  tokens are replicated and matching is per sequence, so rank r matches slice r only and the 32-B list records are
  all-gathered (``scone_shard_gather_match / _plan_ell``).  Columns on the wire (the one-piece form, ``gather_chunks=1``, the
  default since round 4; ``gather_chunks > 1`` is the legacy chunk pipeline with records on the wire): a contribution
  travels as payload rows | scales | the SENDER's hash fragment ``row id -> position`` instead of ``[payload | scales | row id]``
  records -- no receiver indexes anything (0.45M device-scope CAS per rank and step before), the rows land at the table's own
  stride (``scone_shard_cols_pack / _embed``);
* round 4, ``gather_transport="sdma"``: the exchange of the one-piece form sends exact contiguous ranges, so it needs no
  kernel at all -- every rank maps its peers' per-slot receive buffers once (HIP interprocess handles, ``scone_ipc_*``), packs its
  columns into its own range and PUSHES that range to the same offsets of the seven peer buffers with ``hipMemcpyAsync`` on one
  stream per peer (the copy engines move the bytes over xGMI while every wave slot of the chip belongs to the lookup kernel;
  RCCL's send / recv are kernels that must find room beside it), "my pushes for this slot are complete" / "I have reduced this
  slot" are interprocess events, and the two host-side rendezvous a step needs anyway -- the exchange of the counts and one
  barrier on a gloo group -- make sure a wait never sees the previous batch's record (the sync-free form has a second
  barrier where the count exchange was).  Interprocess events come from a per-slot pool (a HIP interprocess event survives 32
  records); a collective self-test (pattern push, event, verify) runs before first use.  Falls back to ``"p2p"`` (on every rank
  alike) when the handles cannot be created or opened or the self-test's bytes do not arrive; ``close()`` is collective;
* ``SCONE_DIST_TRACE=1``: one stderr line BEFORE every collective (name, counts, bytes) -- the last line of a hung job names
  the collective it hangs in;
* exchange ``"partial_sums"`` (kept for comparison): every rank sums the rows it owns
  (``scone_embed_partial``) -> ``reduce_scatter`` -> ``scone_finalize``;
* finally an ``all_gather`` of the finished vectors in the output dtype (skippable when the
  consumer is data-parallel over the same slices).

Sequences are independent, so when the table DOES fit one GPU the path needs no collective
at all: replicate the table and shard the tokens (bench.py's default at N > 1).
"""

import os
import sys
import time
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from scone_amd.tokenization.n_gram_extractor import NGramExtractor

_T_IMPORT = time.time()


def _trace(name: str, group=None, **facts) -> None:
    """``SCONE_DIST_TRACE=1``: one stderr line per collective -- its name, element counts and bytes -- written BEFORE the
    call, so that the last line of a hung job's stderr names the collective it hangs in (bench.py switches it on for
    rank 0 of an N > 1 run; the driver keeps the tail of stderr)."""
    if os.environ.get("SCONE_DIST_TRACE") != "1":
        return
    try:
        r = dist.get_rank(group)
    except Exception:
        r = -1
    sys.stderr.write(f"[scone dist +{time.time() - _T_IMPORT:8.3f}s r{r}] {name} "
                     + " ".join(f"{k}={v}" for k, v in facts.items()) + "\n")
    sys.stderr.flush()


def _host_staged(group) -> bool:
    """True when the group's backend cannot carry device tensors (gloo): the collectives below then go
    through host copies.  RCCL ("nccl") moves device buffers directly over xGMI; the staged form exists
    so that the whole sharded path -- real kernels, real process separation -- can be rehearsed with
    several ranks sharing ONE GPU, which RCCL refuses."""
    return dist.get_backend(group) == "gloo"


def _reduce_scatter_sum(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    _trace("reduce_scatter_tensor(sum)", group, dtype=inp.dtype, in_elems=inp.numel(), out_elems=out.numel(),
           in_bytes=inp.numel() * inp.element_size())
    if _host_staged(group) and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(o, inp.cpu(), op=dist.ReduceOp.SUM, group=group)
        out.copy_(o)
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    _trace("all_gather_into_tensor", group, dtype=inp.dtype, in_elems=inp.numel(), out_bytes=out.numel() * out.element_size())
    if _host_staged(group) and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


class _Done:
    """Stand-in for a completed collective (host-staged groups run synchronously)."""

    def wait(self) -> None:
        pass


def _all_gather_async(out: torch.Tensor, inp: torch.Tensor, group):
    """``all_gather_into_tensor`` that returns a handle with ``wait()``: over RCCL the collective runs on the
    communicator's stream behind everything already queued on the current stream, and the current stream only waits
    for it at ``wait()`` -- kernels enqueued in between overlap the transfer."""
    if _host_staged(group) and inp.is_cuda:
        _all_gather(out, inp, group)
        return _Done()
    _trace("all_gather_into_tensor(async)", group, dtype=inp.dtype, in_elems=inp.numel(), out_bytes=out.numel() * out.element_size())
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)


class _Works:
    """Several outstanding requests behind one ``wait()``."""

    def __init__(self, works) -> None:
        self.works = list(works)

    def wait(self) -> None:
        for w in self.works:
            w.wait()


class _AfterEvent:
    """``wait()``: the current stream waits for a (same-process) event."""

    def __init__(self, event) -> None:
        self.event = event

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


class _SdmaArrival:
    """``wait()``: the current stream waits until every peer's pushes into this slot are complete (their interprocess
    "sent" events; a host-side barrier after the records makes sure the wait sees THIS batch's record)."""

    def __init__(self, table, events) -> None:
        self.table, self.events = table, list(events)

    def wait(self) -> None:
        for e in self.events:
            self.table.ipc_event_wait(e)


def _exchange_exact_async(region: torch.Tensor, offs, counts, rank: int, group):
    """All-gather with UNEQUAL contributions, as point-to-point transfers: ``region [sum(counts), rec]`` holds rank r's
    records at ``[offs[r], offs[r] + counts[r])`` -- mine are already in place -- and every rank sends its own range to every
    peer and receives every peer's range, all in one batch (``batch_isend_irecv``: one RCCL group, every xGMI link carries
    exactly one peer's records in each direction).  ``all_gather_into_tensor`` needs equal contributions, i.e. padding to
    the largest; on the C5-shaped batch the largest is twice the mean (511 MB against 258 MB into every rank)."""
    W = len(counts)
    me = region[offs[rank]:offs[rank] + counts[rank]]
    _trace("batch_isend_irecv(exact ranges)", group, record_bytes=region.shape[1], counts=list(counts),
           send_bytes_per_peer=counts[rank] * region.shape[1], recv_bytes=(sum(counts) - counts[rank]) * region.shape[1])
    if _host_staged(group) and region.is_cuda:
        src = me.cpu()
        bufs = {r: torch.empty((counts[r], region.shape[1]), dtype=region.dtype) for r in range(W) if r != rank and counts[r]}
        ops = []
        for r in range(W):
            if r == rank:
                continue
            if counts[rank]:
                ops.append(dist.P2POp(dist.isend, src, dist.get_global_rank(group, r) if group is not None else r, group))
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, bufs[r], dist.get_global_rank(group, r) if group is not None else r, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for r, b in bufs.items():
            region[offs[r]:offs[r] + counts[r]].copy_(b)
        return _Done()
    ops = []
    for r in range(W):
        if r == rank:
            continue
        peer = dist.get_global_rank(group, r) if group is not None else r
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, me, peer, group))
        if counts[r]:
            ops.append(dist.P2POp(dist.irecv, region[offs[r]:offs[r] + counts[r]], peer, group))
    return _Works(dist.batch_isend_irecv(ops)) if ops else _Done()


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group) -> None:
    _trace("all_to_all_single", group, dtype=inp.dtype, row_bytes=inp.shape[1] * inp.element_size() if inp.dim() == 2 else inp.element_size(),
           in_splits=list(in_splits), out_splits=list(out_splits))
    if _host_staged(group) and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range owned by ``rank``: ``[rank*N//W, (rank+1)*N//W)``."""
    return (rank * n_rows) // world, ((rank + 1) * n_rows) // world


def owner_of(ids: torch.Tensor, n_rows: int, world: int) -> torch.Tensor:
    """Rank owning each id under :func:`shard_range` (inverse of the floor partition)."""
    return ((ids.to(torch.int64) + 1) * world - 1) // n_rows


class ShardedEmbeddingCache:
    """``EmbeddingCache.embed_tokens`` for a table whose rows are split over the ranks of ``group``.

    ``table`` is this rank's device handle (``hip_backend.SconeTable`` created with
    ``row_begin/row_end``); it is built here unless the caller supplies one (tests inject a
    stand-in to exercise the exchange logic over gloo on CPU).
    """

    def __init__(self, n_gram_extractor: NGramExtractor, embedding_dim: int, *, table_format: str = "int8",
                 rank: Optional[int] = None, world: Optional[int] = None, group=None, device=None,
                 n_rows: Optional[int] = None, placement: str = "hbm", table=None, replicated_rows: int = 0,
                 gather_chunks: int = 1, gather_transport: str = "p2p", shard_match="auto", plan_slots: int = 2,
                 sync_free_plan: bool = True, control_group=None) -> None:
        self.group = group
        # gather_transport="sdma" rendezvous on the host twice per step: a gloo group of the same ranks as `group`.  None: the
        # group itself when it is a host group, else one is created (possible only when `group` is the whole world)
        self.control_group = control_group
        # The one-piece "gather_rows" exchange without host round trips (round 4): after a first batch with exact sizes, every
        # later batch sizes its transfers from what the ranks contributed before (+ 12.5 %, never shrinking), plans and packs
        # with the count on the device (scone_shard_gather_plan*_async / scone_shard_cols_pack_cap), ships each rank's count in
        # two extra words behind its hash fragment, and reads the counts back when the batch is REDUCED -- gather_rows_begin no
        # longer blocks the host on the plan and on the exchange of the counts.  A batch whose contribution outgrew its
        # capacity is repeated with exact sizes inside gather_rows_finish (every rank sees the same counts: they agree).
        # Transports "p2p" / "all_gather"; the "sdma" transport keeps the exact form (its rendezvous are host-side anyway).
        self.sync_free_plan = bool(sync_free_plan)
        self._caps = None                              # rows per rank the next sync-free exchange provides for
        self.sync_free_stats = {"exchanges": 0, "overflow_repeats": 0}
        self._hdr_stream = None
        # split-phase "gather_rows": batches in flight (2 .. 4).  The chain plan -> count exchange -> pack -> transfers of
        # a batch must fit (plan_slots - 1) reductions: with 2 it has ONE reduction's time, with 3 it has two -- the
        # setting for links on which a step's 0.26 GB of records take about as long as its reduction
        if not 2 <= int(plan_slots) <= 4:
            raise ValueError("plan_slots must be 2, 3 or 4")
        self.plan_slots = int(plan_slots)
        # The match of a plan sharded over the ranks (rank r matches slice r, the 32-B list records are all-gathered) instead
        # of every rank matching the whole batch: True / False / "auto" (on for batches of >= 65,536 tokens with at least one
        # sequence per rank -- below that the extra small collective costs more than the match it saves)
        if shard_match not in (True, False, "auto"):
            raise ValueError("shard_match must be True, False or 'auto'")
        self.shard_match = shard_match
        # "gather_rows": ONE switch for the form of the exchange.  1 (default, and always in the split-phase loop's measured
        # configuration): one piece, COLUMNS on the wire -- payload rows | scales | the sender's hash fragment (row id ->
        # position), three ranges per peer, no indexing pass on the receiver.  > 1: the legacy chunk pipeline of rounds 2-3 --
        # the batch is exchanged and reduced in this many chunks of sequences, [payload | scales | row id] RECORDS on the wire,
        # indexed by every receiver, chunk c + 1's transfer behind chunk c's reduction inside one call
        self.gather_chunks = int(gather_chunks)
        if gather_transport not in ("p2p", "all_gather", "sdma"):
            raise ValueError("gather_transport must be 'p2p', 'all_gather' or 'sdma'")
        # "gather_rows": how the records travel -- exact point-to-point ranges (RCCL send / recv kernels), all_gather_into_tensor
        # padded to the largest, or -- the one-piece columns form only -- copy-engine pushes into peer-mapped buffers ("sdma";
        # chunked / record exchanges use "p2p" under it)
        self.gather_transport = gather_transport
        self.transport_fallback_reason = None         # why "sdma" was asked for and "p2p" is used (None: it was not / it is)
        self._sdma = None
        # interprocess events of the sdma transport: records per event (HIP's limit is 32), events per slot and direction
        self._sdma_event_records = int(os.environ.get("SCONE_SDMA_EVENT_RECORDS", "30"))
        self._sdma_event_pool = int(os.environ.get("SCONE_SDMA_EVENT_POOL", "16"))
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.n_gram_extractor = n_gram_extractor
        self.embedding_dim = int(embedding_dim)
        self.n_rows = int(len(n_gram_extractor) if n_rows is None else n_rows)
        self.row_begin, self.row_end = shard_range(self.n_rows, self.rank, self.world)
        if table is None:
            from scone_amd.hip_backend import SconeTable
            table = SconeTable(n_gram_extractor.max_n, self.n_rows, dim=self.embedding_dim, table_format=table_format,
                               placement=placement, device=device, row_begin=self.row_begin, row_end=self.row_end)
            n_gram_extractor.build_index(table)          # replicated index
            if replicated_rows:
                table.shard_set_head(replicated_rows)
        self.table = table
        self.replicated_rows = min(int(replicated_rows), self.n_rows)
        self._prof = None
        self._keep = None
        # split-phase "gather_rows" (gather_rows_begin / gather_rows_finish): side stream, two plan slots
        self._side = None
        self._slot_next = 0
        self._slot_done = [None] * 4
        self._slot_full = [None] * 4
        # The SENDER-side scratch of a plan (list of claimed rows, counters, chunk ends) exists once per handle, whatever the
        # slot: the next plan may only overwrite it after the last pack of the previous one has read it, on whichever stream
        # that pack ran -> an event after every plan's last pack, waited for by the stream of the next plan.  And a slot
        # whose ticket has not been finished must not be planned into again.
        self._plan_packed = None
        self._slot_open = [False] * 4
        self._slot_ell = [None] * 4          # sharded match: the gathered list records of the batch a slot holds
        self._slot_cols = [None] * 4         # columns exchange: (rows, scales, frags) receive buffers of a slot
        self._slot_head_ver = [None] * 4     # ... and the version of the replicated head whose scales their front holds
        self._ell_send = None
        if gather_transport == "sdma" and self.world > 1:
            self._sdma_init()

    @property
    def wire_format(self) -> str:
        """What a contribution of the "gather_rows" exchange looks like on the wire (follows ``gather_chunks``)."""
        return "columns" if self.gather_chunks <= 1 and hasattr(self.table, "shard_cols_pack") else "records"

    @classmethod
    def from_synthetic(cls, n_gram_extractor: NGramExtractor, embedding_dim: int, *, table_format: str = "int8",
                       seed: int = 7, base_scale: float = 0.02 / 127, **kw) -> "ShardedEmbeddingCache":
        """Every rank generates exactly its own rows on its GPU (counter-based, so the union equals
        the single-GPU synthetic table)."""
        self = cls(n_gram_extractor, embedding_dim, table_format=table_format, **kw)
        self.table.fill_synthetic(seed, base_scale)
        return self

    def load_rows(self, rows_f32: torch.Tensor, row0: int) -> None:
        """Store the part of ``rows_f32`` (global rows ``row0 ..``) that this rank owns."""
        a = max(row0, self.row_begin)
        b = min(row0 + rows_f32.shape[0], self.row_end)
        if b > a:
            self.table.store_f32(rows_f32[a - row0:b - row0], row0=a)
        hb = min(row0 + rows_f32.shape[0], self.replicated_rows)     # the replicated head: every rank keeps it
        if hb > row0:
            self.table.shard_head_store_f32(rows_f32[:hb - row0], row0=row0)
            self._slot_cols = [None] * 4                             # (their scales buffers start with the head's scales)

    # -- "sdma" transport: peer-mapped receive buffers, interprocess events, copy-engine pushes ------------------
    def _sdma_init(self) -> None:
        """Collective (every rank of the group constructs its cache): can this table / platform do it at all?  If any rank
        cannot, ALL fall back to "p2p" -- the ranks must agree on the transport."""
        t = self.table
        ok, why = True, None
        if not all(hasattr(t, m) for m in ("ipc_alloc", "ipc_event_create", "ipc_push", "shard_cols_pack")):
            ok, why = False, "the table handle has no scone_ipc_* entry points (a stand-in handle)"
        else:
            try:                                        # capability probe: one buffer + one event, exported
                p, _ = t.ipc_alloc(4096)
                e, _ = t.ipc_event_create()
                t.ipc_tensor(p, 4096).zero_()
                t.ipc_event_destroy(e)
                t.ipc_free(p)
            except Exception as ex:                     # noqa: BLE001 -- whatever HIP / torch refuse here means "not available"
                ok, why = False, f"interprocess handles are not available here: {ex!r}"
        flags = [None] * self.world
        _trace("all_gather_object(sdma capability)", self.group, ok=ok)
        dist.all_gather_object(flags, (ok, why), group=self.group)
        bad = [(r, w) for r, (o, w) in enumerate(flags) if not o]
        if bad:
            self.transport_fallback_reason = f"rank {bad[0][0]}: {bad[0][1]}"
            self.gather_transport = "p2p"
            return
        # host-side rendezvous of the step: the group itself when it is a host group (gloo), a gloo group the caller supplies
        # (`control_group=`), or one created beside the group here.  dist.new_group is collective over the DEFAULT world, not
        # over `group`: on a sub-group it would hang the ranks outside it, so without a supplied control group the transport
        # is only offered to caches whose group IS the world; a sub-group without one falls back to p2p (every rank alike).
        if _host_staged(self.group):
            ctrl = self.group
        elif self.control_group is not None:
            ctrl = self.control_group
        else:
            ranks = list(range(dist.get_world_size())) if self.group is None else dist.get_process_group_ranks(self.group)
            if len(ranks) != dist.get_world_size():
                self.transport_fallback_reason = ("gather_transport='sdma' on a sub-group needs control_group= (a gloo group of the "
                                                  "same ranks): dist.new_group is collective over the whole world")
                self.gather_transport = "p2p"
                return
            ctrl = dist.new_group(ranks=ranks, backend="gloo")
        dev = self.table.device
        self._sdma = {"ctrl": ctrl, "slots": [None] * 4,
                      "push": [torch.cuda.Stream(device=dev) for _ in range(self.world)],
                      "copy_engine": os.environ.get("SCONE_SDMA_COPY_ENGINE", "1") != "0"}
        why = self._sdma_self_test()
        if why is not None:                              # every rank gets the same verdict (all-gathered inside)
            self._sdma = None
            self.transport_fallback_reason = why
            self.gather_transport = "p2p"

    def _sdma_self_test(self) -> Optional[str]:
        """The transport's three mechanisms on 4 KB, before any table data depends on them (collective): every rank pushes a
        pattern of its own into its 64-byte cell of every peer's mapped buffer with the copy the transport uses, records an
        interprocess event, and every rank checks what arrived behind a wait for the peers' events.  A platform on which the
        handles open but the bytes do not arrive (peer access, a copy kind the topology does not support) -- which one
        GPU cannot show -- costs a fallback to p2p on every rank alike, not wrong rows.  Returns None, or the reason."""
        t, W, r, dev = self.table, self.world, self.rank, self.table.device
        ctrl = self._sdma["ctrl"]
        ptr = ev = mine = src = None
        peers, peer_ev = {}, {}
        state = {"why": None}

        def step(fn):                                    # a failure never skips a collective: it is recorded and carried along
            if state["why"] is None:
                try:
                    fn()
                except Exception as ex:                  # noqa: BLE001
                    state["why"] = f"self-test: {ex!r}"

        def everyone_ok() -> bool:
            flags = [None] * W
            dist.all_gather_object(flags, state["why"], group=self.group)
            bad = [(q, w) for q, w in enumerate(flags) if w is not None]
            if bad:                                      # the same verdict, worded alike, on every rank
                state["why"] = f"rank {bad[0][0]}: {bad[0][1]}"
            return not bad

        def alloc():
            nonlocal ptr, ev, mine, src
            ptr, h_mem = t.ipc_alloc(4096)
            ev, h_ev = t.ipc_event_create()
            mine = t.ipc_tensor(ptr, 4096)
            mine.zero_()
            src = torch.zeros(64, dtype=torch.uint8, device=dev)
            src[:8] = torch.tensor(list((0xC0DE0000 + r + 1).to_bytes(8, "little")), dtype=torch.uint8, device=dev)
            mine[64 * r:64 * r + 64] = src
            torch.cuda.synchronize(dev)
            state["handles"] = (h_mem, h_ev)

        step(alloc)
        everyone = [None] * W
        _trace("all_gather_object(sdma self-test handles)", self.group)
        dist.all_gather_object(everyone, state.get("handles"), group=self.group)

        def open_peers():
            for q in range(W):
                if q != r:
                    if everyone[q] is None:
                        raise RuntimeError(f"rank {q} exported no handles")
                    peers[q] = t.ipc_open(everyone[q][0])
                    peer_ev[q] = t.ipc_event_open(everyone[q][1])

        step(open_peers)
        if everyone_ok():                                # (the all-gather is also the rendezvous: every buffer is zeroed and mapped)
            sabotage = os.environ.get("SCONE_SDMA_SELF_TEST_SKIP_PUSH_OF_RANK")      # test hook: that rank's bytes never arrive

            def push():
                for q in range(W):
                    if q != r and sabotage != str(r):
                        with torch.cuda.stream(self._sdma["push"][q]):
                            t.ipc_push(peers[q] + 64 * r, src.data_ptr(), 64, self._sdma["copy_engine"])
                        torch.cuda.current_stream().wait_stream(self._sdma["push"][q])
                t.ipc_event_record(ev)

            step(push)
            dist.barrier(group=ctrl)                     # every "sent" is recorded before anyone waits for it

            def check():
                for q in range(W):
                    if q != r:
                        t.ipc_event_wait(peer_ev[q])
                got = mine.clone().cpu()                 # (stream-ordered behind the waits)
                for q in range(W):
                    if bytes(got[64 * q:64 * q + 8].tolist()) != (0xC0DE0000 + q + 1).to_bytes(8, "little"):
                        raise RuntimeError(f"the 64 bytes pushed by rank {q} did not arrive in rank {r}'s mapped buffer")

            step(check)
            everyone_ok()
        try:
            torch.cuda.synchronize(dev)
        except Exception:                                # noqa: BLE001
            pass
        dist.barrier(group=ctrl)                         # nobody still pushes into, or waits on, what goes away
        for q, p_ in peers.items():
            try:
                t.ipc_close(p_)
            except Exception:                            # noqa: BLE001 -- cleaning up after a failed probe must not raise
                pass
        for e in list(peer_ev.values()) + ([ev] if ev is not None else []):
            try:
                t.ipc_event_destroy(e)
            except Exception:                            # noqa: BLE001
                pass
        if ptr is not None:
            try:
                t.ipc_free(ptr)
            except Exception:                            # noqa: BLE001
                pass
        return state["why"]

    def set_gather_transport(self, transport: str) -> str:
        """Switch the transport of the "gather_rows" exchange on an existing cache -- collective: every rank calls it with the
        same value (asking for "sdma" runs the capability probe and the handle exchange set-up).  Returns the transport in use
        ("p2p" when "sdma" is not available; ``transport_fallback_reason`` says why)."""
        if transport not in ("p2p", "all_gather", "sdma"):
            raise ValueError("gather_transport must be 'p2p', 'all_gather' or 'sdma'")
        self.gather_transport = transport
        self.transport_fallback_reason = None
        if transport == "sdma" and self.world > 1 and self._sdma is None:
            self._sdma_init()
        return self.gather_transport

    # Release order, everywhere (close(), a slot that grows, an event pool that is used up): first every rank lets go of what
    # it OPENED from its peers, then -- after a host barrier -- of what it OWNS.  HIP leaves freeing exported memory (or
    # destroying an exported event) that another process still has open undefined.
    def _sdma_close_peer_buffers(self, st) -> None:
        t = self.table
        for r in range(self.world):
            if r != self.rank and st["peer"][r] is not None:
                for ptr in st["peer"][r]:
                    if ptr:
                        t.ipc_close(ptr)
                st["peer"][r] = None

    def _sdma_close_peer_events(self, st) -> None:
        t = self.table
        for r in range(self.world):
            if r != self.rank and st["peer_sent"][r] is not None:
                for e in st["peer_sent"][r] + st["peer_done"][r]:
                    t.ipc_event_destroy(e)
        st["peer_sent"], st["peer_done"] = [None] * self.world, [None] * self.world

    def _sdma_free_own_events(self, st) -> None:
        for e in st["sent"] + st["done"]:
            self.table.ipc_event_destroy(e)
        st["sent"], st["done"] = [], []

    def _sdma_free_own(self, st) -> None:
        for ptr in st["ptrs"]:
            if ptr:
                self.table.ipc_free(ptr)
        st["ptrs"] = []
        self._sdma_free_own_events(st)

    def _sdma_new_events(self, st, slot: int) -> None:
        """A fresh pool of interprocess events for this slot, exchanged with and opened by every peer (collective; the caller
        has made sure nobody still waits for, or records, the old ones).  A HIP interprocess event survives 32 records -- the
        33rd makes every later hipStreamWaitEvent of a process that opened it fail with "invalid argument" (measured:
        tools/ipc_event_probe.py; the split-phase soak found it) -- so each event of the pool is used for
        ``_sdma_event_records`` steps, and the pool is replaced when it is used up."""
        t, W, r = self.table, self.world, self.rank
        K = self._sdma_event_pool
        pairs = [t.ipc_event_create() for _ in range(2 * K)]
        st["sent"], st["done"] = [e for e, _ in pairs[:K]], [e for e, _ in pairs[K:]]
        everyone = [None] * W
        _trace("all_gather_object(sdma event handles)", self.group, slot=slot, events=2 * K)
        dist.all_gather_object(everyone, [h for _, h in pairs], group=self.group)
        for q in range(W):
            if q != r:
                st["peer_sent"][q] = [t.ipc_event_open(h) for h in everyone[q][:K]]
                st["peer_done"][q] = [t.ipc_event_open(h) for h in everyone[q][K:]]
        st["uses"], st["cur_ev"], st["used"] = 0, 0, False

    def _sdma_renew_events(self, st, slot: int) -> None:
        """The slot's event pool is used up: every rank arrives here in the same step (pushes are collective)."""
        torch.cuda.synchronize(self.table.device)        # my waits for the old events, and my records of them, are complete
        dist.barrier(group=self._sdma["ctrl"])           # ... and so are everybody else's
        self._sdma_close_peer_events(st)                 # what I opened goes first ...
        dist.barrier(group=self._sdma["ctrl"])           # ... everywhere, before any owner destroys what it exported
        self._sdma_free_own_events(st)
        self._sdma_new_events(st, slot)
        dist.barrier(group=self._sdma["ctrl"])           # every rank has opened every new event before anyone records one

    def _sdma_slot(self, slot: int, total: int, ftotal: int, dev):
        """This slot's receive buffers (rows, scales, frags) with room for ``total`` rows / ``ftotal`` fragment slots, mapped
        into every peer.  Every rank sees the same totals, so every rank (re)allocates in the same step: free what was
        there (after the device and the peers are idle), allocate, exchange the handles, open the peers'."""
        t, W, r = self.table, self.world, self.rank
        st = self._sdma["slots"][slot]
        if st is not None and st["cap"] >= max(total, 1) and st["fcap"] >= ftotal:
            return st
        pb, sb, nh = t.payload_bytes(), t.scale_bytes(), int(getattr(t, "n_head", 0) or 0)
        torch.cuda.synchronize(dev)
        dist.barrier(group=self._sdma["ctrl"])           # nobody pushes into, or reads from, the buffers that go away
        if st is not None:                               # (every rank takes this branch in the same step: same totals everywhere)
            self._sdma_close_peer_buffers(st)
            self._sdma_close_peer_events(st)
            dist.barrier(group=self._sdma["ctrl"])       # what a rank has opened is closed everywhere before any owner frees it
            self._sdma_free_own(st)
        cap, fcap = max(total + total // 8, 1), max(ftotal + ftotal // 8, 64)
        nbytes = (cap * pb, (nh + cap) * sb, fcap * 8)
        ptrs, handles = [], []
        for nb in nbytes:
            if nb:
                p, hb = t.ipc_alloc(nb)
            else:
                p, hb = 0, b"\0" * 64
            ptrs.append(p)
            handles.append(hb)
        mine = b"".join(handles)
        everyone = [None] * W
        _trace("all_gather_object(sdma handles)", self.group, slot=slot, rows=cap, frag_slots=fcap)
        dist.all_gather_object(everyone, mine, group=self.group)
        peer = [None] * W
        for q in range(W):
            if q == r:
                continue
            hq = everyone[q]
            peer[q] = tuple(t.ipc_open(hq[64 * i:64 * i + 64]) if nbytes[i] else 0 for i in range(3))
        rows = t.ipc_tensor(ptrs[0], nbytes[0]).view(cap, pb)
        scales = t.ipc_tensor(ptrs[1], nbytes[1]).view(nh + cap, sb) if sb else None
        frags = t.ipc_tensor(ptrs[2], nbytes[2]).view(torch.int64)
        torch.cuda.synchronize(dev)                      # (the head's scales go into the front of `scales` in _gather_begin_cols)
        dist.barrier(group=self._sdma["ctrl"])           # every rank has opened every buffer before anyone pushes
        st = {"cap": cap, "fcap": fcap, "ptrs": ptrs, "peer": peer, "sent": [], "done": [], "peer_sent": [None] * W,
              "peer_done": [None] * W, "rows": rows, "scales": scales, "frags": frags, "used": False, "uses": 0, "cur_ev": 0}
        self._sdma_new_events(st, slot)
        torch.cuda.synchronize(dev)
        dist.barrier(group=self._sdma["ctrl"])           # ... and every event
        self._sdma["slots"][slot] = st
        return st

    def _sdma_push(self, st, slot: int, regions, rendezvous: bool = False) -> "_SdmaArrival":
        """``regions``: [(column index, byte offset, bytes)] of this rank's freshly packed range.  Waits (stream-ordered) until
        every peer has reduced the batch that used this slot before, pushes the ranges to the same offsets of every peer's
        buffers -- one stream per peer, so that the copy engines drive all links at once --, records "sent", and rendezvous
        on the host so that the receivers' waits see this record."""
        t, W, r = self.table, self.world, self.rank
        if st["uses"] >= self._sdma_event_pool * self._sdma_event_records:
            self._sdma_renew_events(st, slot)
        cur = torch.cuda.current_stream()
        if st["used"]:
            # A wait sees the most recent record AT THE TIME OF THE CALL: every peer must have CALLED the finish of this slot's
            # previous batch (which records "reduced") before anyone waits for it.  The exact form's count exchange proves that
            # (a peer enters it only after that finish); the sync-free form has no such collective in front, so it asks for a
            # host barrier here -- without it a peer whose host runs behind is overwritten before it has reduced (found by the
            # 100,000-step soak: 2 batches in 11,000 with a row missing).
            if rendezvous:
                dist.barrier(group=self._sdma["ctrl"])
            for q in range(W):
                if q != r:
                    t.ipc_event_wait(st["peer_done"][q][st["cur_ev"]])
        ev = st["uses"] // self._sdma_event_records
        packed = torch.cuda.Event()
        packed.record(cur)
        for q in range(W):
            if q == r:
                continue
            ps = self._sdma["push"][q]
            ps.wait_event(packed)
            with torch.cuda.stream(ps):
                for col, off, nb in regions:
                    if nb:
                        t.ipc_push(st["peer"][q][col] + off, st["ptrs"][col] + off, nb, self._sdma["copy_engine"])
                e = torch.cuda.Event()
                e.record(ps)
            cur.wait_event(e)
        t.ipc_event_record(st["sent"][ev])
        st["cur_ev"], st["uses"] = ev, st["uses"] + 1
        _trace("barrier(sdma: pushes recorded)", self.group, bytes_per_peer=sum(nb for _, _, nb in regions))
        dist.barrier(group=self._sdma["ctrl"])
        return _SdmaArrival(t, [st["peer_sent"][q][ev] for q in range(W) if q != r])

    def close(self) -> None:
        """Release the interprocess buffers and events of the "sdma" transport -- COLLECTIVE: every rank calls it once the loop is
        over (two host barriers inside: what a rank has opened is closed everywhere before any owner frees it)."""
        if self._sdma is not None:
            # collective: nobody frees a buffer a peer may still push into, or destroys an event a peer still waits for or has
            # open -- first everything queued completes everywhere, then every rank lets go of what it OPENED, then of what it OWNS
            ctrl, t = self._sdma["ctrl"], self.table
            torch.cuda.synchronize(t.device)
            dist.barrier(group=ctrl)
            slots = [st for st in self._sdma["slots"] if st is not None]
            for st in slots:
                self._sdma_close_peer_buffers(st)
                self._sdma_close_peer_events(st)
            dist.barrier(group=ctrl)
            for st in slots:
                self._sdma_free_own(st)
            self._sdma = None

    # ------------------------------------------------------------------
    def embed_tokens(self, input_ids: torch.Tensor, *, reduce: str = "mean", wte: Optional[torch.Tensor] = None,
                     wpe: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                     out_dtype: Optional[torch.dtype] = None, gather_output: bool = True,
                     exchange: str = "auto", profile: bool = False, check: bool = False):
        """Same result as ``EmbeddingCache.embed_tokens`` on the unsharded table -- bit-identical with the row
        exchanges, up to the fp32 summation order across shards with ``"partial_sums"``.  Every rank passes the SAME
        ``input_ids [B, T]``.

        ``gather_output=True``: every rank gets the whole ``[B, T, d]``; ``False``: only this rank's slice
        ``[ceil(B/W)*T, d]`` (zero-padded at the tail; for consumers that are themselves data-parallel over the same
        slices).  ``exchange``: ``"auto"`` (``"gather_rows"`` for the whole output, ``"rows"`` for slices), ``"rows"``,
        ``"gather_rows"`` or ``"partial_sums"`` -- see the module docstring.

        ``profile=True`` (row exchanges only): returns ``(out, phases)`` -- the device is synchronised between the phases
        of the step and ``phases`` holds their milliseconds (``plan_ms, pack_ms, collective_ms, embed_ms, gather_out_ms``)
        and ``bytes_received`` (payload this rank received over the group); an instrumented step, not a fast one.
        ``check=True``: raise if the step left a status bit behind (a row that never arrived, a token outside ``wte``)."""
        self._prof = {} if profile else None
        out = self._embed_tokens(input_ids, reduce, wte, wpe, position_ids, out_dtype, gather_output, exchange)
        if check:                                       # (synchronises: the sticky status bits of every kernel of the step)
            self._check_status("embed_tokens")
        if profile:
            prof, self._prof = self._prof, None
            return out, prof
        return out

    def _tick(self, name: str, t0: float) -> float:
        """profile mode: close phase ``name`` (synchronise, add the elapsed milliseconds), return the new start time."""
        import time
        if self._prof is None:
            return t0
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self._prof[name] = self._prof.get(name, 0.0) + (t1 - t0) * 1e3
        return t1

    def _embed_tokens(self, input_ids, reduce, wte, wpe, position_ids, out_dtype, gather_output, exchange):
        tok = torch.as_tensor(input_ids)
        if tok.dim() == 1:
            tok = tok.unsqueeze(0)
        B, T = tok.shape
        ntok, d, W = B * T, self.embedding_dim, self.world
        if out_dtype is None:
            out_dtype = wte.dtype if wte is not None else (wpe.dtype if wpe is not None else torch.float32)
        if exchange == "auto":
            # the whole output on every rank: gather the distinct rows (a tenth of the bytes of gathering the finished
            # vectors); every rank keeps its own slice: send each slice the rows it needs
            exchange = "gather_rows" if gather_output else "rows"
        if W == 1 and exchange in ("rows", "gather_rows") and hasattr(self.table, "embed"):
            # one shard owns every row: the plain fused lookup, nothing to exchange
            out = self.table.embed(tok, wte=wte, wpe=wpe, position_ids=position_ids, reduce=reduce, out_dtype=out_dtype)
            return out if gather_output else out.reshape(ntok, d)
        if exchange == "rows":
            return self._embed_row_exchange_dedup(tok, reduce, wte, wpe, position_ids, out_dtype, gather_output)
        if exchange == "gather_rows":
            out = self._embed_gather_rows(tok, reduce, wte, wpe, position_ids, out_dtype)
            if gather_output:
                return out.reshape(B, T, d)
            bper = (B + W - 1) // W                                          # same slice convention as "rows"
            b0, b1 = min(self.rank * bper, B), min(self.rank * bper + bper, B)
            sl = out.new_zeros((bper * T, d))
            sl[:(b1 - b0) * T] = out[b0 * T:b1 * T]
            return sl
        if exchange != "partial_sums":
            raise ValueError("exchange must be 'rows', 'gather_rows' or 'partial_sums'")
        partial, counts = self.table.embed_partial(tok)                      # [ntok, d] fp32, [ntok] int32
        per = (ntok + W - 1) // W                                             # tokens per rank (last slices padded)
        a = min(self.rank * per, ntok)
        b = min(a + per, ntok)
        if W > 1:
            pad = per * W - ntok
            if pad:
                partial = torch.cat([partial, partial.new_zeros((pad, d))])
            mine = torch.empty((per, d), dtype=torch.float32, device=partial.device)
            _reduce_scatter_sum(mine, partial, self.group)
        else:
            mine = partial                                                    # one shard: nothing to exchange
        if b - a == per:
            out_slice = torch.empty((per, d), dtype=out_dtype, device=partial.device)
        else:
            out_slice = torch.zeros((per, d), dtype=out_dtype, device=partial.device)   # padded tail slice
        if b > a:
            self.table.finalize(mine[:b - a], counts[a:b], tok, a, b, wte=wte, wpe=wpe, position_ids=position_ids,
                                reduce=reduce, out_dtype=out_dtype, out=out_slice[:b - a])
        if not gather_output:
            return out_slice
        if W > 1:
            full = torch.empty((per * W, d), dtype=out_dtype, device=partial.device)
            _all_gather(full, out_slice, self.group)
        else:
            full = out_slice
        return full[:ntok].reshape(B, T, d)

    def _embed_gather_rows(self, tok, reduce, wte, wpe, position_ids, out_dtype):
        """All-gather of the quantised rows the batch references; every rank then embeds the whole batch.

        Pipelined over ``gather_chunks`` runs of sequences: ONE plan (one match of the batch; one claim pass per chunk, a
        row claimed by an earlier chunk is not sent again), ONE exchange of the per-chunk record counts, then the packs and
        the C all-gathers are queued back to back and chunk c is reduced as soon as ITS records are in -- the transfers of
        chunks c+1.. overlap the reduction of chunk c, which is what every rank spends most of the step on (it reduces the
        whole batch).  The receive buffer is one allocation ``[sum_c W * max_c, record]`` (a contribution is padded to
        the largest of its chunk; padding records carry row id 0xFFFFFFFF and are skipped), so the lookup kernel reads
        all records received so far as one row store.

        The two halves are also public -- :meth:`gather_rows_begin` / :meth:`gather_rows_finish` -- so that a serving loop
        can run the first half of batch b + 1 on a side stream while batch b is being reduced."""
        if not hasattr(self.table, "shard_gather_plan_chunks"):          # stand-in handles (CPU tests of the exchange logic)
            return self._embed_gather_rows_unchunked(tok, reduce, wte, wpe, position_ids, out_dtype)
        ticket = self.gather_rows_begin(tok, overlap=False)
        return self.gather_rows_finish(ticket, reduce=reduce, wte=wte, wpe=wpe, position_ids=position_ids,
                                       out_dtype=out_dtype).reshape(-1, self.embedding_dim)

    # -- split-phase form of "gather_rows": plan + pack + collectives | reduction ------------------------------
    def gather_rows_begin(self, input_ids: torch.Tensor, *, overlap: bool = True, tokens_ready="current") -> dict:
        """First half of the ``"gather_rows"`` exchange for one batch (the same ``[B, T]`` on every rank): match + claim
        passes, the exchange of the record counts, the packs, and the all-gathers of the records (asynchronous over RCCL).
        Returns a ticket for :meth:`gather_rows_finish`.

        ``overlap=True``: the work is queued on a side stream of this cache and uses the plan slot the previous ``begin``
        did not, so a loop ::

            ticket = cache.gather_rows_begin(batch[0])
            for b in range(n):
                out = cache.gather_rows_finish(ticket, wte=wte, wpe=wpe)        # queues the reduction of batch b
                ticket = cache.gather_rows_begin(batch[b + 1]) if b + 1 < n else None   # ... and this overlaps it

        hides plan, pack and the transfers of batch b + 1 behind the reduction of batch b: the step is then bound by the
        reduction alone.  At most ``plan_slots`` batches in flight (2 by default; with 3, ``begin`` is called two batches
        ahead and the chain plan -> transfers may take two reductions' time); tickets are finished in the order they were
        begun.  ``begin`` blocks the host until the side stream has planned the batch (the record counts size the
        buffers), not until the device is idle.  (Tried in round 3 and dropped: indexing the records and rewriting the id
        lists on a third stream behind the transfers, so that ``finish`` launches the reduction alone -- at C5's true scale
        the step did not move, 0.926 against 0.921 ms, ``profiles/r03h/c5_rank0_step_alternating_six_flows_slower_box.json``; the side stream never waits for the transfers
        either way, the caller's stream does.)

        ``tokens_ready`` says what the side stream must wait for before it reads the tokens: a ``torch.cuda.Event``
        recorded where they were produced; ``None`` -- nothing, they are complete (uploaded earlier, or produced by work the
        host has synchronised with); ``"current"`` (the default for a device tensor: always safe) -- everything queued on the
        caller's stream so far.  NB the default puts the plan BEHIND a reduction queued just before it and with it most of the
        overlap is lost: a serving loop passes the event of its token producer, or host tokens (copied on the side stream,
        nothing to wait for)."""
        if self._prof is not None and overlap:
            raise ValueError("profile=True measures the phases one after the other: use overlap=False")
        import time
        t = self.table
        tok = torch.as_tensor(input_ids)
        if tok.dim() == 1:
            tok = tok.unsqueeze(0)
        if not overlap:
            return self._gather_begin(t._tok(tok), 0, time.perf_counter() if self._prof is not None else 0.0)
        slot = self._slot_next
        self._slot_next = (slot + 1) % self.plan_slots
        dev = getattr(t, "device", None)
        if dev is None or torch.device(dev).type != "cuda":   # stand-in tables of the CPU tests: the slots, no streams
            return self._gather_begin(t._tok(tok), slot, 0.0)
        if self._side is None:
            # (round 6: a side stream on the hardware queue of the caller's stream overlaps with nothing -- chosen by probe)
            self._side = t.pick_side_stream() if hasattr(t, "pick_side_stream") else torch.cuda.Stream(device=dev)
        side = self._side
        if tok.is_cuda:
            if isinstance(tokens_ready, str):
                if tokens_ready != "current":
                    raise ValueError("tokens_ready must be an event, None or 'current'")
                side.wait_stream(torch.cuda.current_stream())
            elif tokens_ready is not None:
                side.wait_event(tokens_ready)
        if self._slot_done[slot] is not None:                 # the slot's buffers are free once ITS last batch has been reduced
            side.wait_event(self._slot_done[slot])
        with torch.cuda.stream(side):
            ticket = self._gather_begin(t._tok(tok), slot, 0.0)   # host tokens are uploaded here, on the side stream
            ticket["ready"] = torch.cuda.Event()
            ticket["ready"].record(side)
        return ticket

    def _plan_enter(self, slot: int, tok) -> None:
        """Before a plan: the slot must be free, and this stream must not overwrite the sender-side scratch while the
        previous plan's last pack (possibly on another stream) is still reading it."""
        if self._slot_open[slot]:
            raise RuntimeError(f"plan slot {slot} still holds a batch begun with gather_rows_begin: call gather_rows_finish (or "
                               f"gather_rows_abandon) for it first -- at most {self.plan_slots - 1 if self.plan_slots > 2 else 2} batches "
                               f"in flight with plan_slots={self.plan_slots}, finished in the order they were begun")
        if self._plan_packed is not None and tok.is_cuda:
            torch.cuda.current_stream().wait_event(self._plan_packed)
        self.table.shard_select_slot(slot)

    def _plan_packed_here(self, tok) -> None:
        if tok.is_cuda:
            self._plan_packed = torch.cuda.Event()
            self._plan_packed.record(torch.cuda.current_stream())

    def _plan(self, tok, slot: int, n_chunks: int, dedup_across_chunks: bool) -> list:
        """Match + claim passes of one batch on plan slot ``slot``; returns the chunk ends (synchronises the stream).
        With the sharded match this rank matches only its own run of sequences and the list records of all runs are
        all-gathered -- index and tokens are replicated, so the gathered lists are what a local match of the whole batch
        would have produced, at 1/W of the probes per rank."""
        B, T = tok.shape
        W, t = self.world, self.table
        on = self.shard_match
        if on == "auto":
            on = B * T >= 65536
        if not (on and W > 1 and B >= W and hasattr(t, "shard_gather_match")):
            return t.shard_gather_plan_chunks(tok, n_chunks, dedup_across_chunks)
        bper = (B + W - 1) // W                                           # the slice convention of every exchange here
        b0, b1 = min(self.rank * bper, B), min(self.rank * bper + bper, B)
        wd = t.ell_width()
        ell = self._slot_ell[slot]
        if ell is None or ell.shape[0] < W * bper * T or ell.shape[1] != wd or ell.device != tok.device:
            ell = torch.empty((W * bper * T, wd), dtype=torch.int32, device=tok.device)
            self._slot_ell[slot] = ell
        send = self._ell_send
        if send is None or send.shape[0] != bper * T or send.shape[1] != wd or send.device != tok.device:
            send = torch.zeros((bper * T, wd), dtype=torch.int32, device=tok.device)   # (a short last slice leaves zero records:
            self._ell_send = send                                                       #  tokens past B * T, never read)
        t.shard_gather_match(tok, b0, b1, send)
        _all_gather(ell[:W * bper * T].view(-1), send.view(-1), self.group)             # 32 B per token: W * bper * T * 32 B in all
        return t.shard_gather_plan_ell(ell, B, T, n_chunks, dedup_across_chunks)

    def _note_counts(self, counts) -> None:
        """What the ranks contributed to an exchange -> the capacities of the next sync-free ones (+ 12.5 %, never shrinking)."""
        want = [int(c) + int(c) // 8 + 16 for c in counts]
        self._caps = want if self._caps is None else [max(a, b) for a, b in zip(self._caps, want)]

    def _plan_async(self, tok, slot: int) -> None:
        """:meth:`_plan` (one chunk) without its host round trip: the count of claimed rows stays on the device."""
        B, T = tok.shape
        W, t = self.world, self.table
        on = self.shard_match
        if on == "auto":
            on = B * T >= 65536
        if not (on and W > 1 and B >= W and hasattr(t, "shard_gather_match")):
            return t.shard_gather_plan_async(tok)
        bper = (B + W - 1) // W
        b0, b1 = min(self.rank * bper, B), min(self.rank * bper + bper, B)
        wd = t.ell_width()
        ell = self._slot_ell[slot]
        if ell is None or ell.shape[0] < W * bper * T or ell.shape[1] != wd or ell.device != tok.device:
            ell = torch.empty((W * bper * T, wd), dtype=torch.int32, device=tok.device)
            self._slot_ell[slot] = ell
        send = self._ell_send
        if send is None or send.shape[0] != bper * T or send.shape[1] != wd or send.device != tok.device:
            send = torch.zeros((bper * T, wd), dtype=torch.int32, device=tok.device)
            self._ell_send = send
        t.shard_gather_match(tok, b0, b1, send)
        _all_gather(ell[:W * bper * T].view(-1), send.view(-1), self.group)
        t.shard_gather_plan_ell_async(ell, B, T)

    def _gather_begin_cols_sync_free(self, tok, slot):
        """The one-piece exchange sized from the previous batches: nothing here waits for the device.  Rank q's region of the
        receive buffers holds ``caps[q]`` rows / ``frag_slots(caps[q]) + 2`` fragment words -- the two extra words are its
        header (rows it claimed, overflow flag), written by its pack kernel; the headers are copied to pinned host memory on a
        stream of their own behind the transfers and read in :meth:`gather_rows_finish`."""
        B, T = tok.shape
        W, t, r = self.world, self.table, self.rank
        HDR = 2
        caps = list(self._caps)
        slots_r = [t.cols_frag_slots(c) for c in caps]
        sdma = self.gather_transport == "sdma" and self._sdma is not None
        exact = self.gather_transport == "p2p" or sdma
        if not exact:                                    # all_gather_into_tensor: every region as large as the largest
            caps, slots_r = [max(caps)] * W, [max(slots_r)] * W
        rec_base = [sum(caps[:q]) for q in range(W)]
        frag_off = [sum(slots_r[:q]) + HDR * q for q in range(W)]
        total, ftotal = sum(caps), sum(slots_r) + HDR * W
        pb, sb, nh = t.payload_bytes(), t.scale_bytes(), int(getattr(t, "n_head", 0) or 0)
        bufs = self._slot_cols[slot]
        st = None
        if sdma:                                         # peer-mapped receive buffers (every rank has the same capacities:
            st = self._sdma_slot(slot, total, ftotal, tok.device)      # every rank re-allocates in the same step, rarely)
            bufs = (st["rows"], st["scales"], st["frags"])
        elif bufs is None or bufs[0].shape[0] < total or bufs[2].numel() < ftotal or bufs[0].device != tok.device:
            cap = total + total // 8
            rows = torch.empty((cap, pb), dtype=torch.uint8, device=tok.device)
            scales = torch.empty((nh + cap, sb), dtype=torch.uint8, device=tok.device) if sb else None
            frags = torch.empty(max(ftotal + ftotal // 8, 64), dtype=torch.int64, device=tok.device)
            bufs = self._slot_cols[slot] = (rows, scales, frags)
            self._slot_head_ver[slot] = None
        rows, scales, frags = bufs
        if scales is not None and nh:
            hv = t.shard_head_version() if hasattr(t, "shard_head_version") else 0
            key = (hv, scales.data_ptr())
            if self._slot_head_ver[slot] != key:
                t.shard_head_scales_into(scales)
                self._slot_head_ver[slot] = key
        self._plan_enter(slot, tok)
        self._plan_async(tok, slot)
        works, keep = [], None
        if exact:
            fr = frags[frag_off[r]:frag_off[r] + slots_r[r] + HDR]
            t.shard_cols_pack_cap(caps[r], rows[rec_base[r]:rec_base[r] + caps[r]],
                                  None if scales is None else scales[nh + rec_base[r]:nh + rec_base[r] + caps[r]],
                                  fr[:slots_r[r]], fr[slots_r[r]:])
            if sdma:                                     # capacity-sized ranges (the count never comes to the host), header included
                works.append(self._sdma_push(st, slot, [(0, rec_base[r] * pb, caps[r] * pb), (1, (nh + rec_base[r]) * sb, caps[r] * sb),
                                                        (2, frag_off[r] * 8, (slots_r[r] + HDR) * 8)], rendezvous=True))
            else:
                works.append(_exchange_exact_async(rows[:total], rec_base + [total], caps, r, self.group))
                if scales is not None:
                    works.append(_exchange_exact_async(scales[nh:nh + total], rec_base + [total], caps, r, self.group))
                works.append(_exchange_exact_async(frags[:ftotal].view(-1, 1), frag_off + [ftotal], [s + HDR for s in slots_r], r, self.group))
        else:
            m, ms = caps[0], slots_r[0]
            s_rows = torch.empty((m, pb), dtype=torch.uint8, device=tok.device)
            s_scales = torch.empty((m, sb), dtype=torch.uint8, device=tok.device) if sb else None
            s_frag = torch.empty(ms + HDR, dtype=torch.int64, device=tok.device)
            t.shard_cols_pack_cap(m, s_rows, s_scales, s_frag[:ms], s_frag[ms:])
            works.append(_all_gather_async(rows[:total].view(-1), s_rows.reshape(-1), self.group))
            if scales is not None:
                works.append(_all_gather_async(scales[nh:nh + total].view(-1), s_scales.reshape(-1), self.group))
            works.append(_all_gather_async(frags[:ftotal], s_frag, self.group))
            keep = (s_rows, s_scales, s_frag)
        self._plan_packed_here(tok)
        self._slot_open[slot] = True
        # the headers: behind the fragment transfer, on their own stream (the stream of this call must not wait for transfers)
        idx = torch.tensor([frag_off[q] + slots_r[q] + k for q in range(W) for k in range(HDR)], dtype=torch.int64)
        if tok.is_cuda:
            if self._hdr_stream is None:
                self._hdr_stream = torch.cuda.Stream(device=tok.device)
            hs = self._hdr_stream
            hs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hs):
                works[-1].wait()                                      # (this stream waits; the host does not)
                if sdma:                                              # an interprocess "sent" record is waited for ONCE: whoever
                    arrived = torch.cuda.Event()                      # needs the columns later waits for this stream's event
                    arrived.record(hs)
                    works = [_AfterEvent(arrived)]
                hdr = torch.empty(W * HDR, dtype=torch.int64).pin_memory()
                hdr.copy_(frags.index_select(0, idx.to(tok.device, non_blocking=True)), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(hs)
        else:                                                        # (host tensors: gloo completes requests in posting order,
            for w in works:                                          #  and a request must be waited for exactly once)
                w.wait()
            works = [_Done()]
            hdr, ev = frags.index_select(0, idx).clone(), None
        self.sync_free_stats["exchanges"] += 1
        return {"slot": slot, "tok": tok, "C": 1, "per": B, "works": works, "ready": None, "t0": 0.0, "keep": keep, "sdma": st,
                "hdr": (hdr, ev, caps),
                "cols": {"rows": rows, "scales": scales, "frags": frags, "total": total, "frag_off": frag_off,
                         "frag_slots": slots_r, "rec_base": rec_base}}

    def _gather_begin_cols(self, tok, slot, t0, exact_only: bool = False):
        """One-piece ``gather_rows`` with COLUMNS on the wire: plan, exchange of the counts, then this rank's payload rows,
        scales and hash fragment go out as three ranges (``p2p``: packed in place into this rank's range of the receive
        buffers, one ``batch_isend_irecv`` per column) or three padded all-gathers.  (From the second batch on, unless
        ``exact_only``: the sync-free form, :meth:`_gather_begin_cols_sync_free`.)"""
        B, T = tok.shape
        W, t, r = self.world, self.table, self.rank
        if (not exact_only and self.sync_free_plan and self._caps is not None and W > 1 and self._prof is None
                and (self.gather_transport in ("p2p", "all_gather") or (self.gather_transport == "sdma" and self._sdma is not None))
                and hasattr(t, "shard_cols_pack_cap")):
            return self._gather_begin_cols_sync_free(tok, slot)
        self._plan_enter(slot, tok)
        n_me = self._plan(tok, slot, 1, True)[0]
        t0 = self._tick("plan_ms", t0)
        if W > 1:
            cnt = torch.tensor([n_me], dtype=torch.int64, device=tok.device)
            allc = torch.empty(W, dtype=torch.int64, device=tok.device)
            _all_gather(allc, cnt, self.group)
            counts = [int(c) for c in allc.tolist()]
        else:
            counts = [int(n_me)]
        sdma = W > 1 and self.gather_transport == "sdma" and self._sdma is not None
        exact = W > 1 and self.gather_transport in ("p2p", "sdma")
        self._note_counts(counts)
        slots_r = [t.cols_frag_slots(c) for c in counts]
        if exact or W == 1:
            rec_base = [sum(counts[:q]) for q in range(W)]
            frag_off = [sum(slots_r[:q]) for q in range(W)]
            total, ftotal = sum(counts), sum(slots_r)
        else:                                            # all_gather_into_tensor: every column padded to its largest contribution
            m, ms = max(counts), max(slots_r)
            slots_r = [ms] * W                           # (a fragment may have more slots than it needs: every rank uses ms)
            rec_base, frag_off = [q * m for q in range(W)], [q * ms for q in range(W)]
            total, ftotal = W * m, W * ms
        pb, sb, nh = t.payload_bytes(), t.scale_bytes(), int(getattr(t, "n_head", 0) or 0)
        bufs = self._slot_cols[slot]
        st = None
        if sdma:                                         # receive buffers every peer has mapped (collective when they grow)
            st = self._sdma_slot(slot, total, ftotal, tok.device)
            bufs = (st["rows"], st["scales"], st["frags"])
        elif bufs is None or bufs[0].shape[0] < max(total, 1) or bufs[2].numel() < ftotal or bufs[0].device != tok.device:
            cap = max(total + total // 8, 1)
            rows = torch.empty((cap, pb), dtype=torch.uint8, device=tok.device)
            scales = torch.empty((nh + cap, sb), dtype=torch.uint8, device=tok.device) if sb else None
            frags = torch.empty(max(ftotal + ftotal // 8, 64), dtype=torch.int64, device=tok.device)
            bufs = self._slot_cols[slot] = (rows, scales, frags)
            self._slot_head_ver[slot] = None
        rows, scales, frags = bufs
        if scales is not None and nh:
            # [head scales | received scales]: the front part once per buffer AND per version of the head -- a head changed
            # through the table handle (shard_head_store_f32, fill_synthetic, ...) must not leave stale scales here
            hv = t.shard_head_version() if hasattr(t, "shard_head_version") else 0
            key = (hv, scales.data_ptr())
            if self._slot_head_ver[slot] != key:
                t.shard_head_scales_into(scales)
                self._slot_head_ver[slot] = key
        t0 = self._tick("collective_ms", t0)
        works, keep = [], None
        if exact or W == 1:
            t.shard_cols_pack(0, n_me, rows[rec_base[r]:rec_base[r] + n_me], None if scales is None else
                              scales[nh + rec_base[r]:nh + rec_base[r] + n_me], frags[frag_off[r]:frag_off[r] + slots_r[r]])
            t0 = self._tick("pack_ms", t0)
            if sdma:
                works.append(self._sdma_push(st, slot, [(0, rec_base[r] * pb, n_me * pb), (1, (nh + rec_base[r]) * sb, n_me * sb),
                                                  (2, frag_off[r] * 8, slots_r[r] * 8)]))
            elif W > 1:
                works.append(_exchange_exact_async(rows[:total], rec_base + [total], counts, r, self.group))
                if scales is not None:
                    works.append(_exchange_exact_async(scales[nh:nh + total], rec_base + [total], counts, r, self.group))
                works.append(_exchange_exact_async(frags[:ftotal].view(-1, 1), frag_off + [ftotal], slots_r, r, self.group))
        else:
            m, ms = total // W, ftotal // W
            s_rows = torch.empty((max(m, 1), pb), dtype=torch.uint8, device=tok.device)
            s_scales = torch.empty((max(m, 1), sb), dtype=torch.uint8, device=tok.device) if sb else None
            s_frag = torch.empty(ms, dtype=torch.int64, device=tok.device)
            t.shard_cols_pack(0, n_me, s_rows, s_scales, s_frag)
            t0 = self._tick("pack_ms", t0)
            if m:
                works.append(_all_gather_async(rows[:total].view(-1), s_rows[:m].reshape(-1), self.group))
                if scales is not None:
                    works.append(_all_gather_async(scales[nh:nh + total].view(-1), s_scales[:m].reshape(-1), self.group))
            works.append(_all_gather_async(frags[:ftotal], s_frag, self.group))
            keep = (s_rows, s_scales, s_frag)
        if self._prof is not None:
            for w in works:
                w.wait()
            t0 = self._tick("collective_ms", t0)
            got = (total - n_me) * (pb + sb) + (ftotal - slots_r[r]) * 8 if (exact or W == 1) else (total * (pb + sb) + ftotal * 8) * (W - 1) // W
            self._prof["bytes_received"] = float(got)
        self._plan_packed_here(tok)
        self._slot_open[slot] = True
        return {"slot": slot, "tok": tok, "C": 1, "per": B, "works": works, "ready": None, "t0": t0, "keep": keep, "sdma": st,
                "cols": {"rows": rows, "scales": scales, "frags": frags, "total": total, "frag_off": frag_off,
                         "frag_slots": slots_r, "rec_base": rec_base}}

    def _gather_begin(self, tok, slot, t0):
        B, T = tok.shape
        W, t = self.world, self.table
        C = max(1, min(self.gather_chunks, B, 64))
        if C == 1 and hasattr(t, "shard_cols_pack"):
            return self._gather_begin_cols(tok, slot, t0)
        per = (B + C - 1) // C
        self._plan_enter(slot, tok)
        ends = self._plan(tok, slot, C, True)                           # synchronises its stream: this rank's record counts
        mine = [ends[0]] + [ends[c] - ends[c - 1] for c in range(1, C)]
        t0 = self._tick("plan_ms", t0)
        exact = W > 1 and self.gather_transport == "p2p"
        if W > 1:
            cnt = torch.tensor(mine, dtype=torch.int64, device=tok.device)
            allc = torch.empty((W, C), dtype=torch.int64, device=tok.device)
            _all_gather(allc.view(-1), cnt, self.group)
            allc = allc.tolist()                                        # host: every rank's count for every chunk
            maxc = [max(allc[r][c] for r in range(W)) for c in range(C)]
        else:
            allc = [list(mine)]
            maxc = list(mine)
        base = [0]
        for c in range(C):
            # exact: contributions back to back; all-gather: every contribution padded to the chunk's largest
            base.append(base[-1] + (sum(allc[r][c] for r in range(W)) if exact else W * int(maxc[c])))
        total = base[-1]
        rec = t.shard_record_bytes()
        full = self._slot_full[slot]
        if full is None or full.shape[0] < max(total, 1) or full.shape[1] != rec or full.device != tok.device:
            # one receive buffer per slot, kept between steps and grown with some room (the counts move a little from
            # batch to batch); a dropped buffer returns to the allocator only after this stream has waited for the
            # slot's last reduction
            full = torch.empty((max(total + total // 8, 1), rec), dtype=torch.uint8, device=tok.device)
            self._slot_full[slot] = full
        t0 = self._tick("collective_ms", t0)
        works = []
        first = 0
        for c in range(C):
            m = int(maxc[c])
            region = full[base[c]:base[c + 1]]
            if m == 0:
                works.append(None)
            elif exact:
                counts = [int(allc[r][c]) for r in range(W)]
                offs = [0]
                for r in range(W):
                    offs.append(offs[-1] + counts[r])
                if mine[c]:
                    t.shard_gather_pack_range(first, mine[c], region[offs[self.rank]:offs[self.rank] + mine[c]])   # in place
                t0 = self._tick("pack_ms", t0)
                work = _exchange_exact_async(region, offs, counts, self.rank, self.group)
                if self._prof is not None:
                    work.wait()
                    t0 = self._tick("collective_ms", t0)
                works.append((work, None))
            elif W > 1:
                send = torch.empty((m, rec), dtype=torch.uint8, device=tok.device)
                t.shard_gather_pack_range(first, mine[c], send)          # my records of chunk c + padding to m
                t0 = self._tick("pack_ms", t0)
                work = _all_gather_async(region.view(-1), send.view(-1), self.group)
                if self._prof is not None:                               # instrumented step: no overlap, phases add up
                    work.wait()
                    t0 = self._tick("collective_ms", t0)
                works.append((work, send))
            else:
                t.shard_gather_pack_range(first, mine[c], region)
                t0 = self._tick("pack_ms", t0)
                works.append(None)
            first += mine[c]
        self._plan_packed_here(tok)
        self._slot_open[slot] = True
        if self._prof is not None:
            self._prof["bytes_received"] = float((total - sum(mine)) * rec if exact else total * rec * (W - 1) // max(W, 1))
        return {"slot": slot, "tok": tok, "C": C, "per": per, "base": base, "total": total, "records": full[:total],
                "works": works, "ready": None, "t0": t0}

    def gather_rows_finish(self, ticket: dict, *, reduce: str = "mean", wte: Optional[torch.Tensor] = None,
                           wpe: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                           out_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None,
                           check: bool = False) -> torch.Tensor:
        """Second half: chunk by chunk, the caller's stream waits for the chunk's records, adds them to the row map and
        reduces the chunk's sequences out of ``[replicated head | records received so far]``.  Returns ``[B, T, d]`` --
        bit-identical to ``EmbeddingCache.embed_tokens`` on the unsharded table.  ``check=True``: read the handle's sticky
        status bits afterwards (synchronises) and raise if a referenced row never arrived (the ranks disagreed about the
        batch), a token was out of range, ...: a serving loop checks every N-th step, a test every step."""
        t, d = self.table, self.embedding_dim
        tok = ticket["tok"]
        B, T = tok.shape
        if out_dtype is None:
            out_dtype = wte.dtype if wte is not None else (wpe.dtype if wpe is not None else torch.float32)
        if position_ids is not None:
            position_ids = position_ids.to(device=tok.device, dtype=torch.int32).expand(B, T).contiguous()
        cur = torch.cuda.current_stream() if tok.is_cuda else None
        if ticket["ready"] is not None:                                  # begun on the side stream
            cur.wait_event(ticket["ready"])
            tok.record_stream(cur)                                       # (allocated there, read here)
        t.shard_select_slot(ticket["slot"])
        self._slot_open[ticket["slot"]] = False
        if out is None:
            out = torch.empty((B * T, d), dtype=out_dtype, device=tok.device)
        else:
            assert out.is_contiguous() and out.dtype == out_dtype and out.numel() == B * T * d
        t0 = ticket["t0"]
        if ticket.get("hdr") is not None:                                # sync-free exchange: did every contribution fit?
            hdr, ev, caps_used = ticket["hdr"]
            if ev is not None:
                ev.synchronize()                                         # (the transfers of this batch are long done in a loop that
            counts = [int(c) for c in hdr.view(-1, 2)[:, 0].tolist()]   #  runs ahead: the host does not wait here)
            self._note_counts(counts)
            if any(c > cap for c, cap in zip(counts, caps_used)):        # every rank reads the same headers: they all repeat
                self.sync_free_stats["overflow_repeats"] += 1
                for w in ticket["works"]:
                    w.wait()                                             # the slot's buffers are about to be reused
                self._slot_open[ticket["slot"]] = False
                redo = self._gather_begin_cols(tok, ticket["slot"], 0.0, exact_only=True)   # exact sizes, on this stream
                ticket = dict(redo, ready=None)
                self._slot_open[ticket["slot"]] = False
        if "cols" in ticket:                                             # columns on the wire: no indexing pass
            for w in ticket["works"]:
                w.wait()                                                 # the current stream waits for the three columns
            t0 = self._tick("collective_ms", t0)
            c = ticket["cols"]
            row_lo = [shard_range(self.n_rows, q, self.world)[0] for q in range(self.world)] + [self.n_rows]   # who owns which rows
            t.shard_cols_embed(tok, 0, B, c["rows"], c["total"], c["scales"], c["frags"], c["frag_off"], c["frag_slots"],
                               c["rec_base"], out, wte=wte, wpe=wpe, position_ids=position_ids, reduce=reduce, row_lo=row_lo)
            t0 = self._tick("embed_ms", t0)
            if cur is not None:
                done = torch.cuda.Event()
                done.record(cur)
                self._slot_done[ticket["slot"]] = done
            if ticket.get("sdma") is not None:                           # the peers may push the next batch into this slot
                t.ipc_event_record(ticket["sdma"]["done"][ticket["sdma"]["cur_ev"]])
                ticket["sdma"]["used"] = True
            self._keep = (ticket, position_ids, wte, wpe, out)
            if check:
                self._check_status("gather_rows_finish")
            return out.view(B, T, d)
        C, per, base, records, works = ticket["C"], ticket["per"], ticket["base"], ticket["records"], ticket["works"]
        for c in range(C):
            if works[c] is not None:
                works[c][0].wait()                                       # the current stream waits for chunk c's records
                t0 = self._tick("collective_ms", t0)
            if c == 0 or base[c + 1] > base[c]:                          # chunk 0 always: it starts the exchange (clears the map)
                t.shard_gather_add_records(records, base[c], base[c + 1] - base[c])
            s0, s1 = min(c * per, B), min(c * per + per, B)
            if s1 > s0:
                t.shard_gather_embed_range(tok, s0, s1, records, out, wte=wte, wpe=wpe, position_ids=position_ids,
                                           reduce=reduce)
            t0 = self._tick("embed_ms", t0)
        if cur is not None:
            done = torch.cuda.Event()
            done.record(cur)
            self._slot_done[ticket["slot"]] = done
        self._keep = (ticket, position_ids, wte, wpe, out)               # read in place until the stream has passed
        if check:
            self._check_status("gather_rows_finish")
        return out.view(B, T, d)

    def _check_status(self, who: str) -> None:
        bits = int(self.table.status()) if hasattr(self.table, "status") else 0
        if bits:
            names = [n for b, n in ((1, "token or position outside wte / wpe"), (2, "a referenced row never arrived / row id outside the table"),
                                    (4, "index full"), (8, "cold-row cache overflow")) if bits & b]
            raise RuntimeError(f"{who}: device status bits {bits:#x} ({'; '.join(names)}) -- the output of this step is not "
                               "the unsharded table's")

    def gather_rows_abandon(self, ticket: Optional[dict]) -> None:
        """Give up a batch begun with :meth:`gather_rows_begin` without reducing it (an exception between begin and finish, a
        ticket the caller drops): waits for its outstanding transfers and frees its plan slot, so that later begins do not
        find the slot occupied.  Collective like begin / finish: every rank abandons the same ticket."""
        if ticket is None:
            return
        for w in ticket.get("works") or []:
            w = w[0] if isinstance(w, tuple) else w
            if w is not None:
                w.wait()
        tok = ticket["tok"]
        if tok.is_cuda:
            cur = torch.cuda.current_stream()
            if ticket.get("ready") is not None:
                cur.wait_event(ticket["ready"])
            done = torch.cuda.Event()
            done.record(cur)
            self._slot_done[ticket["slot"]] = done
            if ticket.get("sdma") is not None:
                self.table.ipc_event_record(ticket["sdma"]["done"][ticket["sdma"]["cur_ev"]])
                ticket["sdma"]["used"] = True
        self._slot_open[ticket["slot"]] = False

    def reset_slots(self) -> None:
        """Forget every batch in flight (after an error, when tickets were lost): the device is synchronised and every plan
        slot is free again.  The peers must do the same before the next exchange."""
        if torch.cuda.is_available() and getattr(self.table, "device", None) is not None and torch.device(self.table.device).type == "cuda":
            torch.cuda.synchronize(self.table.device)
        self._slot_open = [False] * 4
        self._slot_done = [None] * 4
        self._slot_next = 0
        self._plan_packed = None
        if hasattr(self.table, "shard_select_slot"):
            self.table.shard_select_slot(0)

    def _embed_gather_rows_unchunked(self, tok, reduce, wte, wpe, position_ids, out_dtype):
        """One plan, one all-gather, one reduction (the form the chunked path degenerates to with one chunk)."""
        B, T = tok.shape
        W = self.world
        # one record per DISTINCT row I own (outside the replicated head) that the batch references
        send = self.table.shard_gather_pack(self.table.shard_gather_plan(tok))   # uint8 [c_me, record_bytes]
        rec = send.shape[1]
        if W > 1:
            mine = torch.tensor([send.shape[0]], dtype=torch.int64, device=send.device)
            allc = torch.empty(W, dtype=torch.int64, device=send.device)
            _all_gather(allc, mine, self.group)
            counts = [int(c) for c in allc.tolist()]
            maxc = max(counts)
            if maxc:
                padded = torch.empty((maxc, rec), dtype=torch.uint8, device=send.device)
                padded[:send.shape[0]] = send
                full = torch.empty((W * maxc, rec), dtype=torch.uint8, device=send.device)
                _all_gather(full, padded, self.group)
                recv = torch.cat([full[r * maxc:r * maxc + counts[r]] for r in range(W)])
            else:
                recv = send
        else:
            recv = send
        return self.table.shard_gather_embed(tok, recv, wte=wte, wpe=wpe, position_ids=position_ids, reduce=reduce,
                                             out_dtype=out_dtype)

    def _embed_row_exchange_dedup(self, tok, reduce, wte, wpe, position_ids, out_dtype, gather_output):
        """The slice exchange with one record per DISTINCT row and destination (the first form sent one per reference: an
        f-gram row covering three tokens of a slice crossed three times, 0.97M records instead of 0.43M on the C5-shaped
        batch).  Built from the chunked-gather primitives with chunk q = the sequences rank q finalises and a claim
        generation per chunk: plan (one match of the batch + W claim passes) -> my records for every destination, already
        grouped by destination in the list; the W x W record counts are exchanged (one tiny all-gather); ONE pack, ONE
        all_to_all_single; the receiver indexes what arrived by row id and reduces its slice.  Bit-identical to the
        unsharded table."""
        import time
        t0 = time.perf_counter() if self._prof is not None else 0.0
        B, T = tok.shape
        d, W, r, t = self.embedding_dim, self.world, self.rank, self.table
        tok = t._tok(tok)
        if position_ids is not None:
            position_ids = position_ids.to(device=tok.device, dtype=torch.int32).expand(B, T).contiguous()
        bper = (B + W - 1) // W
        self._plan_enter(0, tok)                                                 # one-call form: slot 0, the caller's stream
        ends = self._plan(tok, 0, W, False)                                      # chunk q = slice q (ceil(B / W) sequences)
        send_counts = [ends[0]] + [ends[q] - ends[q - 1] for q in range(1, W)]
        t0 = self._tick("plan_ms", t0)
        rec = t.shard_record_bytes()
        if W > 1:
            mine = torch.tensor(send_counts, dtype=torch.int64, device=tok.device)
            allc = torch.empty((W, W), dtype=torch.int64, device=tok.device)
            _all_gather(allc.view(-1), mine, self.group)
            recv_counts = allc[:, r].tolist()                                        # what every owner sends to my slice
        else:
            recv_counts = list(send_counts)
        t0 = self._tick("collective_ms", t0)
        n_send, n_recv = int(sum(send_counts)), int(sum(recv_counts))
        send = torch.empty((max(n_send, 1), rec), dtype=torch.uint8, device=tok.device)
        if n_send:
            t.shard_gather_pack_range(0, n_send, send[:n_send])
        self._plan_packed_here(tok)
        t0 = self._tick("pack_ms", t0)
        if W > 1:
            recv = torch.empty((max(n_recv, 1), rec), dtype=torch.uint8, device=tok.device)
            _all_to_all(recv[:n_recv], send[:n_send], [int(c) for c in recv_counts], [int(c) for c in send_counts], self.group)
        else:
            recv = send
        t0 = self._tick("collective_ms", t0)
        if self._prof is not None:
            self._prof["bytes_received"] = float((n_recv - int(recv_counts[r])) * rec)
        b0, b1 = min(r * bper, B), min(r * bper + bper, B)
        out_slice = torch.zeros((bper * T, d), dtype=out_dtype, device=tok.device) if (b1 - b0) < bper else \
            torch.empty((bper * T, d), dtype=out_dtype, device=tok.device)
        records = recv[:n_recv]
        t.shard_gather_add_records(records, 0, n_recv)
        if b1 > b0:
            t.shard_gather_embed_range(tok, b0, b1, records, out_slice, wte=wte, wpe=wpe, position_ids=position_ids,
                                       reduce=reduce, out_is_slice=True)
        self._keep = (recv, send)
        t0 = self._tick("embed_ms", t0)
        if not gather_output:
            return out_slice
        if W > 1:
            full = torch.empty((bper * T * W, d), dtype=out_dtype, device=tok.device)
            _all_gather(full, out_slice, self.group)
            if self._prof is not None:
                self._prof["bytes_received"] += float(full.numel() * full.element_size() * (W - 1) // W)
        else:
            full = out_slice
        self._tick("gather_out_ms", t0)
        return full[:B * T].reshape(B, T, d)
