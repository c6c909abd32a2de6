"""CPU, world_size 2, gloo: the row-sharded exchange (reduce-scatter of partial sums ->
finalise 1/W of the tokens -> all-gather) against the single-table oracle.  The device
handle is replaced by an oracle-backed stand-in (test infrastructure) so that the host
logic -- shard ranges, padding, slice bookkeeping, collective pattern -- runs here."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleShard:
    """Stand-in for hip_backend.SconeTable on a shard: same embed_partial / finalize contract,
    computed with oracle/ref_port.py."""

    def __init__(self, keys, lens, max_n, table, row_begin, row_end):
        from oracle import ref_port as R
        self.R, self.keys, self.lens, self.max_n = R, keys, lens, max_n
        self.table, self.row_begin, self.row_end = table, row_begin, row_end
        self.n_rows, self.dim = table.shape[0], table.shape[1]
        self.n_head, self.stored = 0, []

    # ---- ingestion stand-ins: record what ShardedEmbeddingCache.load_rows stores where
    def store_f32(self, rows, row0=0, ids=None):
        self.stored.append(("owned", int(row0), int(rows.shape[0])))
        assert np.array_equal(rows.numpy(), self.table[row0:row0 + rows.shape[0]])

    def shard_set_head(self, n_head):
        self.n_head = int(n_head)

    def shard_head_store_f32(self, rows, row0=0):
        self.stored.append(("head", int(row0), int(rows.shape[0])))
        assert row0 + rows.shape[0] <= self.n_head
        assert np.array_equal(rows.numpy(), self.table[row0:row0 + rows.shape[0]])

    def embed_partial(self, tok):
        R = self.R
        t = tok.numpy()
        off, ids = R.hits_to_csr(R.match_hits(self.keys, self.lens, t, self.max_n))
        counts = np.diff(off).astype(np.int32)
        owned = (ids >= self.row_begin) & (ids < self.row_end)
        # drop the ids of other shards, keep order
        seg = np.repeat(np.arange(len(counts)), counts)
        off_own = np.zeros(len(counts) + 1, dtype=np.int64)
        np.cumsum(np.bincount(seg[owned], minlength=len(counts)), out=off_own[1:])
        partial = R.embed_numpy(self.table, off_own, ids[owned], "sum")
        return torch.from_numpy(partial), torch.from_numpy(counts)

    def _refs(self, tok):
        R = self.R
        t = tok.numpy()
        off, ids = R.hits_to_csr(R.match_hits(self.keys, self.lens, t, self.max_n))
        counts = np.diff(off)
        tix = np.repeat(np.arange(len(counts)), counts)
        jix = np.arange(len(ids)) - np.repeat(off[:-1], counts)
        return off, ids, tix, jix

    # ---- row exchange stand-ins (same contract as SconeTable.shard_gather_*): a record = [fp32 row | int32 row id | int32 marker]
    def shard_gather_plan(self, tok):
        off, ids, tix, jix = self._refs(tok)
        mine = np.unique(ids[(ids >= max(self.row_begin, self.n_head)) & (ids < self.row_end)])
        self._uniq = mine
        return len(mine)

    def shard_gather_pack(self, n_records):
        assert n_records == len(self._uniq)
        n = self.dim * 4 + 8
        recs = [np.concatenate([self.table[i].view(np.uint8), np.array([i, -1], dtype=np.int32).view(np.uint8)])
                for i in self._uniq[::-1]]                           # any order is allowed
        return torch.from_numpy(np.stack(recs) if recs else np.zeros((0, n), dtype=np.uint8))

    def shard_gather_embed(self, tok, records, wte=None, wpe=None, position_ids=None, reduce="mean",
                           out_dtype=torch.float32, out=None):
        B, T = tok.shape
        off, ids, tix, jix = self._refs(tok)
        r = records.numpy()
        by_id = {int(rec[self.dim * 4:].view(np.int32)[0]): rec[:self.dim * 4].view(np.float32) for rec in r}
        assert len(by_id) == len(r), "a row arrived twice"
        rows = np.zeros((len(ids), self.dim), dtype=np.float32)
        for k, i in enumerate(ids.tolist()):
            rows[k] = self.table[i] if i < self.n_head else by_id[i]   # head rows are local everywhere
        assert np.array_equal(rows, self.table[ids])
        x = torch.from_numpy(self.R.embed_numpy(rows, off, np.arange(len(ids)), reduce))
        flat = tok.reshape(-1).long()
        if wte is not None:
            x = wte.float()[flat] + x
        if wpe is not None:
            pos = (torch.arange(flat.numel()) % T) if position_ids is None else position_ids.reshape(-1).long()
            x = x + wpe.float()[pos]
        return x.to(out_dtype)

    # ---- chunked all-gather form (scone_shard_gather_plan_chunks / _pack_range / _add_records / _embed_range)
    def _tok(self, tok):
        return tok

    def shard_select_slot(self, slot):
        # the receiver-side state of a planned batch exists twice (scone_shard_select_slot)
        st = self.__dict__.setdefault("_slots", {0: {}, 1: {}, 2: {}, 3: {}})
        cur = self.__dict__.setdefault("_slot", 0)
        if slot == cur:
            return
        st[cur] = {k: self.__dict__.pop(k) for k in ("_planned", "_by_id", "_added", "_planned_ell") if k in self.__dict__}
        self.__dict__.update(st[slot])
        self._slot = slot

    def shard_record_bytes(self):
        return self.dim * 4 + 8

    def shard_gather_plan_chunks(self, tok, n_chunks, dedup_across_chunks=True):
        B, T = tok.shape
        per = (B + n_chunks - 1) // n_chunks
        claimed, order, ends = set(), [], []
        for c in range(n_chunks):
            if not dedup_across_chunks:
                claimed = set()                                       # every chunk (= destination) gets its own copy
            s0, s1 = min(c * per, B), min(c * per + per, B)
            if s1 > s0:
                off, ids, tix, jix = self._refs(tok[s0:s1])
                for i in np.unique(ids[(ids >= max(self.row_begin, self.n_head)) & (ids < self.row_end)])[::-1].tolist():
                    if i not in claimed:                              # claimed by an earlier chunk: not sent again
                        claimed.add(i)
                        order.append(i)
            ends.append(len(order))
        self._uniq, self._planned = np.asarray(order, dtype=np.int64), tuple(tok.shape)
        self._planned_ell = None
        return ends

    # ---- the match of a plan, sharded over the ranks (scone_shard_gather_match / _plan_ell): list records as the device
    #      writes them -- ids in the reference's order, -1 padding, K_own | K_full << 8 in word W - 2
    def ell_width(self):
        return 8 if self.max_n <= 3 else 16

    def shard_gather_match(self, tok, seq_begin, seq_end, out_ell):
        W = self.ell_width()
        off, ids, tix, jix = self._refs(tok[seq_begin:seq_end])
        rec = np.full((len(off) - 1, W), -1, dtype=np.int32)
        rec[tix, jix] = ids
        k = np.diff(off).astype(np.int32)
        rec[:, W - 2] = k | (k << 8)
        rec[:, W - 1] = 0
        out_ell.numpy()[:rec.shape[0]] = rec

    def shard_gather_plan_ell(self, ell, B, T, n_chunks, dedup_across_chunks=True):
        # the claim passes over the GATHERED lists (no match of its own: what the all-gather delivered is what counts)
        W = self.ell_width()
        e = ell.numpy()[:B * T].reshape(B, T, W)
        per = (B + n_chunks - 1) // n_chunks
        claimed, order, ends = set(), [], []
        for c in range(n_chunks):
            if not dedup_across_chunks:
                claimed = set()
            s0, s1 = min(c * per, B), min(c * per + per, B)
            if s1 > s0:
                r = e[s0:s1].reshape(-1, W)
                k = r[:, W - 2] & 0xFF
                ids = np.concatenate([r[i, :k[i]] for i in range(r.shape[0])]) if r.shape[0] else np.zeros(0, dtype=np.int32)
                for i in np.unique(ids[(ids >= max(self.row_begin, self.n_head)) & (ids < self.row_end)])[::-1].tolist():
                    if i not in claimed:
                        claimed.add(i)
                        order.append(i)
            ends.append(len(order))
        self._uniq, self._planned = np.asarray(order, dtype=np.int64), (B, T)
        self._planned_ell = e.copy()
        self.plans_from_gathered_lists = getattr(self, "plans_from_gathered_lists", 0) + 1
        return ends

    # ---- columns on the wire (scone_shard_cols_*): payload rows | (no scales: fp32 rows) | the sender's fragment.  The
    #      stand-in's fragment is a list, not a hash table: entry k = (row id + 1) << 32 | position, zeros behind
    def payload_bytes(self):
        return self.dim * 4

    def scale_bytes(self):
        return 0

    @staticmethod
    def cols_frag_slots(count):
        s = 64
        while s < 4 * count:
            s <<= 1
        return s

    def shard_head_scales_into(self, scales_full):
        raise AssertionError("fp32 rows have no scales")

    def shard_cols_pack(self, first, count, rows_out, scales_out, frag_out):
        assert first + count <= len(self._uniq) and scales_out is None and frag_out.numel() >= self.cols_frag_slots(count)
        ro, fo = rows_out.numpy(), frag_out.numpy()
        fo[:] = 0
        for k, i in enumerate(self._uniq[first:first + count]):
            ro[k] = self.table[i].view(np.uint8)
            fo[k] = ((int(i) + 1) << 32) | k
        self.cols_packs = getattr(self, "cols_packs", 0) + 1

    # ---- the sync-free plan (scone_shard_gather_plan*_async / scone_shard_cols_pack_cap): the count is not returned; the pack
    #      clips to the capacity and writes the header (rows claimed, overflow flag)
    def shard_gather_plan_async(self, tok):
        self.shard_gather_plan_chunks(tok, 1)

    def shard_gather_plan_ell_async(self, ell, B, T):
        self.shard_gather_plan_ell(ell, B, T, 1)

    def shard_cols_pack_cap(self, cap_rows, rows_out, scales_out, frag_out, header_out):
        n = len(self._uniq)
        k = min(n, int(cap_rows))
        assert scales_out is None and frag_out.numel() == self.cols_frag_slots(cap_rows) and header_out.numel() == 2
        self.shard_cols_pack(0, k, rows_out, None, frag_out)
        header_out.numpy()[:] = (n, int(n > cap_rows))
        self.cap_packs = getattr(self, "cap_packs", 0) + 1

    def shard_cols_embed(self, tok, seq_begin, seq_end, rows, n_total, scales_full, frags, frag_off, frag_slots, rec_base, out,
                         wte=None, wpe=None, position_ids=None, reduce="mean", row_lo=None):
        from scone_amd.distributed import owner_of
        assert tuple(tok.shape) == self._planned and scales_full is None
        assert row_lo is None or (len(row_lo) == len(frag_off) + 1 and row_lo[0] == 0 and list(row_lo) == sorted(row_lo))
        B, T = tok.shape
        sl = tok[seq_begin:seq_end]
        off, ids, tix, jix = self._refs(sl)
        f, r = frags.numpy(), rows.numpy()
        world = len(frag_off)
        by_id = {}
        for q in range(world):
            for v in f[frag_off[q]:frag_off[q] + frag_slots[q]].tolist():
                if v:
                    i, pos = (v >> 32) - 1, v & 0xFFFFFFFF
                    assert int(owner_of(torch.tensor([i]), self.n_rows, world)[0]) == q, "a row in another owner's fragment"
                    assert i not in by_id, "a row arrived twice"
                    by_id[i] = rec_base[q] + pos
        got = np.zeros((len(ids), self.dim), dtype=np.float32)
        for k, i in enumerate(ids.tolist()):                          # KeyError = a needed row did not arrive
            if i < self.n_head:
                got[k] = self.table[i]
            else:
                assert by_id[i] < n_total
                got[k] = r[by_id[i], :self.dim * 4].view(np.float32)
        assert np.array_equal(got, self.table[ids])
        x = torch.from_numpy(self.R.embed_numpy(got, off, np.arange(len(ids)), reduce))
        flat = sl.reshape(-1).long()
        if wte is not None:
            x = wte.float()[flat] + x
        if wpe is not None:
            pos = (torch.arange(flat.numel()) % T) if position_ids is None else position_ids[seq_begin:seq_end].reshape(-1).long()
            x = x + wpe.float()[pos]
        out.view(B * T, self.dim)[seq_begin * T:seq_end * T] = x.to(out.dtype)

    def shard_gather_pack_range(self, first, count, out):
        assert first + count <= len(self._uniq) and out.shape[1] == self.dim * 4 + 8
        o = out.numpy()
        for k, i in enumerate(self._uniq[first:first + count]):
            o[k, :self.dim * 4] = self.table[i].view(np.uint8)
            o[k, self.dim * 4:] = np.array([i, -1], dtype=np.int32).view(np.uint8)
        o[count:] = 0xAB                                              # stale bytes ...
        o[count:, self.dim * 4:] = 0xFF                               # ... marked as padding (row id 0xFFFFFFFF)

    def shard_gather_add_records(self, records, record0, n_records):
        if record0 == 0:
            self._by_id, self._added = {}, 0
        if n_records == 0:
            return
        assert record0 == self._added, "records must be added in order"
        r = records.numpy()
        for p in range(record0, record0 + n_records):
            i = int(r[p, self.dim * 4:].view(np.int32)[0])
            if i == -1:
                continue                                              # padding
            assert i not in self._by_id, "a row arrived twice"
            self._by_id[i] = p
        self._added = record0 + n_records

    def shard_gather_embed_range(self, tok, seq_begin, seq_end, records, out, wte=None, wpe=None, position_ids=None,
                                 reduce="mean", out_is_slice=False):
        assert tuple(tok.shape) == self._planned
        B, T = tok.shape
        sl = tok[seq_begin:seq_end]
        off, ids, tix, jix = self._refs(sl)
        if getattr(self, "_planned_ell", None) is not None:            # sharded match: the gathered lists ARE these lists
            e = self._planned_ell[seq_begin:seq_end].reshape(-1, self.ell_width())
            assert np.array_equal(e[:, self.ell_width() - 2] & 0xFF, np.diff(off)) and np.array_equal(e[tix, jix], ids)
        r = records.numpy()
        rows = np.zeros((len(ids), self.dim), dtype=np.float32)
        for k, i in enumerate(ids.tolist()):                          # KeyError = a needed row has not arrived yet
            rows[k] = self.table[i] if i < self.n_head else r[self._by_id[i], :self.dim * 4].view(np.float32)
        assert np.array_equal(rows, self.table[ids])
        x = torch.from_numpy(self.R.embed_numpy(rows, off, np.arange(len(ids)), reduce))
        flat = sl.reshape(-1).long()
        if wte is not None:
            x = wte.float()[flat] + x
        if wpe is not None:
            pos = (torch.arange(flat.numel()) % T) if position_ids is None else position_ids[seq_begin:seq_end].reshape(-1).long()
            x = x + wpe.float()[pos]
        if out_is_slice:
            out.view(-1, self.dim)[:(seq_end - seq_begin) * T] = x.to(out.dtype)
        else:
            out.view(B * T, self.dim)[seq_begin * T:seq_end * T] = x.to(out.dtype)

    def finalize(self, sums, counts, tok, a, b, wte=None, wpe=None, position_ids=None, reduce="mean",
                 out_dtype=torch.float32, out=None):
        x = sums.clone()
        if reduce == "mean":
            k = counts.to(torch.float32).clamp(min=1).unsqueeze(1)
            x = torch.where(counts.unsqueeze(1) > 1, x / k, x)
        B, T = tok.shape
        flat = tok.reshape(-1)[a:b].long()
        if wte is not None:
            x = wte.float()[flat] + x
        if wpe is not None:
            pos = (torch.arange(B * T) % T)[a:b] if position_ids is None else position_ids.reshape(-1)[a:b].long()
            x = x + wpe.float()[pos]
        x = x.to(out_dtype)
        if out is not None:
            out.copy_(x)
            return out
        return x


def _worker(rank, world, port, ntok_shape, out_dtype_name, exchange, q, chunks=4, head=0, transport="p2p", shard_match="auto"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_port as R
        from scone_amd import NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache, shard_range
        rng = np.random.default_rng(42)
        vocab, n, d, max_n = 19, 301, 32, 3
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = rng.standard_normal((n, d)).astype(np.float32)
        B, T = ntok_shape
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32))
        wpe = torch.from_numpy(rng.standard_normal((T, d)).astype(np.float32))
        out_dtype = getattr(torch, out_dtype_name)
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        a, b = shard_range(n, rank, world)
        shard = OracleShard(keys, lens, max_n, table, a, b)
        if chunks == 0:                                               # the one-shot form (plan / pack / embed)
            del OracleShard.shard_gather_plan_chunks                 # (this worker process only)
        if head:
            shard.shard_set_head(head)
        cache = ShardedEmbeddingCache(ex, d, rank=rank, world=world, n_rows=n, table=shard, replicated_rows=head,
                                      gather_chunks=max(chunks, 1), gather_transport=transport, shard_match=shard_match)
        assert (cache.row_begin, cache.row_end) == (a, b)
        if transport == "sdma":        # no scone_ipc_* on a stand-in handle: EVERY rank falls back to p2p, and says why
            assert cache.gather_transport == "p2p" and "stand-in" in cache.transport_fallback_reason, cache.transport_fallback_reason
        out = cache.embed_tokens(tok, wte=wte.to(out_dtype), wpe=wpe.to(out_dtype), out_dtype=out_dtype,
                                 exchange=exchange, check=True)
        ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok.numpy(), max_n))
        fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(B, T, d))
        ref = R.combine(tok, fg, wte.to(out_dtype).float(), wpe.to(out_dtype).float())
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        # the slice-only form returns this rank's finished tokens
        sl = cache.embed_tokens(tok, wte=wte.to(out_dtype), wpe=wpe.to(out_dtype), out_dtype=out_dtype,
                                gather_output=False, exchange=exchange)
        per = (B * T + world - 1) // world if exchange == "partial_sums" else ((B + world - 1) // world) * T
        lo, hi = min(rank * per, B * T), min(rank * per + per, B * T)
        ok_slice = torch.equal(sl[:hi - lo], out.reshape(-1, d)[lo:hi])
        if shard_match is True and world > 1 and B >= world and exchange in ("rows", "gather_rows") and chunks > 0:
            ok_slice = ok_slice and getattr(shard, "plans_from_gathered_lists", 0) == 2     # both calls planned from gathered lists
        elif shard_match == "auto":
            ok_slice = ok_slice and not hasattr(shard, "plans_from_gathered_lists")         # (tiny batches: every rank matches)
        if exchange == "gather_rows":                                                        # one piece = columns, the chunk pipeline = records
            ok_slice = ok_slice and (getattr(shard, "cols_packs", 0) == 2) == (chunks > 0 and min(chunks, B) == 1)   # (a batch of one
            ok_slice = ok_slice and (chunks == 0 or cache.wire_format == ("columns" if chunks == 1 else "records"))  # sequence has one chunk)
        q.put((rank, err, tuple(out.shape), ok_slice))
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent
        q.put((rank, repr(e), None, False))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exchange", ["rows", "gather_rows", "partial_sums"])
@pytest.mark.parametrize("shape,dtype", [((3, 17), "float32"), ((2, 8), "float16"), ((1, 5), "float32")])
def test_sharded_exchange_world2_gloo(shape, dtype, exchange):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, dtype, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, out_shape, ok_slice in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert out_shape == (shape[0], shape[1], 32)
        assert err < (1e-6 if dtype == "float32" else 2e-3), (rank, err)
        assert ok_slice


@pytest.mark.parametrize("chunks,head,transport", [(0, 0, "p2p"), (1, 0, "p2p"), (2, 40, "p2p"), (3, 0, "p2p"), (8, 25, "p2p"),
                                                   (1, 0, "all_gather"), (2, 40, "all_gather"), (8, 25, "all_gather")])
def test_gather_rows_chunked_pipeline_world2_gloo(chunks, head, transport):
    """The pipelined all-gather form: one plan with a claim pass per chunk (a row sent by an earlier chunk is not sent
    again); the records of a chunk travel as exact point-to-point ranges (``p2p``: every rank's contribution at its own
    size) or through per-chunk all-gathers padded to the largest contribution (``all_gather``: padding records skipped);
    records added and sequences reduced chunk by chunk -- same output as the unsharded table for any chunk count (0 = the
    one-shot form), with and without a replicated head."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    shape = (5, 13)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, "float32", "gather_rows", q, chunks, head, transport))
             for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, out_shape, ok_slice in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert out_shape == (shape[0], shape[1], 32) and err < 1e-6 and ok_slice


@pytest.mark.parametrize("exchange,chunks,head,world,shape", [("gather_rows", 1, 0, 2, (5, 13)), ("gather_rows", 3, 25, 3, (4, 19)),
                                                               ("gather_rows", 2, 40, 2, (2, 8)), ("rows", 4, 25, 3, (7, 9)),
                                                               ("rows", 4, 0, 2, (1, 5)), ("gather_rows", 2, 10, 3, (2, 7))])
def test_match_sharded_over_the_ranks_world_gloo(exchange, chunks, head, world, shape):
    """The plan's match sharded over the ranks: rank r matches only its own run of sequences (`shard_gather_match`), the
    list records are all-gathered, and the claim passes, the reduction and the slice exchange work from the GATHERED lists
    (`shard_gather_plan_ell`) -- same output as the unsharded table, B not a multiple of W, fewer sequences than ranks
    (falls back to matching everything everywhere), chunked, with and without head."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, "float32", exchange, q, chunks, head, "p2p", True))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, out_shape, ok_slice in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert out_shape == (shape[0], shape[1], 32) and err < 1e-6 and ok_slice


@pytest.mark.parametrize("transport,world,head,shape", [("p2p", 2, 0, (5, 13)), ("p2p", 3, 20, (4, 19)),
                                                        ("all_gather", 2, 40, (5, 13)), ("all_gather", 3, 0, (7, 9)),
                                                        ("p2p", 3, 20, (1, 5)), ("sdma", 2, 10, (5, 13))])
def test_gather_rows_columns_on_the_wire_world_gloo(transport, world, head, shape):
    """``gather_rows`` in one piece with COLUMNS on the wire: every rank's payload rows, (scales,) and hash fragment travel as
    three ranges -- exact point-to-point ranges or three padded all-gathers -- and the receiver resolves the id lists through
    the owners' fragments: same output as the unsharded table (the stand-in checks that every row sits in ITS owner's
    fragment, arrives once and inside the receive buffer).  (``gather_chunks > 1`` -- the legacy chunk pipeline -- keeps the
    record form: ``test_gather_rows_chunked_pipeline_world2_gloo``.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, "float32", "gather_rows", q, 1, head, transport, "auto"))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, out_shape, ok_slice in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert out_shape == (shape[0], shape[1], 32) and err < 1e-6 and ok_slice


@pytest.mark.parametrize("transport", ["p2p", "all_gather"])
def test_gather_rows_world3_gloo(transport):
    """Three ranks: with ``p2p`` every rank sends its records to two peers and receives two ranges of different sizes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    shape = (4, 19)
    procs = [ctx.Process(target=_worker, args=(r, 3, port, shape, "float32", "gather_rows", q, 2, 20, transport)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, out_shape, ok_slice in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert out_shape == (shape[0], shape[1], 32) and err < 1e-6 and ok_slice


def _worker_split_phase(rank, world, port, q, slots=2, chunks=2):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_port as R
        from scone_amd import NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache, shard_range
        rng = np.random.default_rng(7)
        vocab, n, d, max_n = 19, 301, 32, 3
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = rng.standard_normal((n, d)).astype(np.float32)
        B, T = 4, 11
        batches = [torch.from_numpy(rng.integers(0, vocab, size=(B, T))) for _ in range(4)]
        wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32))
        wpe = torch.from_numpy(rng.standard_normal((T, d)).astype(np.float32))
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        a, b = shard_range(n, rank, world)
        shard = OracleShard(keys, lens, max_n, table, a, b)
        shard.shard_set_head(30)
        cache = ShardedEmbeddingCache(ex, d, rank=rank, world=world, n_rows=n, table=shard, replicated_rows=30, gather_chunks=chunks,
                                      shard_match=True, plan_slots=slots)   # every slot plans from gathered lists: each needs its own buffer
        outs, tickets, nxt = [], [cache.gather_rows_begin(batches[0])], 1
        for i in range(len(batches)):
            while nxt < len(batches) and len(tickets) < slots:        # batches i+1 (.. i+slots-1) planned and exchanged
                tickets.append(cache.gather_rows_begin(batches[nxt]))     # before batch i is reduced
                nxt += 1
            outs.append(cache.gather_rows_finish(tickets.pop(0), wte=wte, wpe=wpe))
        ts = []
        with pytest.raises(RuntimeError):                              # more batches in flight than slots: refused, not corrupted
            for _ in range(slots + 1):
                ts.append(cache.gather_rows_begin(batches[0]))
        assert len(ts) == slots
        for tk in ts:                                                  # (finished in order: the slots are free again)
            assert torch.equal(cache.gather_rows_finish(tk, wte=wte, wpe=wpe), outs[0])
        worst = 0.0
        for tok, out in zip(batches, outs):
            ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok.numpy(), max_n))
            fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(B, T, d))
            ref = R.combine(tok, fg, wte, wpe)
            worst = max(worst, float((out.float() - ref).abs().max() / ref.abs().max()))
        same = torch.equal(outs[1], cache.embed_tokens(batches[1], wte=wte, wpe=wpe, exchange="gather_rows"))
        q.put((rank, worst, same))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, repr(e) + traceback.format_exc(), False))


def _worker_sync_free(rank, world, port, q, transport, sync_free, slots):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_port as R
        from scone_amd import NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache, shard_range
        rng = np.random.default_rng(23)
        vocab, n, d, max_n = 23, 900, 16, 3
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = rng.standard_normal((n, d)).astype(np.float32)
        shapes = [(3, 7), (3, 7), (3, 8), (24, 31), (24, 31), (2, 5), (30, 33), (3, 7)]     # small, small, ..., BIG (overflows), ...
        batches = [torch.from_numpy(rng.integers(0, vocab, size=s)) for s in shapes]
        wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32))
        wpe = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32))
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        a, b = shard_range(n, rank, world)
        shard = OracleShard(keys, lens, max_n, table, a, b)
        shard.shard_set_head(20)
        cache = ShardedEmbeddingCache(ex, d, rank=rank, world=world, n_rows=n, table=shard, replicated_rows=20, gather_transport=transport,
                                      shard_match=True, plan_slots=slots, sync_free_plan=sync_free)
        outs, tickets, nxt = [], [], 0
        for _ in range(slots - 1):
            tickets.append(cache.gather_rows_begin(batches[nxt]))
            nxt += 1
        for i in range(len(batches)):
            outs.append(cache.gather_rows_finish(tickets.pop(0), wte=wte, wpe=wpe, check=True))
            if nxt < len(batches):
                tickets.append(cache.gather_rows_begin(batches[nxt]))
                nxt += 1
        worst = 0.0
        for tok, out in zip(batches, outs):
            B, T = tok.shape
            ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok.numpy(), max_n))
            fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(B, T, d))
            worst = max(worst, float((out.float() - R.combine(tok, fg, wte, wpe[:T])).abs().max()))
        st = cache.sync_free_stats
        ok = (st["exchanges"] >= len(batches) - slots and st["overflow_repeats"] >= 1 and getattr(shard, "cap_packs", 0) == st["exchanges"]) \
            if sync_free else (st["exchanges"] == 0 and not hasattr(shard, "cap_packs"))
        # the one-call form goes the same way (slot 0)
        again = torch.equal(outs[3], cache.embed_tokens(batches[3], wte=wte, wpe=wpe, exchange="gather_rows", check=True))
        q.put((rank, worst, bool(ok and again), dict(st)))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, repr(e) + traceback.format_exc(), False, None))


@pytest.mark.parametrize("world,transport,sync_free,slots", [(2, "p2p", True, 2), (3, "all_gather", True, 3), (3, "p2p", True, 2), (2, "p2p", False, 2)])
def test_sync_free_plan_sizes_from_previous_batches_and_repeats_an_overflow_world_gloo(world, transport, sync_free, slots):
    """Round 4: after a first batch with exact sizes, the one-piece exchange sizes its transfers from the previous batches
    (+ 12.5 %), plans and packs with the count on the device (stand-in: `shard_cols_pack_cap`), ships every rank's count behind
    its fragment and checks the counts when the batch is reduced.  Batches that grow tenfold overflow their capacity: they are
    repeated with exact sizes inside gather_rows_finish, on every rank alike -- every output equals the oracle's, whatever
    the order of small and big batches, for exact ranges and for padded all-gathers; `sync_free_plan=False` never takes the path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sync_free, args=(r, world, port, q, transport, sync_free, slots)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, worst, ok, st in results:
        assert isinstance(worst, float), f"rank {rank} failed: {worst}"
        assert worst < 1e-5 and ok, (rank, worst, st)


@pytest.mark.parametrize("slots,chunks", [(2, 2), (3, 2), (2, 1), (3, 1)])          # chunks = 1: columns on the wire
def test_gather_rows_split_phase_two_batches_in_flight_world2_gloo(slots, chunks):
    """gather_rows_begin / gather_rows_finish with the next batch begun (planned, packed, gathered) BEFORE the current one is
    reduced: the two plan slots keep the batches apart; every output equals the unsharded lookup of ITS batch and the
    one-call form."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split_phase, args=(r, 2, port, q, slots, chunks)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, same in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert err < 1e-6 and same


def _worker_soak(rank, world, port, slots, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_port as R
        from scone_amd import NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache, shard_range
        rng = np.random.default_rng(11)
        vocab, n, d, max_n, head = 13, 240, 16, 3, 25
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = rng.standard_normal((n, d)).astype(np.float32)
        wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32))
        wpe = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32))
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        a, b = shard_range(n, rank, world)
        shard = OracleShard(keys, lens, max_n, table, a, b)
        shard.shard_set_head(head)
        cache = ShardedEmbeddingCache(ex, d, rank=rank, world=world, n_rows=n, table=shard, replicated_rows=head, plan_slots=slots)
        tickets, batches, worst = [], [], 0.0
        steps = 24
        for i in range(steps + slots - 1):
            if i < steps:
                B, T = int(rng.integers(1, 8)), int(rng.integers(1, 40))
                tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
                cache.gather_chunks = int(rng.integers(1, 4))
                cache.shard_match = bool(rng.integers(2))
                cache.gather_transport = ("p2p", "all_gather")[int(rng.integers(2))]
                tickets.append(cache.gather_rows_begin(tok))
                batches.append(tok)
            if i >= slots - 1:
                k = i - (slots - 1)
                out = cache.gather_rows_finish(tickets[k], wte=wte, wpe=wpe)
                tok = batches[k]
                ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok.numpy(), max_n))
                fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(tok.shape[0], tok.shape[1], d))
                ref = R.combine(tok, fg, wte, wpe)
                worst = max(worst, float((out.float() - ref).abs().max() / ref.abs().max()))
        q.put((rank, worst))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("world,slots", [(2, 2), (3, 3), (2, 4)])
def test_split_phase_soak_random_forms_world_gloo(world, slots):
    """24 batches of random shape (fewer sequences than ranks included) through the split-phase loop with `slots` batches in
    flight, every batch with its own form: one piece (columns or records on the wire) or 2-3 chunks, match sharded over the
    ranks or not, exact ranges or padded all-gathers -- every output equals the oracle's for ITS batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_soak, args=(r, world, port, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in results:
        assert isinstance(err, float), f"rank {rank} failed: {err}"
        assert err < 1e-6


def test_gather_transport_is_validated():
    from scone_amd import NGramExtractor
    from scone_amd.distributed import ShardedEmbeddingCache
    keys = np.array([[1, 0, 0], [2, 3, 0]], dtype=np.uint32)
    lens = np.array([1, 2], dtype=np.uint8)
    ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
    shard = OracleShard(keys, lens, 3, np.zeros((2, 8), dtype=np.float32), 0, 2)
    with pytest.raises(ValueError):
        ShardedEmbeddingCache(ex, 8, rank=0, world=1, n_rows=2, table=shard, gather_transport="ring")
    c = ShardedEmbeddingCache(ex, 8, rank=0, world=1, n_rows=2, table=shard, gather_transport="all_gather")
    assert c.gather_transport == "all_gather" and c.gather_chunks == 1 and c.wire_format == "columns"   # round 4: one piece by default
    c.gather_chunks = 3
    assert c.wire_format == "records"                     # the legacy chunk pipeline keeps records on the wire
    with pytest.raises(ValueError):
        c.set_gather_transport("ring")
    assert c.set_gather_transport("sdma") == "sdma"      # world 1: nothing to set up, nothing to fall back from


def test_load_rows_stores_owned_range_and_replicated_head():
    """ShardedEmbeddingCache.load_rows: every rank keeps its own row range and, with replicated_rows, the head of
    the table, whatever chunking the rows arrive in (no process group needed: rank / world are given)."""
    from scone_amd import NGramExtractor
    from scone_amd.distributed import ShardedEmbeddingCache, shard_range
    rng = np.random.default_rng(1)
    n, d, world, head = 1000, 8, 4, 130
    keys = np.zeros((n, 3), dtype=np.uint32)
    keys[:, 0] = np.arange(n)
    ex = NGramExtractor.from_arrays(keys, np.ones(n, dtype=np.uint8), max_n=3)
    table = rng.standard_normal((n, d)).astype(np.float32)
    for rank in range(world):
        a, b = shard_range(n, rank, world)
        shard = OracleShard(keys, np.ones(n, dtype=np.uint8), 3, table, a, b)
        shard.shard_set_head(head)
        cache = ShardedEmbeddingCache(ex, d, rank=rank, world=world, n_rows=n, table=shard, replicated_rows=head)
        for r0 in range(0, n, 96):                                   # chunks that straddle the head and the shard edges
            cache.load_rows(torch.from_numpy(table[r0:r0 + 96]), r0)
        owned = sorted((r0, r0 + m) for kind, r0, m in shard.stored if kind == "owned")
        heads = sorted((r0, r0 + m) for kind, r0, m in shard.stored if kind == "head")
        assert owned[0][0] == a and owned[-1][1] == b and all(x[1] == y[0] for x, y in zip(owned, owned[1:]))
        assert heads[0][0] == 0 and heads[-1][1] == head and all(x[1] == y[0] for x, y in zip(heads, heads[1:]))


def test_shard_ranges_partition_and_owner():
    from scone_amd.distributed import owner_of, shard_range
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (1, 2), (1000, 1)):
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[r][1] == rs[r + 1][0] for r in range(w - 1))
        ids = torch.arange(n)
        own = owner_of(ids, n, w)
        for r, (a, b) in enumerate(rs):
            assert torch.all(own[a:b] == r)
