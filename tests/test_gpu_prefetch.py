"""`scone_embed_prefetch` and the staging pipeline of a pinned-host table under concurrency (round 5).

* the staging pipeline called from several host threads (one pipeline per handle: the calls are serialised on the handle's
  staging lock; SURVEY 8b "lookups are thread-safe and stream-ordered"),
* the discard race the round-4 advisor found (a prefetched batch dropped by the lookup of another one on a cold cache: its
  copy kernel against the preparation that re-uses its record set),
* tables without a staging pipeline: the announcement is a no-op (the match of the next batch on a side stream was built,
  measured slower than match-then-gather on one stream at every batch size, and removed: profiles/r05b, profiles/r05c).

Every result is compared with the HBM-resident twin of the same table, which the parity suites pin to the oracle and the
golden fixtures (tests/test_gpu_parity.py, tests/test_gpu_bench_shape.py)."""

import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _vocab(rng, vocab, n, max_n=3):
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    return keys, lens


def _table(keys, lens, table, fmt, max_n=3, **kw):
    from scone_amd.hip_backend import SconeTable
    t = SconeTable(max_n, len(lens), table.shape[1], fmt, **kw)
    t.index_build(keys, lens)
    t.store_f32(torch.from_numpy(table))
    return t


def test_small_batches_and_other_dims_ignore_the_announcement():
    """Tables without a staging pipeline: the announcement is a no-op, whatever the batch size; the lookup is what it was."""
    rng = np.random.default_rng(503)
    vocab, n = 31, 1000
    keys, lens = _vocab(rng, vocab, n)
    for fmt, d in (("int8", 768), ("fp32", 36)):
        t = _table(keys, lens, rng.standard_normal((n, d)).astype(np.float32), fmt)
        for shape in ((4, 32), (80, 512)):                     # one-launch kernel / match + gather
            tok = torch.from_numpy(rng.integers(0, vocab, size=shape)).to("cuda", torch.int32)
            ref = t.embed(tok, out_dtype=torch.float32).clone()
            t.embed_prefetch(tok, tokens_ready=True)
            t.embed_prefetch(tok)
            assert torch.equal(t.embed(tok, out_dtype=torch.float32), ref)
        assert t.status() == 0


# ------------------------------------------------------------------ the staging pipeline of a pinned-host table
def _pinned_pair(rng, fmt="int8", d=768, n=6000, vocab=47, hot=200, stage=256, **kw):
    keys, lens = _vocab(rng, vocab, n)
    table = rng.standard_normal((n, d)).astype(np.float32)
    ref = _table(keys, lens, table, fmt)
    pin = _table(keys, lens, table, fmt, placement="pinned_host", hot_rows=hot, stage_tokens=stage, **kw)
    return ref, pin, vocab


def test_staging_pipeline_from_two_host_threads():
    """SURVEY 8b "lookups are thread-safe and stream-ordered" on a stage_tokens > 0 handle (round-4 VERDICT, weak #7): two host
    threads, each on its own stream, run lookups of several chunks and announcements on ONE pinned-host handle.  The calls are
    serialised on the handle's staging lock; every result equals the HBM-resident twin's."""
    rng = np.random.default_rng(506)
    ref, pin, vocab = _pinned_pair(rng)
    d = 768
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((64, d)).astype(np.float32)).half().cuda()
    toks = [torch.from_numpy(rng.integers(0, vocab, size=s)).to("cuda", torch.int32) for s in ((40, 64), (9, 50), (64, 64), (1, 3))]
    want = [ref.embed(x, wte=wte, wpe=wpe).clone() for x in toks]
    torch.cuda.synchronize()
    errors = []

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(60):
                    k = int(r.integers(len(toks)))
                    out = pin.embed(toks[k], wte=wte, wpe=wpe)
                    if it % 3 == 0:
                        pin.embed_prefetch(toks[int(r.integers(len(toks)))], tokens_ready=True)
                    if it % 4 == 0:
                        stream.synchronize()
                        if not torch.equal(out, want[k]):
                            errors.append((seed, it, k))
            stream.synchronize()
        except Exception as e:
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert pin.status() == 0
    assert pin.stage_counters()["chunks"] > 0


def test_discarded_announcement_on_a_cold_cache():
    """Round-4 advisor, medium: an announced batch A that the lookup of ANOTHER batch B discards has no lookup of its own, so
    nothing but the copy stream's own event orders A's copy kernel against the preparation that re-uses its record set four
    chunks later.  Cold cache, every row a miss (the copies are PCIe-bound and slow), B of 8 chunks queued right behind the
    announcement of A -- many rounds, fresh rows each time; B's output must equal the HBM-resident twin's and no status bit
    may rise."""
    rng = np.random.default_rng(507)
    ref, pin, vocab = _pinned_pair(rng, n=60_000, vocab=211, hot=0, stage=2048, d=1024, fmt="int4")
    for rep in range(12):
        a = torch.from_numpy(rng.integers(0, vocab, size=(64, 64))).to("cuda", torch.int32)      # 2 chunks prepared ahead
        b = torch.from_numpy(rng.integers(0, vocab, size=(256, 64))).to("cuda", torch.int32)     # 8 chunks
        torch.cuda.synchronize()
        pin.embed_prefetch(a, tokens_ready=True)
        out = pin.embed(b, out_dtype=torch.float32)
        assert torch.equal(out, ref.embed(b, out_dtype=torch.float32)), rep
        # what A's copies brought in must be A's rows: look A up afterwards as well
        assert torch.equal(pin.embed(a, out_dtype=torch.float32), ref.embed(a, out_dtype=torch.float32)), rep
    assert pin.status() == 0


def test_index_mutation_drops_a_pending_announcement():
    """Records prepared before the index changed are stale (they hold cache slots of rows found through the OLD index): an index
    build between the announcement and the lookup drops the staging pipeline with them, and the lookup sees the new f-grams --
    equal to an HBM-resident table built with the full index."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(508)
    vocab, n, d = 23, 900, 768
    keys, lens = _vocab(rng, vocab, n)
    _, first = np.unique(np.concatenate([keys, lens[:, None].astype(np.uint32)], axis=1), axis=0, return_index=True)
    first = np.sort(first)
    keys, lens = keys[first], lens[first]                     # distinct keys: id = row number in both tables
    n = len(lens)
    table = rng.standard_normal((n, d)).astype(np.float32)
    half = n // 2
    full = _table(keys, lens, table, "int8")
    pin = SconeTable(3, n, d, "int8", placement="pinned_host", hot_rows=50, stage_tokens=128)
    pin.store_f32(torch.from_numpy(table))
    pin.index_build(keys[:half], lens[:half])
    tok = torch.from_numpy(rng.integers(0, vocab, size=(16, 64))).to("cuda", torch.int32)
    before = pin.embed(tok, out_dtype=torch.float32).clone()
    pin.embed_prefetch(tok, tokens_ready=True)                # two chunks matched, placed and copied against the half index
    pin.index_build(keys[half:], lens[half:], id0=half)
    after = pin.embed(tok, out_dtype=torch.float32)
    assert torch.equal(after, full.embed(tok, out_dtype=torch.float32))
    assert not torch.equal(after, before)
    assert pin.status() == 0


def test_a_failed_staged_call_drops_the_pipeline_and_the_next_one_starts_cold():
    """Round-5 advisor: a staged call that fails must not leave a cache whose bookkeeping names rows that were never copied.  A
    call that fails AFTER its chunks were prepared (here, through the bare ABI: an output dtype the lookup refuses, found only
    when the first chunk's lookup is launched) drops the whole pipeline -- cache included -- and the next call rebuilds it:
    counters restart, results equal the HBM-resident twin's before and after."""
    from scone_amd import _lib as L
    rng = np.random.default_rng(509)
    ref, pin, vocab = _pinned_pair(rng, stage=256)
    toks = [torch.from_numpy(rng.integers(0, vocab, size=(12, 128))).to("cuda", torch.int32) for _ in range(3)]
    for t in toks[:2]:
        assert torch.equal(pin.embed(t, out_dtype=torch.float32), ref.embed(t, out_dtype=torch.float32))
    before = pin.stage_counters()
    assert before["chunks"] > 0 and before["rows_copied"] > 0
    out = torch.empty(12, 128, 768, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.lib().scone_embed(pin._h, toks[2].data_ptr(), 12, 128, None, 0, None, 0, None, L.REDUCE_MEAN, out.data_ptr(), 99, stream)
    assert rc != L.OK and b"out_dtype" in L.lib().scone_last_error(pin._h)
    torch.cuda.synchronize()
    assert pin.stage_counters()["chunks"] == 0                         # the pipeline is gone (no cache, no counters) ...
    for t in toks:                                                       # ... and comes back cold: same results
        assert torch.equal(pin.embed(t, out_dtype=torch.float32), ref.embed(t, out_dtype=torch.float32))
    after = pin.stage_counters()
    assert after["chunks"] > 0 and after["rows_copied"] > 0
    assert pin.status() == 0 and ref.status() == 0


def test_pipeline_chooses_its_side_streams_against_the_callers_stream_and_rebinds():
    """Round 6: the pipeline picks PREP / COPY among six candidate streams by MEASURED overlap with the caller's stream
    (scone_stage_bind) -- with streams simply created at that point the cached step was 0.92 or 1.5 ms depending on how many
    other streams the process had used before (profiles/r06i, r06j).  Correctness here: other streams used first, lookups from
    the default stream, then a run of lookups from a second stream (the pipeline re-binds after a few), then back; every
    result equals the HBM-resident twin's."""
    rng = np.random.default_rng(511)
    x = torch.zeros(1 << 16, device="cuda")
    keep = []
    for _ in range(5):                                                   # the process has used other streams before
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x.add_(1.0)
        keep.append(s)
    torch.cuda.synchronize()
    ref, pin, vocab = _pinned_pair(rng, stage=256)
    toks = [torch.from_numpy(rng.integers(0, vocab, size=(10, 128))).to("cuda", torch.int32) for _ in range(4)]
    for t in toks[:2]:
        assert torch.equal(pin.embed(t, out_dtype=torch.float32), ref.embed(t, out_dtype=torch.float32))
    other = torch.cuda.Stream()
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):
        for i in range(7):                                               # >= 4 consecutive calls: the pipeline re-binds to `other`
            t = toks[i % 4]
            pin.embed_prefetch(toks[(i + 1) % 4], tokens_ready=True) if i % 2 else None
            got = pin.embed(t, out_dtype=torch.float32)
            want = ref.embed(t, out_dtype=torch.float32)
            other.synchronize()
            assert torch.equal(got, want), i
    torch.cuda.current_stream().wait_stream(other)
    for t in toks:
        assert torch.equal(pin.embed(t, out_dtype=torch.float32), ref.embed(t, out_dtype=torch.float32))
    assert pin.status() == 0 and ref.status() == 0


def test_streams_overlap_probe_and_side_stream_choice():
    """scone_streams_overlap: a stream never overlaps with itself; a side stream picked by SconeTable.pick_side_stream runs
    beside the current stream; among more streams than the runtime has hardware queues some pair shares a queue (which is why
    the probe exists: profiles/r06i)."""
    rng = np.random.default_rng(512)
    keys, lens = _vocab(rng, 31, 500)
    t = _table(keys, lens, rng.standard_normal((500, 64)).astype(np.float32), "fp32")
    cur = torch.cuda.current_stream()
    assert t.streams_overlap(cur, cur) is False
    side = t.pick_side_stream()
    assert side.cuda_stream != cur.cuda_stream and t.streams_overlap(cur, side) is True
    streams = [torch.cuda.Stream() for _ in range(9)]
    shares = [(i, j) for i in range(len(streams)) for j in range(i + 1, len(streams)) if not t.streams_overlap(streams[i], streams[j])]
    assert shares, "nine streams on four hardware queues: at least one pair must share a queue"
    assert len(shares) < 36, "and not all of them"
    # the lookup is unaffected by all of this
    tok = torch.from_numpy(rng.integers(0, 31, size=(4, 32))).to("cuda", torch.int32)
    a = t.embed(tok, out_dtype=torch.float32).clone()
    with torch.cuda.stream(side):
        b = t.embed(tok, out_dtype=torch.float32)
    side.synchronize()
    assert torch.equal(a, b) and t.status() == 0
