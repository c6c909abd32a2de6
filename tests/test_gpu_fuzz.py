"""GPU parity fuzz: random vocabularies (tiny token alphabets -> overlapping, nested and duplicate f-grams), random batch
shapes, every table format, both lookup forms, both lookup modes, optional position ids -- ids bit-exact and fp32 results
bit-exact against the oracle on the dequantised table.  SCONE_FUZZ_CASES scales the number of cases (default 48)."""

import os

import numpy as np
import pytest
import torch

from oracle import ref_port as R

pytestmark = pytest.mark.gpu


def _dequantised(fmt, table):
    if fmt == "fp32":
        return table
    if fmt == "fp16":
        return table.astype(np.float16).astype(np.float32)
    if fmt == "int8":
        return R.dequantize_i8(*R.quantize_i8(table))
    return R.dequantize_i4(*R.quantize_i4(table))


@pytest.mark.parametrize("form", ["one_launch", "two_kernels"])
def test_fuzz_lookup_vs_oracle(form, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd import EmbeddingCache, NGramExtractor
    if form == "two_kernels":
        monkeypatch.setenv("SCONE_FUSED_MAX_TOKENS", "0")
    else:
        monkeypatch.delenv("SCONE_FUSED_MAX_TOKENS", raising=False)
    n_cases = int(os.environ.get("SCONE_FUZZ_CASES", "48"))
    rng = np.random.default_rng(20260 + (form == "two_kernels"))
    for case in range(n_cases):
        max_n = int(rng.integers(1, 5))
        vocab = int(rng.choice([2, 3, 5, 17, 300]))
        fmt, d = [("fp32", 768), ("fp16", 768), ("int8", 768), ("int8", 1024), ("int4", 1024), ("fp16", 1280), ("int8", 64),
                  ("fp32", 24), ("int4", 256)][int(rng.integers(9))]
        n = int(rng.integers(1, 400))
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = (rng.standard_normal((n, d)) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
        if rng.random() < 0.3:
            table[rng.integers(0, n)] = 0.0                                   # an all-zero row (scale 0)
        B, T = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        tok = rng.integers(-1 if rng.random() < 0.2 else 0, vocab + 1, size=(B, T))   # -1 and `vocab` never match
        mode = "cover" if rng.random() < 0.7 else "longest_suffix"
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        cache = EmbeddingCache(ex, d, table_format=fmt, lookup_mode=mode)
        cache.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
        off, ids = cache.match(torch.from_numpy(tok))
        ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
        tag = (form, case, max_n, vocab, fmt, d, n, B, T, mode)
        assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(ids.cpu().numpy(), ri), tag
        deq = _dequantised(fmt, table)
        wte = (rng.standard_normal((vocab + 1, d)) * 0.1).astype(np.float32)
        wpe = (rng.standard_normal((T + 3, d)) * 0.1).astype(np.float32)
        pos = rng.integers(0, T + 3, size=(B, T)) if (rng.random() < 0.4 and mode == "cover") else None
        tok_c = np.clip(tok, 0, vocab)                                        # wte rows exist for 0..vocab
        got = cache.embed_tokens(torch.from_numpy(tok_c), wte=torch.from_numpy(wte).cuda(), wpe=torch.from_numpy(wpe).cuda(),
                                 position_ids=None if pos is None else torch.from_numpy(pos),
                                 out_dtype=torch.float32).cpu().numpy()
        ro2, ri2 = R.hits_to_csr(R.match_hits(keys, lens, tok_c, max_n))
        if mode == "cover":
            fg = R.embed_numpy(deq, ro2, ri2, "mean").reshape(B, T, d)
            want = (wte[tok_c] + fg) + wpe[np.arange(T)[None, :] if pos is None else pos]
        else:
            want = R.paper_embed(R._key_dict(keys, lens), max_n, tok_c, deq, wte, wpe)
        assert np.array_equal(got, want.astype(np.float32)), tag


def test_fuzz_row_exchange_vs_unsharded():
    """Random (world, replicated head, format, max_n, batch shape): the by-hand row exchange between W shards on one GPU
    is bit-identical to the unsharded table, and the number of records equals the references outside the head."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd.distributed import shard_range
    from scone_amd.hip_backend import SconeTable
    n_cases = max(4, int(os.environ.get("SCONE_FUZZ_CASES", "48")) // 4)
    rng = np.random.default_rng(4711)
    for case in range(n_cases):
        world = int(rng.integers(1, 9))
        max_n = int(rng.integers(1, 5))
        fmt, d = [("int8", 768), ("int4", 1024), ("fp16", 1280), ("fp32", 768), ("int8", 1024)][int(rng.integers(5))]
        vocab = int(rng.choice([3, 11, 200]))
        n = int(rng.integers(world, 600))
        head = int(rng.choice([0, 1, n // 3, n]))
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = rng.standard_normal((n, d)).astype(np.float32)
        B, T = int(rng.integers(1, 30)), int(rng.integers(1, 50))
        tok = torch.from_numpy(rng.integers(0, vocab + 1, size=(B, T)))
        wte = torch.from_numpy(rng.standard_normal((vocab + 1, d)).astype(np.float32)).half().cuda()
        wpe = torch.from_numpy(rng.standard_normal((T, d)).astype(np.float32)).half().cuda()
        full = SconeTable(max_n, n, d, fmt)
        full.index_build(keys, lens)
        full.store_f32(torch.from_numpy(table))
        want = full.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        _, ids = full.match_csr(tok)
        shards = []
        for r in range(world):
            a, b = shard_range(n, r, world)
            s = SconeTable(max_n, n, d, fmt, row_begin=a, row_end=b)
            s.index_build(keys, lens)
            if b > a:
                s.store_f32(torch.from_numpy(table[a:b]), row0=a)
            if head:
                s.shard_set_head(head)
                s.shard_head_store_f32(torch.from_numpy(table[:head]), row0=0)
            shards.append(s)
        tag = (case, world, max_n, fmt, d, vocab, n, head, B, T)
        # the slice exchange: chunk q of a shard's plan = the distinct rows of its own that slice q references
        rec = shards[0].shard_record_bytes()
        ends = [s.shard_gather_plan_chunks(tok, world, dedup_across_chunks=False) for s in shards]
        cnt = [[e[0]] + [e[q] - e[q - 1] for q in range(1, world)] for e in ends]
        sends = []
        for r, s in enumerate(shards):
            buf = torch.empty((max(ends[r][-1], 1), rec), dtype=torch.uint8, device="cuda")
            s.shard_gather_pack_range(0, ends[r][-1], buf[:ends[r][-1]])
            sends.append(buf)
        bper = (B + world - 1) // world
        for q in range(world):
            recv = torch.cat([sends[r][sum(cnt[r][:q]):sum(cnt[r][:q + 1])] for r in range(world)]).contiguous()
            b0, b1 = min(q * bper, B), min(q * bper + bper, B)
            shards[q].shard_gather_add_records(recv, 0, recv.shape[0])
            if b1 > b0:
                got = torch.empty(((b1 - b0) * T, d), dtype=torch.float16, device="cuda")
                shards[q].shard_gather_embed_range(tok, b0, b1, recv, got, wte=wte, wpe=wpe, out_is_slice=True)
                assert torch.equal(got, want[b0 * T:b1 * T]), tag + (q,)
            assert shards[q].status() == 0, tag
        # the all-gather form: every shard packs one record per DISTINCT row it owns that the batch references, any shard
        # reduces the whole batch out of [head | all records]
        sends = [s_.shard_gather_pack(s_.shard_gather_plan(tok)) for s_ in shards]
        order = rng.permutation(world)                                       # records may arrive in any order
        recv = torch.cat([sends[r] for r in order]).contiguous()
        assert recv.shape[0] == int(torch.unique(ids[ids >= head]).numel()), tag
        for q in {int(rng.integers(world)), 0}:
            if case % 3 == 0:
                # the plan's match sharded over the ranks: shard r matches run r of the sequences, the list records laid end
                # to end are what the all-gather delivers, shard q plans from them (same claims as its own match gives)
                wd = shards[q].ell_width()
                ell = torch.empty((B * T, wd), dtype=torch.int32, device="cuda")
                for r in range(world):
                    b0, b1 = min(r * bper, B), min(r * bper + bper, B)
                    shards[r].shard_gather_match(tok, b0, b1, ell[b0 * T:max(b1, b0) * T])
                assert shards[q].shard_gather_plan_ell(ell, B, T, 1) == [sends[q].shape[0]], tag
            else:
                shards[q].shard_gather_plan(tok)                             # a rank embeds after ITS OWN plan of this batch
            got = shards[q].shard_gather_embed(tok, recv, wte=wte, wpe=wpe, out_dtype=torch.float16)
            assert torch.equal(got, want), tag + ("gather_rows", q)
            assert shards[q].status() == 0, tag
        if fmt != "fp32" or d % 8 == 0:
            # the same exchange with columns on the wire: payload rows | scales | the senders' hash fragments
            cnts = [s_.shard_gather_plan(tok) for s_ in shards]
            slots = [SconeTable.cols_frag_slots(c) for c in cnts]
            rb, fo = [sum(cnts[:r]) for r in range(world)], [sum(slots[:r]) for r in range(world)]
            tot, pb, sb = sum(cnts), shards[0].payload_bytes(), shards[0].scale_bytes()
            c_rows = torch.empty((max(tot, 1), pb), dtype=torch.uint8, device="cuda")
            c_sc = torch.empty((head + max(tot, 1), sb), dtype=torch.uint8, device="cuda") if sb else None
            c_fr = torch.empty(sum(slots), dtype=torch.int64, device="cuda")
            for r, s_ in enumerate(shards):
                s_.shard_cols_pack(0, cnts[r], c_rows[rb[r]:rb[r] + cnts[r]], None if c_sc is None else c_sc[head + rb[r]:head + rb[r] + cnts[r]],
                                   c_fr[fo[r]:fo[r] + slots[r]])
            q = int(rng.integers(world))
            if c_sc is not None and head:
                shards[q].shard_head_scales_into(c_sc)
            got = torch.empty((B * T, d), dtype=torch.float16, device="cuda")
            shards[q].shard_cols_embed(tok, 0, B, c_rows, tot, c_sc, c_fr, fo, slots, rb, got, wte=wte, wpe=wpe)
            assert torch.equal(got, want), tag + ("gather_rows, columns", q)
            assert shards[q].status() == 0, tag


def test_fuzz_pinned_host_vs_hbm():
    """Random (format, hot head, staging chunk, batch shape): rows in pinned host DRAM -- read in place or staged through
    HBM in chunks -- give the bits the HBM-resident table gives."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd import EmbeddingCache, NGramExtractor
    n_cases = max(4, int(os.environ.get("SCONE_FUZZ_CASES", "48")) // 4)
    rng = np.random.default_rng(99)
    for case in range(n_cases):
        max_n = int(rng.integers(1, 5))
        fmt, d = [("int8", 768), ("int4", 1024), ("fp16", 1280), ("fp32", 768), ("int8", 2048)][int(rng.integers(5))]
        vocab = int(rng.choice([5, 40]))
        n = int(rng.integers(2, 500))
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        B, T = int(rng.integers(1, 60)), int(rng.integers(1, 40))
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
        wpe = torch.from_numpy(rng.standard_normal((T, d)).astype(np.float32)).half().cuda()
        ref = EmbeddingCache(ex, d, table_format=fmt)
        ref.cache_embeddings(list(range(n)), table, verbose=False)
        want = ref.embed_tokens(tok, wte=wte, wpe=wpe)
        hot = int(rng.choice([0, 1, n // 2, n - 1]))
        stage = int(rng.choice([0, T, 3 * T, 100000]))
        c = EmbeddingCache(ex, d, table_format=fmt, placement="pinned_host", hot_rows=hot, stage_tokens=stage)
        c.cache_embeddings(list(range(n)), table, verbose=False)
        for _ in range(2):                                                    # twice: staging buffers and generations are reused
            assert torch.equal(c.embed_tokens(tok, wte=wte, wpe=wpe), want), (case, max_n, fmt, d, n, B, T, hot, stage)


def test_staged_prefetch_generation_wrap():
    """The staging slot map tags claims with an 8-bit generation; more than 255 chunks on one handle wrap it (the map is
    cleared and generations restart): results stay bit-identical across the wrap."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd import EmbeddingCache, NGramExtractor
    rng = np.random.default_rng(5)
    vocab, n, d, T = 30, 400, 768, 16
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    table = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
    ref = EmbeddingCache(ex, d, table_format="int8")
    ref.cache_embeddings(list(range(n)), table, verbose=False)
    c = EmbeddingCache(ex, d, table_format="int8", placement="pinned_host", hot_rows=10, stage_tokens=T)   # one sequence per chunk
    c.cache_embeddings(list(range(n)), table, verbose=False)
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    chunks = 0
    for it in range(12):
        B = int(rng.integers(20, 40))
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        assert torch.equal(c.embed_tokens(tok, wte=wte), ref.embed_tokens(tok, wte=wte)), it
        chunks += B
    assert chunks > 255 + 40


def test_fuzz_fit_gpu_vs_host_fit():
    """Random corpora over tiny alphabets (many count ties, first-seen order decides): scone_fit gives the f-grams and ids of
    the host fit (= the reference's Counter.most_common order) for random max_n / min_freq / max_f_grams."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd import NGramExtractor
    n_cases = max(6, int(os.environ.get("SCONE_FUZZ_CASES", "48")) // 2)
    rng = np.random.default_rng(808)
    for case in range(n_cases):
        max_n = int(rng.integers(1, 5))
        vocab = int(rng.choice([2, 4, 30]))
        texts = [rng.integers(0, vocab, size=int(rng.integers(0, 60))).tolist() for _ in range(int(rng.integers(1, 25)))]
        min_freq = int(rng.integers(1, 4))
        max_f = int(rng.choice([1, 5, 50, 10_000]))
        host = NGramExtractor(max_n=max_n, min_freq=min_freq, max_f_grams=max_f).fit(texts, verbose=False)
        gpu = NGramExtractor(max_n=max_n, min_freq=min_freq, max_f_grams=max_f).fit_gpu(texts, verbose=False)
        hk, hl = host.key_arrays()
        gk, gl = gpu.key_arrays()
        assert np.array_equal(hl, gl) and np.array_equal(hk, gk), (case, max_n, vocab, min_freq, max_f, len(texts))


def test_shard_workspaces_survive_mixed_modes_and_growing_batches():
    """ONE set of shard handles, batches that grow and shrink, slice exchange and all-gather form interleaved: every
    workspace of the shard state is sized for the call that uses it (each has its own capacity)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd.distributed import shard_range
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(2)
    world, max_n, fmt, d, vocab, n, head = 4, 3, "int8", 768, 13, 900, 40
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    full = SconeTable(max_n, n, d, fmt)
    full.index_build(keys, lens)
    full.store_f32(torch.from_numpy(table))
    shards = []
    for r in range(world):
        a, b = shard_range(n, r, world)
        s = SconeTable(max_n, n, d, fmt, row_begin=a, row_end=b)
        s.index_build(keys, lens)
        s.store_f32(torch.from_numpy(table[a:b]), row0=a)
        s.shard_set_head(head)
        s.shard_head_store_f32(torch.from_numpy(table[:head]), row0=0)
        shards.append(s)
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    for step, (B, T, mode) in enumerate([(4, 8, "slice"), (60, 40, "gather"), (8, 16, "gather"), (96, 64, "slice"), (2, 5, "slice"),
                                         (128, 64, "gather"), (100, 50, "slice")]):
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        want = full.embed(tok, wte=wte).reshape(B * T, d)
        if mode == "gather":
            recv = torch.cat([s.shard_gather_pack(s.shard_gather_plan(tok)) for s in shards]).contiguous()
            for q in range(world):
                assert torch.equal(shards[q].shard_gather_embed(tok, recv, wte=wte, out_dtype=torch.float16), want), (step, q)
        else:
            rec = shards[0].shard_record_bytes()
            ends = [s.shard_gather_plan_chunks(tok, world, dedup_across_chunks=False) for s in shards]
            cnt = [[e[0]] + [e[q] - e[q - 1] for q in range(1, world)] for e in ends]
            sends = []
            for r, s in enumerate(shards):
                buf = torch.empty((max(ends[r][-1], 1), rec), dtype=torch.uint8, device="cuda")
                s.shard_gather_pack_range(0, ends[r][-1], buf[:ends[r][-1]])
                sends.append(buf)
            bper = (B + world - 1) // world
            for q in range(world):
                recv = torch.cat([sends[r][sum(cnt[r][:q]):sum(cnt[r][:q + 1])] for r in range(world)]).contiguous()
                b0, b1 = min(q * bper, B), min(q * bper + bper, B)
                shards[q].shard_gather_add_records(recv, 0, recv.shape[0])
                if b1 > b0:
                    got = torch.empty(((b1 - b0) * T, d), dtype=torch.float16, device="cuda")
                    shards[q].shard_gather_embed_range(tok, b0, b1, recv, got, wte=wte, out_is_slice=True)
                    assert torch.equal(got, want[b0 * T:b1 * T]), (step, q)
        assert all(s.status() == 0 for s in shards), step
