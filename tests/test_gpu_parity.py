"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures captured from the reference.  Run with ``-m gpu`` on an MI355X.

Bars: id lists bit-exact; fp32 table + fp32 out bit-exact against the reference's own
``mean`` (sequential fp32 sum, IEEE division); quantised tables / fp16 out within
1e-3 relative of the oracle run on the dequantised table (BASELINE.json north_star).
"""

import os
import time

import numpy as np
import pytest
import torch

from oracle import ref_port as R

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3     # north_star: "within 1e-3 rel for the fp16 summed embedding"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scone_amd import _lib
    _lib.lib()          # fail loudly if the extension is missing


@pytest.fixture(params=["one_launch", "two_kernels"])
def lookup_form(request, monkeypatch):
    """Both forms of the fused lookup on the same inputs: batches up to 32768 tokens normally take the one-launch kernel
    (k_embed_fused); SCONE_FUSED_MAX_TOKENS=0 (read when a handle is created) sends every batch through
    k_match_ell + k_embed_wave."""
    if request.param == "two_kernels":
        monkeypatch.setenv("SCONE_FUSED_MAX_TOKENS", "0")
    else:
        monkeypatch.delenv("SCONE_FUSED_MAX_TOKENS", raising=False)
    return request.param


def _extractor(keys, lens, max_n):
    from scone_amd import NGramExtractor
    return NGramExtractor.from_arrays(keys, lens, max_n=max_n)


def _cache(keys, lens, max_n, table, fmt="fp32", **kw):
    from scone_amd import EmbeddingCache
    ex = _extractor(keys, lens, max_n)
    c = EmbeddingCache(ex, table.shape[1], table_format=fmt, **kw)
    c.cache_embeddings(list(range(table.shape[0])), torch.from_numpy(table), verbose=False)
    return c


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))


# ------------------------------------------------------------------ a1-a3: index + match
def test_match_golden_bit_exact(golden_dir):
    from scone_amd.hip_backend import SconeTable
    z = np.load(os.path.join(golden_dir, "match.npz"))
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        t = SconeTable(max_n, len(lens))
        t.index_build(keys, lens)
        nk, cap, dups = t.index_stats()
        assert nk == len(lens) and dups == 0 and cap >= 2 * nk
        for si in range(int(z["n_streams"])):
            tok = torch.from_numpy(z[f"{c}_s{si}_tok"])
            off, ids = t.match_csr(tok)
            assert np.array_equal(off.cpu().numpy(), z[f"{c}_s{si}_off"]), (c, si)
            assert np.array_equal(ids.cpu().numpy(), z[f"{c}_s{si}_ids"]), (c, si)


def test_get_token_f_grams_matches_reference_dict(golden_dir):
    z = np.load(os.path.join(golden_dir, "match.npz"))
    c = "c12"          # the [7,7,7,7] multiplicity case
    ex = _extractor(z[f"{c}_keys"], z[f"{c}_lens"], 3)
    got = ex.get_token_f_grams([7, 7, 7, 7])
    assert got[1] == [(7,), (7, 7), (7, 7), (7, 7, 7), (7, 7, 7)]
    assert got[0] == [(7,), (7, 7), (7, 7, 7)]
    assert ex.get_token_f_grams([]) == {}
    assert ex.get_token_f_grams([8]) == {0: []}
    d = {tuple(int(x) for x in z[f"{c}_keys"][i, :z[f"{c}_lens"][i]]): i for i in range(3)}
    for toks in ([7, 8, 7, 7, 7, 9], [7, 7], [9, 9, 9]):
        assert ex.get_token_f_grams(toks) == R.get_token_f_grams(set(d), 3, toks)


@pytest.mark.parametrize("max_n", [1, 2, 3, 4])
def test_match_batched_random_vs_oracle(max_n):
    """[B, T] batches: windows never cross a sequence boundary; hits array and CSR bit-exact."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(100 + max_n)
    vocab = 23
    n = 600
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    t = SconeTable(max_n, n)
    t.index_build(keys, lens)
    nk, _, dups = t.index_stats()
    assert nk + dups == n and nk == len(R._key_dict(keys, lens))
    for B, T in ((1, 1), (3, 2), (7, 37), (2, 1025), (64, 3)):
        tok = rng.integers(-1, vocab + 1, size=(B, T))
        ref_hits = R.match_hits(keys, lens, tok, max_n)
        hits = t.match(torch.from_numpy(tok)).cpu().numpy()
        assert np.array_equal(hits, ref_hits), (B, T)
        off, ids = t.match_csr(torch.from_numpy(tok))
        ro, ri = R.hits_to_csr(ref_hits)
        assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(ids.cpu().numpy(), ri), (B, T)


def test_index_wide_tokens_and_chunked_build():
    """32-bit token ids (max_n <= 3) and 24-bit ids (max_n = 4); ids assigned by id0 across chunks."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(9)
    for max_n, top in ((3, 2**31 - 1), (4, 2**24 - 2)):
        n = 5000
        lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = rng.integers(0, top, size=(n, max_n), dtype=np.int64).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        t = SconeTable(max_n, n)
        t.index_build(keys[:2000], lens[:2000], id0=0)
        t.index_build(keys[2000:], lens[2000:], id0=2000)
        assert t.index_stats()[0] == n
        pick = rng.integers(0, n, size=64)
        T = max_n
        tok = np.zeros((64, T), dtype=np.int64)
        for r, i in enumerate(pick):
            tok[r, :lens[i]] = keys[i, :lens[i]]
            tok[r, lens[i]:] = top            # never part of a key
        hits = t.match(torch.from_numpy(tok)).cpu().numpy()
        for r, i in enumerate(pick):
            assert hits[lens[i] - 1, r, 0] == i


def test_index_full_and_bad_keys_raise():
    from scone_amd.hip_backend import SconeTable
    t = SconeTable(2, 4, index_capacity=4)
    keys = np.arange(16, dtype=np.uint32).reshape(8, 2)
    with pytest.raises(MemoryError):
        t.index_build(keys, np.full(8, 2, dtype=np.uint8))
    t2 = SconeTable(2, 4)
    with pytest.raises(IndexError):
        t2.index_build(keys[:2], np.array([3, 1], dtype=np.uint8))       # len > max_n


# ------------------------------------------------------------------ a4-a6: gather + aggregate
@pytest.mark.parametrize("use_mm", [False, True])
def test_lookup_golden_fp32_bit_exact(golden_dir, tmp_path, use_mm):
    from scone_amd import EmbeddingCache
    z = np.load(os.path.join(golden_dir, "lookup.npz"))
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        table = z[f"{c}_table"]
        n, d = table.shape
        ex = _extractor(keys, lens, max_n)
        cache = EmbeddingCache(ex, d, cache_dir=str(tmp_path / f"{c}{use_mm}") if use_mm else None,
                               use_memory_map=use_mm)
        cache.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
        # a4: get_embeddings returns the cached rows (fresh CPU fp32 tensor)
        g = cache.get_embeddings(z[f"{c}_gather_ids"].tolist())
        assert g.dtype == torch.float32 and g.device.type == "cpu"
        assert np.array_equal(g.numpy(), z[f"{c}_gather_out"])
        assert cache.get_embeddings([0], torch.device("cuda")).is_cuda
        # a5: per-position stacks, empty positions omitted
        te = cache.get_token_embeddings(z[f"{c}_tok"].tolist())
        assert sorted(te.keys()) == z[f"{c}_te_positions"].tolist()
        p = 0
        for pos, rows in zip(z[f"{c}_te_positions"], z[f"{c}_te_rows"]):
            assert np.array_equal(te[int(pos)].numpy(), z[f"{c}_te_stacks"][p:p + rows])
            p += rows
        # a6: mean over hits, zeros where K = 0, [1, T, d]; fp32 bit-exact, .half() bit-exact
        tok = torch.from_numpy(z[f"{c}_tok"])[None, :]
        agg = cache.embed_tokens(tok, reduce="mean", out_dtype=torch.float32)
        assert agg.shape == (1, tok.shape[1], d)
        assert np.array_equal(agg.cpu().numpy(), z[f"{c}_agg_f32"]), c
        agg16 = cache.embed_tokens(tok, reduce="mean", out_dtype=torch.float16)
        assert np.array_equal(agg16.cpu().numpy().view(np.uint16), z[f"{c}_agg_f16"].view(np.uint16)), c
        # CSR entry point gives the same vectors
        off, ids = cache.match(tok)
        agg2 = cache.table.gather_reduce(off, ids, "mean")
        assert np.array_equal(agg2.cpu().numpy(), z[f"{c}_agg_f32"][0]), c


def test_missing_rows_and_bad_ids_raise(golden_dir):
    z = np.load(os.path.join(golden_dir, "lookup.npz"))
    c = "c4"
    cache = _cache(z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"]), z[f"{c}_table"])
    with pytest.raises(KeyError):
        cache.get_embeddings([10_000])
    del cache.embeddings[0]
    cache._dirty = True
    with pytest.raises(KeyError):
        cache.get_embeddings([0])


@pytest.mark.parametrize("fmt", ["fp32", "fp16", "int8", "int4"])
@pytest.mark.parametrize("d", [128, 768, 1024])
@pytest.mark.parametrize("max_n", [3, 4])
def test_embed_formats_vs_oracle(fmt, d, max_n, lookup_form):
    """Quantise on the GPU, read the dequantised table back, run the oracle on it."""
    rng = np.random.default_rng(sum(map(ord, fmt)) * 10007 + d * 13 + max_n)
    vocab, n = 31, 900
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    table[5] = 0.0
    cache = _cache(keys, lens, max_n, table, fmt)
    # the device table is exactly the numpy statement of the format
    deq = cache.table.gather_rows(torch.arange(n)).cpu().numpy()
    if fmt == "fp32":
        expect = table
    elif fmt == "fp16":
        expect = table.astype(np.float16).astype(np.float32)
    elif fmt == "int8":
        expect = R.dequantize_i8(*R.quantize_i8(table))
    else:
        expect = R.dequantize_i4(*R.quantize_i4(table))
    assert np.array_equal(deq, expect), "device quantiser differs from oracle/ref_port.py"
    B, T = 5, 70
    tok = rng.integers(0, vocab + 1, size=(B, T))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
    for reduce in ("mean", "sum"):
        ref = R.embed_numpy(expect, ro, ri, reduce).reshape(B, T, d)
        out = cache.embed_tokens(torch.from_numpy(tok), reduce=reduce, out_dtype=torch.float32).cpu().numpy()
        assert np.array_equal(out, ref), (fmt, d, reduce, _rel(out, ref))
        out16 = cache.embed_tokens(torch.from_numpy(tok), reduce=reduce, out_dtype=torch.float16)
        assert _rel(out16.float().cpu().numpy(), ref) < REL_TOL
        outbf = cache.embed_tokens(torch.from_numpy(tok), reduce=reduce, out_dtype=torch.bfloat16)
        assert _rel(outbf.float().cpu().numpy(), ref) < 8e-3       # bf16 has 8 significand bits


# ------------------------------------------------------------------ a7: combine
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_fused_combine_vs_oracle(dtype, lookup_form):
    rng = np.random.default_rng(77)
    vocab, n, d, max_n, B, T = 50, 700, 768, 3, 3, 40
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    cache = _cache(keys, lens, max_n, table, "int8")
    deq = R.dequantize_i8(*R.quantize_i8(table))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).to(dtype).cuda()
    wpe = torch.from_numpy(rng.standard_normal((64, d)).astype(np.float32)).to(dtype).cuda()
    tok = rng.integers(0, vocab, size=(B, T))
    pos = rng.integers(0, 64, size=(B, T))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
    fg = torch.from_numpy(R.embed_numpy(deq, ro, ri, "mean").reshape(B, T, d))
    for position_ids in (None, torch.from_numpy(pos)):
        ref = R.combine(torch.from_numpy(tok), fg, wte.float().cpu(), wpe.float().cpu(),
                        position_ids=position_ids).numpy()
        out = cache.embed_tokens(torch.from_numpy(tok), wte=wte, wpe=wpe, position_ids=position_ids, check=True)
        assert out.dtype == dtype and out.shape == (B, T, d)
        if dtype == torch.float32:
            assert np.array_equal(out.cpu().numpy(), ref)
        else:
            assert _rel(out.float().cpu().numpy(), ref) < (REL_TOL if dtype == torch.float16 else 8e-3)
    # token outside wte raises like nn.Embedding
    bad = tok.copy()
    bad[0, 0] = vocab + 5
    with pytest.raises(IndexError):
        cache.embed_tokens(torch.from_numpy(bad), wte=wte, wpe=wpe, check=True)


def test_scone_embedding_module_paths(golden_dir):
    """SconeEmbedding: explicit f_gram_embeddings (reference arithmetic, golden) and fused lookup."""
    from scone_amd.models.language_model import SconeEmbedding, fold_projection
    z = np.load(os.path.join(golden_dir, "combine.npz"))
    for c in z["cases"]:
        wte = torch.nn.Embedding.from_pretrained(torch.from_numpy(z[f"{c}_wte"])).cuda()
        wpe = torch.nn.Embedding.from_pretrained(torch.from_numpy(z[f"{c}_wpe"])).cuda()
        proj = torch.nn.Linear(z[f"{c}_proj"].shape[1], z[f"{c}_proj"].shape[0], bias=False).cuda()
        proj.weight.data.copy_(torch.from_numpy(z[f"{c}_proj"]))
        emb = SconeEmbedding(wte, wpe, proj)
        pos = z[f"{c}_pos"]
        x = emb(torch.from_numpy(z[f"{c}_input_ids"]).cuda(), torch.from_numpy(z[f"{c}_fg"]).cuda(),
                torch.from_numpy(pos).cuda() if pos.size else None)
        np.testing.assert_allclose(x.detach().cpu().numpy(), z[f"{c}_embeds"], rtol=1e-4, atol=1e-5)
    # fused: projection folded into the table == project the aggregated vector (linearity)
    rng = np.random.default_rng(5)
    vocab, n, d_f, H, T = 40, 300, 64, 128, 33
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    table = torch.from_numpy(rng.standard_normal((n, d_f)).astype(np.float32))
    W = torch.from_numpy(rng.standard_normal((H, d_f)).astype(np.float32) / 8)
    folded = fold_projection(table, W)
    cache = _cache(keys, lens, 3, folded.numpy())
    wte = torch.nn.Embedding(vocab, H).cuda()
    wpe = torch.nn.Embedding(64, H).cuda()
    fused = SconeEmbedding(wte, wpe, None, cache)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(2, T)))
    out = fused(tok.cuda())
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok.numpy(), 3))
    fg = torch.from_numpy(R.embed_numpy(table.numpy(), ro, ri, "mean").reshape(2, T, d_f))
    ref = R.combine(tok, fg, wte.weight.detach().cpu(), wpe.weight.detach().cpu(), proj_weight=W)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)


# ------------------------------------------------------------------ edge cases
def test_empty_and_short_inputs(lookup_form):
    rng = np.random.default_rng(1)
    keys = np.array([[1, 0, 0], [1, 2, 0], [1, 2, 3]], dtype=np.uint32)
    lens = np.array([1, 2, 3], dtype=np.uint8)
    table = rng.standard_normal((3, 64)).astype(np.float32)
    cache = _cache(keys, lens, 3, table)
    assert cache.embed_tokens(torch.zeros((0, 5), dtype=torch.int64)).shape == (0, 5, 64)
    assert cache.embed_tokens(torch.zeros((4, 0), dtype=torch.int64)).shape == (4, 0, 64)
    off, ids = cache.match(torch.zeros((2, 0), dtype=torch.int64))
    assert off.cpu().tolist() == [0] and ids.numel() == 0
    assert cache.get_token_embeddings([]) == {}
    # T < max_n: only the windows that fit (n_gram_extractor.py:118)
    out = cache.embed_tokens(torch.tensor([[1, 2]]), out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out[0, 0], (table[0] + table[1]) / np.float32(2))
    assert np.array_equal(out[0, 1], table[1])
    # all misses -> zeros; negative tokens never match
    out = cache.embed_tokens(torch.tensor([[9, -1, 9]]), out_dtype=torch.float32).cpu().numpy()
    assert not out.any()
    # every candidate hits: K = 6 at an interior position
    keys2 = np.array([[4, 0, 0], [4, 4, 0], [4, 4, 4]], dtype=np.uint32)
    cache2 = _cache(keys2, lens, 3, table)
    off, ids = cache2.match(torch.tensor([[4, 4, 4, 4, 4]]))
    assert (off[3] - off[2]).item() == 6 and ids[off[2]:off[3]].cpu().tolist() == [0, 1, 1, 2, 2, 2]


def test_results_independent_of_batch_shape(lookup_form):
    """Idempotence / shape independence: the same sequence gives identical bits alone or inside a batch."""
    rng = np.random.default_rng(2)
    vocab, n, d = 17, 400, 256
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = _cache(keys, lens, 3, rng.standard_normal((n, d)).astype(np.float32), "int8")
    tok = torch.from_numpy(rng.integers(0, vocab, size=(9, 130)))
    full = cache.embed_tokens(tok, out_dtype=torch.float32)
    again = cache.embed_tokens(tok, out_dtype=torch.float32)
    assert torch.equal(full, again)
    for b in (0, 4, 8):
        assert torch.equal(cache.embed_tokens(tok[b:b + 1], out_dtype=torch.float32)[0], full[b])


# ------------------------------------------------------------------ row shards (single GPU, two handles)
def test_row_sharded_partial_sums_equal_full():
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(4)
    vocab, n, d, max_n = 29, 1000, 768, 3
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(4, 50)))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((50, d)).astype(np.float32)).half().cuda()
    full = SconeTable(max_n, n, d, "int8")
    full.index_build(keys, lens)
    full.store_f32(torch.from_numpy(table))
    ref = full.embed(tok, wte=wte, wpe=wpe)
    ref_sum = full.embed(tok, reduce="sum", out_dtype=torch.float32).reshape(-1, d)
    shards = []
    for a, b in ((0, 300), (300, 1000)):
        s = SconeTable(max_n, n, d, "int8", row_begin=a, row_end=b)
        s.index_build(keys, lens)                     # the index is replicated
        s.store_f32(torch.from_numpy(table[a:b]), row0=a)
        shards.append(s)
    parts = [s.embed_partial(tok) for s in shards]
    assert torch.equal(parts[0][1], parts[1][1])      # every shard knows the full K
    total = parts[0][0] + parts[1][0]                 # what the RCCL reduce-scatter computes
    assert _rel(total.cpu().numpy(), ref_sum.cpu().numpy()) < 1e-6
    ntok = tok.numel()
    halves = [shards[r].finalize(total[a:b], parts[0][1][a:b], tok, a, b, wte=wte, wpe=wpe, out_dtype=torch.float16)
              for r, (a, b) in enumerate(((0, ntok // 2), (ntok // 2, ntok)))]
    out = torch.cat(halves).reshape(ref.shape)
    assert _rel(out.float().cpu().numpy(), ref.float().cpu().numpy()) < REL_TOL


# ------------------------------------------------------------------ synthetic table (full-size property)
def test_synthetic_int4_table_matches_host_generator():
    from scone_amd.hip_backend import SconeTable
    n, d = 50_000, 1024
    t = SconeTable(3, n, d, "int4")
    t.fill_synthetic(7, 0.02 / 127)
    ids = np.array([0, 1, 2, 777, n - 1], dtype=np.int64)
    got = t.gather_rows(torch.from_numpy(ids)).cpu().numpy()
    assert np.array_equal(got, R.dequantize_i4(*R.synth_rows_i4(7, ids, d, 0.02 / 127)))
    rows, scales = t.download(0, 3)
    p, s = R.synth_rows_i4(7, np.arange(3), d, 0.02 / 127)
    assert np.array_equal(rows, p) and np.array_equal(scales, s)


@pytest.mark.parametrize("fmt", ["int8", "fp16", "fp32"])
def test_synthetic_table_matches_host_generator(fmt):
    from scone_amd.hip_backend import SconeTable
    n, d = 50_000, 768
    t = SconeTable(3, n, d, fmt)
    t.fill_synthetic(7, 0.02 / 127)
    ids = np.array([0, 1, 2, 777, n - 1], dtype=np.int64)
    got = t.gather_rows(torch.from_numpy(ids)).cpu().numpy()
    q = R.synth_rows_i8(7, ids, d).astype(np.float32)
    s = R.synth_scale_f16(7, ids, 0.02 / 127).astype(np.float32)
    expect = q * s[:, None]
    if fmt == "fp16":
        expect = expect.astype(np.float16).astype(np.float32)
    assert np.array_equal(got, expect)


def test_sharded_cache_world1_equals_plain_cache():
    """ShardedEmbeddingCache on one rank (no collective) goes through embed_partial + finalize."""
    from scone_amd import EmbeddingCache
    from scone_amd.distributed import ShardedEmbeddingCache
    rng = np.random.default_rng(8)
    vocab, n, d = 37, 800, 1024
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, 3)
    plain = EmbeddingCache.from_synthetic(ex, d, table_format="int8", seed=3)
    sharded = ShardedEmbeddingCache.from_synthetic(ex, d, table_format="int8", seed=3, rank=0, world=1)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(3, 41)))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((41, d)).astype(np.float32)).half().cuda()
    a = plain.embed_tokens(tok, wte=wte, wpe=wpe)
    b = sharded.embed_tokens(tok, wte=wte, wpe=wpe)
    assert torch.equal(a, b)


# ------------------------------------------------------------------ BASELINE.json full size
@pytest.mark.parametrize("fmt,d", [("int8", 768), ("fp16", 768), ("int4", 1024), ("int8", 1024), ("int8", 1280)])
def test_full_batch_high_occupancy_kernel_equals_the_per_token_position_kernel(fmt, d):
    """The bench's full batch (2048 x 512 = 1M tokens, 1M-row table) through the two instantiations of k_embed_wave that a
    lookup can take: default positions -> the high-occupancy variant (position row in LDS, 7-8 waves / SIMD); the same
    positions passed explicitly -> the per-token-position kernel (round-1 register layout).  Size-independent property at
    BASELINE's full size: the outputs are identical bit for bit, for fp16 and fp32 output, and the one-launch kernel agrees
    on the first 32k tokens."""
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    keys, lens = S.make_keys(1_000_000, S.GPT2_VOCAB, 3, seed=11)
    ex = _extractor(keys, lens, 3)
    cache = EmbeddingCache.from_synthetic(ex, d, table_format=fmt, seed=7, base_scale=0.02 / 127)
    B, T = 2048, 512
    tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to("cuda", torch.int32)
    g = torch.Generator().manual_seed(1)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g) * 0.02).half().cuda()
    wpe = (torch.randn(T, d, generator=g) * 0.01).half().cuda()
    pos = torch.arange(T, dtype=torch.int32, device="cuda").unsqueeze(0).expand(B, T).contiguous()
    a = cache.embed_tokens(tok, wte=wte, wpe=wpe)
    b = cache.embed_tokens(tok, wte=wte, wpe=wpe, position_ids=pos)
    assert torch.equal(a, b)
    small = cache.embed_tokens(tok[:64], wte=wte, wpe=wpe)                # 32768 tokens: match + gather in one launch
    assert torch.equal(small, a[:64])
    del b, small
    a32 = cache.embed_tokens(tok[:512], wte=wte.float(), wpe=wpe.float(), out_dtype=torch.float32)
    b32 = cache.embed_tokens(tok[:512], wte=wte.float(), wpe=wpe.float(), position_ids=pos[:512], out_dtype=torch.float32)
    assert torch.equal(a32, b32)
    assert cache.table.status() == 0


@pytest.mark.parametrize("fmt,d,n_rows", [("int8", 768, 1_000_000), ("fp16", 768, 1_000_000), ("int4", 1024, 1_000_000)])
def test_full_size_table_spot_check_vs_oracle(fmt, d, n_rows):
    """Headline-size table (1M rows, synthetic on the GPU): ids bit-exact and embeddings within
    1e-3 rel against the oracle, which recomputes only the referenced rows from the counter-based
    generator; plus size-independent properties (sum == K * mean, determinism)."""
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    keys, lens = S.make_keys(n_rows, S.GPT2_VOCAB, 3, seed=11)
    ex = _extractor(keys, lens, 3)
    cache = EmbeddingCache.from_synthetic(ex, d, table_format=fmt, seed=7, base_scale=0.02 / 127)
    nk, cap, dups = cache.table.index_stats()
    assert nk == n_rows and dups == 0
    B, T = 16, 512
    tok_np = np.concatenate([S.stream_uniform_ids(keys, lens, B // 2, T, 99), S.stream_zipf(S.GPT2_VOCAB, B // 2, T, 98)])
    tok = torch.from_numpy(tok_np)
    off, ids = cache.match(tok)
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok_np, 3))
    assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(ids.cpu().numpy(), ri)
    # oracle table restricted to the referenced rows
    uniq, inv = np.unique(ri, return_inverse=True)
    if fmt == "int4":
        rows = cache.table.gather_rows(torch.from_numpy(uniq)).cpu().numpy()      # generator twin exists for i8 only
        words = R.hash32((R.hash32(uniq.astype(np.uint32)) ^ np.uint32(7))[:, None] + np.arange(d // 8, dtype=np.uint32)[None, :])
        nib = words.view(np.uint8).reshape(len(uniq), d // 2)
        q = np.empty((len(uniq), d), dtype=np.float32)
        q[:, 0::2], q[:, 1::2] = (nib & 0xF).astype(np.float32) - 8, (nib >> 4).astype(np.float32) - 8
        ng = d // 128
        sc = R.synth_scale_f16(7, (uniq[:, None] * ng + np.arange(ng)[None, :]).reshape(-1), 0.02 / 127).astype(np.float32)
        expect = q * np.repeat(sc.reshape(len(uniq), ng), 128, axis=1)
        assert np.array_equal(rows, expect)
    else:
        expect = R.synth_rows_i8(7, uniq, d).astype(np.float32) * R.synth_scale_f16(7, uniq, 0.02 / 127).astype(np.float32)[:, None]
        if fmt == "fp16":
            expect = expect.astype(np.float16).astype(np.float32)
    ref = R.embed_numpy(expect, ro, inv, "mean").reshape(B, T, d)
    out = cache.embed_tokens(tok, out_dtype=torch.float32)
    assert np.array_equal(out.cpu().numpy(), ref)
    g = torch.Generator().manual_seed(1)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g) * 0.02).half().cuda()
    wpe = (torch.randn(T, d, generator=g) * 0.01).half().cuda()
    fused = cache.embed_tokens(tok, wte=wte, wpe=wpe, check=True)
    want = R.combine(tok, torch.from_numpy(ref), wte.float().cpu(), wpe.float().cpu()).numpy()
    assert _rel(fused.float().cpu().numpy(), want) < REL_TOL
    # properties: sum == K * mean (within fp32 rounding); repeat run identical
    s = cache.embed_tokens(tok, reduce="sum", out_dtype=torch.float32).cpu().numpy().reshape(B * T, d)
    k = np.diff(ro).astype(np.float32)[:, None]
    np.testing.assert_allclose(s, ref.reshape(B * T, d) * np.maximum(k, 1), rtol=2e-6, atol=1e-9)
    assert torch.equal(fused, cache.embed_tokens(tok, wte=wte, wpe=wpe))


def test_plain_c_client(tmp_path):
    """The ABI is usable from plain C (no Python/torch in the loop): tests/cabi_smoke.c."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cabi_smoke")
    lib = os.path.join(root, "scone_amd", "csrc")
    subprocess.run(["gcc", os.path.join(root, "tests", "cabi_smoke.c"), "-I" + os.path.join(root, "include"),
                    "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib, "-lscone_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "cabi_smoke ok: 16 hits" in r.stdout


def test_plain_c_client_two_threads_on_a_staged_handle(tmp_path):
    """Round 5: SURVEY 8b "lookups are thread-safe and stream-ordered" through the bare ABI -- tests/cabi_threads.c runs two
    pthreads, each on its own HIP stream, on ONE pinned-host handle with a staging pipeline (scone_embed of 8 chunks +
    scone_embed_prefetch), every result compared byte for byte with an HBM-resident twin."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cabi_threads")
    lib = os.path.join(root, "scone_amd", "csrc")
    subprocess.run(["gcc", os.path.join(root, "tests", "cabi_threads.c"), "-I" + os.path.join(root, "include"),
                    "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib, "-lscone_hip", "-L/opt/rocm/lib",
                    "-lamdhip64", "-lpthread", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "cabi_threads ok: 2 threads" in r.stdout


@pytest.mark.parametrize("hot_rows", [0, 137, 10_000])
@pytest.mark.parametrize("fmt", ["int8", "fp32"])
def test_pinned_host_placement_matches_hbm(fmt, hot_rows):
    """Rows in mapped pinned host DRAM (optionally with the head of the table in HBM) give the same
    bits as the HBM-resident table; uploads and fp32 stores that straddle the boundary included."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(31)
    vocab, n, d = 41, 900, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(4, 64)))
    outs = []
    for placement, hot in (("hbm", 0), ("pinned_host", hot_rows)):
        t = SconeTable(3, n, d, fmt, placement=placement, hot_rows=hot)
        t.index_build(keys, lens)
        t.store_f32(torch.from_numpy(table[:500]), row0=0)          # straddles the hot/cold boundary
        t.store_f32(torch.from_numpy(table[500:]), row0=500)
        outs.append((t.embed(tok, out_dtype=torch.float32), t.gather_rows(torch.arange(n))))
        if fmt == "fp32":                                             # raw upload path, split at the boundary
            t2 = SconeTable(3, n, d, fmt, placement=placement, hot_rows=hot)
            t2.index_build(keys, lens)
            t2.upload(table, row0=0)
            assert torch.equal(t2.gather_rows(torch.arange(n)), outs[-1][1])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("fmt,d,max_n", [("int8", 768, 3), ("int8", 1024, 4), ("int4", 1024, 3), ("fp16", 768, 4),
                                         ("fp32", 1024, 3), ("int8", 768, 4), ("int8", 1280, 4), ("fp16", 1280, 3),
                                         ("int4", 1280, 3), ("int8", 2048, 3), ("fp16", 4096, 4), ("int4", 2048, 3),
                                         ("fp32", 136, 3), ("int8", 48, 4), ("fp32", 100, 2)])
def test_wave_kernel_shape_sweep(fmt, d, max_n, lookup_form):
    """Every (B, T) geometry of the wave kernel -- T not a multiple of 4, T < max_n, one sequence, many short
    sequences, explicit and default position ids, all output dtypes -- bit-exact in fp32 against the oracle."""
    rng = np.random.default_rng(1000 + d + max_n)
    vocab, n = 13, 500
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    from scone_amd import EmbeddingCache
    ex = _extractor(keys, lens, max_n)
    cache = EmbeddingCache.from_synthetic(ex, d, table_format=fmt, seed=5, base_scale=0.01)
    deq = cache.table.gather_rows(torch.arange(n)).cpu().numpy()
    wte32 = rng.standard_normal((vocab, d)).astype(np.float32)
    wpe32 = rng.standard_normal((600, d)).astype(np.float32)
    for B, T in ((1, 1), (1, 2), (1, 3), (2, 5), (3, 7), (1, 513), (9, 130), (257, 4), (64, 9), (5, 1)):
        tok = rng.integers(0, vocab + 1, size=(B, T))          # vocab itself never matches
        tokc = np.minimum(tok, vocab - 1)
        ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
        fg = R.embed_numpy(deq, ro, ri, "mean").reshape(B, T, d)
        out = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
        assert np.array_equal(out, fg), (B, T)
        pos = rng.integers(0, 600, size=(B, T))
        for position_ids in (None, torch.from_numpy(pos)):
            ref = R.combine(torch.from_numpy(tokc), torch.from_numpy(fg), torch.from_numpy(wte32),
                            torch.from_numpy(wpe32), position_ids=position_ids).numpy()
            got = cache.embed_tokens(torch.from_numpy(tokc), wte=torch.from_numpy(wte32).cuda(),
                                     wpe=torch.from_numpy(wpe32).cuda(), position_ids=position_ids,
                                     out_dtype=torch.float32, check=True)
            # tokens equal to `vocab` were clamped for wte but must still not match any f-gram: compare where equal
            same = (tok == tokc).all(axis=1)
            assert np.array_equal(got.cpu().numpy()[same], ref[same]), (B, T, position_ids is None)
        h16 = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float16).float().cpu().numpy()
        assert _rel(h16, fg) < REL_TOL if np.abs(fg).max() > 0 else not h16.any()


@pytest.mark.parametrize("fmt", ["int8", "int4", "fp16"])
def test_native_table_file_round_trip(tmp_path, fmt):
    """save_native / load_native: quantised rows + scales + keys restored bit for bit, no re-quantisation."""
    from scone_amd import EmbeddingCache
    rng = np.random.default_rng(77)
    vocab, n, d = 23, 600, 1024
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = _cache(keys, lens, 3, rng.standard_normal((n, d)).astype(np.float32), fmt)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(3, 50)))
    want = cache.embed_tokens(tok, out_dtype=torch.float32)
    p = str(tmp_path / "table")
    cache.save_native(p)
    for placement, hot in (("hbm", 0), ("pinned_host", 100)):
        again = EmbeddingCache.load_native(p, placement=placement, hot_rows=hot)
        assert again.table_format == fmt and again.embedding_dim == d and len(again.n_gram_extractor) == n
        assert torch.equal(again.embed_tokens(tok, out_dtype=torch.float32), want)
        assert torch.equal(again.table.gather_rows(torch.arange(n)), cache.table.gather_rows(torch.arange(n)))


# ------------------------------------------------------------------ vocabulary construction
def test_fit_gpu_matches_reference_order(golden_dir):
    """scone_fit reproduces the reference's fit: same f-grams, same ids (Counter.most_common order with
    first-seen ties), on every golden corpus (max_n 1..4, tiny and GPT-2-sized vocabularies)."""
    from scone_amd import NGramExtractor
    z = np.load(os.path.join(golden_dir, "match.npz"))
    for c in z["cases"]:
        flat, cl = z[f"{c}_corpus_flat"], z[f"{c}_corpus_lens"]
        min_freq, max_f = (int(x) for x in z[f"{c}_fit_args"])
        corpus, p = [], 0
        for n in cl:
            corpus.append(flat[p:p + n].tolist())
            p += n
        ex = NGramExtractor(max_n=int(z[f"{c}_max_n"]), min_freq=min_freq, max_f_grams=max_f).fit_gpu(corpus, verbose=False)
        keys, lens = ex.key_arrays()
        assert np.array_equal(lens, z[f"{c}_lens"]), c
        assert np.array_equal(keys, z[f"{c}_keys"]), c


def test_fit_gpu_large_corpus_vs_host_fit():
    from scone_amd import NGramExtractor
    from scone_amd import synthetic as S
    rng = np.random.default_rng(4)
    cdf = S.zipf_cdf(5000)
    corpus = [S.zipf_tokens(rng, cdf, int(rng.integers(1, 600))).tolist() for _ in range(700)] + [[], [3]]
    for max_n, min_freq, max_f in ((3, 2, 50_000), (4, 1, 20_000), (2, 5, 10**9)):
        host = NGramExtractor(max_n=max_n, min_freq=min_freq, max_f_grams=max_f).fit(corpus, verbose=False)
        dev = NGramExtractor(max_n=max_n, min_freq=min_freq, max_f_grams=max_f).fit_gpu(corpus, verbose=False)
        hk, hl = host.key_arrays()
        dk, dl = dev.key_arrays()
        assert np.array_equal(hl, dl) and np.array_equal(hk, dk), (max_n, min_freq)
    with pytest.raises(ValueError):
        NGramExtractor(max_n=2, min_freq=1).fit_gpu([[1, -2, 3]], verbose=False)


# ------------------------------------------------------------------ the paper's lookup
@pytest.mark.parametrize("fmt,d,max_n", [("fp32", 768, 3), ("int8", 1024, 4), ("int8", 768, 2)])
def test_paper_mode_longest_suffix_vs_oracle(fmt, d, max_n, lookup_form):
    """lookup_mode='longest_suffix' (paper, Algorithm 2): the longest f-gram of length >= 2 ending at the token
    replaces the token embedding; otherwise wte; + wpe.  Bit-exact in fp32 against oracle.paper_embed."""
    from scone_amd import EmbeddingCache
    rng = np.random.default_rng(11 + max_n)
    vocab, n = 9, 400
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    table = rng.standard_normal((n, d)).astype(np.float32)
    cache = EmbeddingCache(ex, d, table_format=fmt, lookup_mode="longest_suffix")
    cache.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
    deq = cache.table.gather_rows(torch.arange(n)).cpu().numpy()
    f2id = R._key_dict(keys, lens)
    wte = rng.standard_normal((vocab, d)).astype(np.float32)
    wpe = rng.standard_normal((64, d)).astype(np.float32)
    for B, T in ((1, 1), (2, 7), (5, 64), (33, 3), (600, 64)):     # the last one is past the one-launch limit: two kernels
        tok = rng.integers(0, vocab, size=(B, T))
        ref = R.paper_embed(f2id, max_n, tok, deq, wte=wte, wpe=wpe)
        got = cache.embed_tokens(torch.from_numpy(tok), wte=torch.from_numpy(wte).cuda(), wpe=torch.from_numpy(wpe).cuda(),
                                 out_dtype=torch.float32)
        assert np.array_equal(got.cpu().numpy(), ref), (B, T)
        only = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
        assert np.array_equal(only, R.paper_embed(f2id, max_n, tok, deq))
    # the mode needs the wave kernel's dims
    bad = EmbeddingCache(ex, 100, table_format="fp32", lookup_mode="longest_suffix")      # needs d % 8 == 0
    bad.cache_embeddings(list(range(n)), torch.zeros(n, 100), verbose=False)
    with pytest.raises(ValueError):
        bad.embed_tokens(torch.zeros((1, 4), dtype=torch.int64))


# ------------------------------------------------------------------ engine glue
@pytest.mark.parametrize("mode", ["cover", "longest_suffix"])
def test_engine_generation_consumes_fgram_embeddings(mode):
    """Greedy decoding through SconeInferenceEngine equals a naive loop over SconeLanguageModel.forward with the
    cache attached (the call the reference's forward makes); in causal (paper) mode the incremental
    KV-cache path gives the same tokens as full recomputation."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from scone_amd import EmbeddingCache, SconeLanguageModel
    from scone_amd.inference import SconeInferenceEngine
    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    vocab, H, n = 61, 768, 500
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, 3)
    cache = EmbeddingCache(ex, H, table_format="fp32", lookup_mode=mode)
    cache.cache_embeddings(list(range(n)), torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32) * 0.5), verbose=False)
    base = GPT2LMHeadModel(GPT2Config(vocab_size=vocab, n_positions=64, n_embd=H, n_layer=2, n_head=4)).eval()
    model = SconeLanguageModel(base, None, cache).cuda().eval()
    engine = SconeInferenceEngine(model, tokenizer=None, f_gram_tokenizer=None, embedding_cache=cache)
    prompt = torch.from_numpy(rng.integers(0, vocab, size=(2, 6)))
    got = engine.generate_ids(prompt, max_length=18, do_sample=False)
    ids = prompt.cuda()
    with torch.no_grad():
        while ids.shape[1] < 18:
            logits = model(input_ids=ids)["logits"][:, -1, :].float()
            ids = torch.cat([ids, logits.argmax(-1, keepdim=True)], dim=1)
    assert torch.equal(got, ids)
    # the f-gram table matters: with an all-zero table the continuation differs
    zero = EmbeddingCache(ex, H, table_format="fp32", lookup_mode=mode)
    zero.cache_embeddings(list(range(n)), torch.zeros(n, H), verbose=False)
    plain = SconeInferenceEngine(SconeLanguageModel(base, None, zero).cuda().eval(), embedding_cache=zero)
    assert not torch.equal(plain.generate_ids(prompt, max_length=18, do_sample=False), got)
    out = engine.generate(prompt[0].tolist(), max_length=10, do_sample=True, top_k=5, top_p=0.9)
    assert len(out) == 1 and len(out[0]) == 10
    stats = engine.benchmark_inference(prompt[0].tolist(), max_length=10, num_runs=2, warmup_runs=1)
    assert stats["tokens_per_second"] > 0


def test_workspaces_of_many_streams_are_recycled():
    """A handle keeps one large-batch workspace per stream, at most 16: a process that keeps creating streams recycles
    the least recently used idle one instead of growing by 2 MB per stream; results stay exact and device memory flat."""
    rng = np.random.default_rng(33)
    vocab, n, d = 41, 1200, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = _cache(keys, lens, 3, rng.standard_normal((n, d)).astype(np.float32), "int8")
    tok = torch.from_numpy(rng.integers(0, vocab, size=(128, 512))).to("cuda", torch.int32)      # 64k tokens: two-kernel form
    want = cache.embed_tokens(tok, out_dtype=torch.float32).clone()
    torch.cuda.synchronize()
    free0 = None
    got = torch.empty_like(want)                  # one output buffer: torch's allocator keeps a pool PER STREAM otherwise
    for i in range(48):
        st = torch.cuda.Stream()
        got.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            cache.embed_tokens(tok, out_dtype=torch.float32, out=got)
        st.synchronize()
        assert torch.equal(got, want), i
        if i == 20:
            free0 = torch.cuda.mem_get_info()[0]
    assert free0 - torch.cuda.mem_get_info()[0] < 16 << 20          # 27 more streams, no growth beyond allocator noise


def test_cu_reserve_runs_the_lookup_on_a_masked_stream_with_identical_results():
    """`scone_set_cu_reserve`: the large-batch lookup kernels go to a stream of the handle whose CU mask leaves R compute
    units free, tied into the caller's stream by two events.  Same bytes out for R = 0 / 8 / 64, on the default stream and on a
    side stream, work queued before and after the call stays ordered around it; bad values are refused."""
    from scone_amd import EmbeddingCache
    rng = np.random.default_rng(17)
    vocab, n, d, max_n = 53, 3000, 768, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    n = len(ex)
    cache = EmbeddingCache(ex, d, table_format="int8")
    cache.cache_embeddings(list(range(n)), torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)), verbose=False)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(160, 512))).to("cuda", torch.int32)      # above the one-launch limit
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((512, d)).astype(np.float32)).half().cuda()
    t = cache.table
    want = cache.embed_tokens(tok, wte=wte, wpe=wpe).clone()
    assert t.cu_reserve()[0] == 0 and t.cu_reserve()[1] >= 64
    for r in (8, 64, 0, 16):
        t.set_cu_reserve(r)
        assert t.cu_reserve()[0] == r
        out = torch.zeros_like(want)
        out.fill_(7.0)                                         # queued BEFORE the lookup on the same stream: must not win
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
        chk = out.float().sum()                                # queued AFTER: must see the lookup's output
        assert torch.equal(out, want) and float(chk) == float(want.float().sum()), r
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out2 = cache.embed_tokens(tok, wte=wte, wpe=wpe)
        side.synchronize()
        assert torch.equal(out2, want), r
    for bad in (-8, 12, 100000):
        with pytest.raises(ValueError):
            t.set_cu_reserve(bad)
    assert t.status() == 0


def test_staging_buffer_overflow_is_flagged_and_never_reads_out_of_bounds(monkeypatch):
    """k_stage_place's "no evictable slot" path.  The cache of cold rows is sized for the worst case of the chunks in flight,
    so the path is unreachable from scone_embed; the test hook SCONE_STAGE_CAP_ROWS shrinks the cache to 8 rows while a chunk
    references hundreds of distinct cold rows.  Every reference that found no slot must have been redirected INSIDE the cache
    (a PENDING tag left in the slot map would send the lookup 4G rows past it) and the call must be flagged: status bit 3."""
    from scone_amd import EmbeddingCache
    from scone_amd import _lib as L
    rng = np.random.default_rng(91)
    vocab, n, d, max_n = 41, 2500, 768, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()
    n = len(lens)
    table = rng.standard_normal((n, d)).astype(np.float32)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(6, 64)))
    ref = EmbeddingCache(ex, d, table_format="int8")
    ref.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
    want = ref.embed_tokens(tok, out_dtype=torch.float32)
    monkeypatch.setenv("SCONE_STAGE_CAP_ROWS", "8")
    tiny = EmbeddingCache(ex, d, table_format="int8", placement="pinned_host", hot_rows=16, stage_tokens=128)
    tiny.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
    got = tiny.embed_tokens(tok, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert tiny.table.status() & L.ST_STAGE_OVERFLOW
    assert bool(torch.isfinite(got).all())
    # second call: the flag is sticky per call, not stuck
    monkeypatch.delenv("SCONE_STAGE_CAP_ROWS")
    ok = EmbeddingCache(ex, d, table_format="int8", placement="pinned_host", hot_rows=16, stage_tokens=128)
    ok.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
    assert torch.equal(ok.embed_tokens(tok, out_dtype=torch.float32), want) and not ok.table.status() & L.ST_STAGE_OVERFLOW


def _oracle_inputs_embeds(mode, keys, lens, max_n, table, ids, wte, wpe):
    """``inputs_embeds [B, T, H]`` fp32 as the reference computes them, from the oracle alone (no scone_amd lookup):
    cover: match (n_gram_extractor.py:106-126) -> mean of the rows (engine.py:247-259) -> wte + fg + wpe
    (language_model.py:239-254); longest_suffix: the paper's Algorithm 2."""
    tok = ids.cpu().numpy()
    B, T = tok.shape
    if mode == "cover":
        ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
        fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(B, T, -1))
        return R.combine(ids.cpu(), fg, wte, wpe)
    return torch.from_numpy(R.paper_embed(R._key_dict(keys, lens), max_n, tok, table, wte.numpy(), wpe.numpy()))


@pytest.mark.parametrize("mode", ["cover", "longest_suffix"])
def test_engine_greedy_decoding_equals_hf_gpt2_driven_by_the_oracle(mode):
    """SURVEY 8f rank 3, anchored to the oracle: every greedy token SconeInferenceEngine.generate_ids emits (KV-cache
    decoding, roll-back in cover mode) equals the token of an HF GPT-2 that is handed, at every step and for the whole
    prefix, ``inputs_embeds`` built by the ORACLE (R.combine(R.embed_numpy(...)) / R.paper_embed) -- the engine's lookup is
    not on that side of the comparison.  Step-0 logits agree within 1e-3."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from scone_amd import EmbeddingCache, SconeLanguageModel
    from scone_amd.inference import SconeInferenceEngine
    torch.manual_seed(1)
    rng = np.random.default_rng(4)
    vocab, H, n, max_n = 61, 768, 500, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()                                     # de-duplicated: smallest id wins
    table = (rng.standard_normal((len(lens), H)) * 0.5).astype(np.float32)
    cache = EmbeddingCache(ex, H, table_format="fp32", lookup_mode=mode)
    cache.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
    base = GPT2LMHeadModel(GPT2Config(vocab_size=vocab, n_positions=64, n_embd=H, n_layer=2, n_head=4)).eval()
    model = SconeLanguageModel(base, None, cache).cuda().eval()
    engine = SconeInferenceEngine(model, embedding_cache=cache)
    wte, wpe = base.transformer.wte.weight.detach().float().cpu(), base.transformer.wpe.weight.detach().float().cpu()
    prompt = torch.from_numpy(rng.integers(0, vocab, size=(2, 7)))
    L = 20
    got = engine.generate_ids(prompt, max_length=L, do_sample=False).cpu()
    ids = prompt.clone()
    with torch.no_grad():
        first = True
        while ids.shape[1] < L:
            x = _oracle_inputs_embeds(mode, keys, lens, max_n, table, ids, wte, wpe).cuda()
            h = base.transformer(inputs_embeds=x, return_dict=True).last_hidden_state[:, -1, :]
            logits = base.lm_head(h).float()
            if first:                                               # the engine's own first step, same prefix
                mine = base.lm_head(base.transformer(inputs_embeds=engine.embed(ids.cuda()), return_dict=True)
                                    .last_hidden_state[:, -1, :]).float()
                assert float((mine - logits).abs().max()) <= 1e-3 * float(logits.abs().max())
                first = False
            ids = torch.cat([ids, logits.argmax(-1, keepdim=True).cpu()], dim=1)
    assert torch.equal(got, ids)


def test_engine_beam_search_and_repetition_penalty_vs_oracle_scoring():
    """generate(num_beams=, repetition_penalty=) (engine.py:192-204 passes both to HF generate): beam search keeps the
    num_beams best prefixes by summed log-probability, checked against an independent beam search in this test whose
    scoring model is HF GPT-2 fed ORACLE inputs_embeds; with one beam it is greedy decoding; the repetition penalty
    follows HF's processor (seen logits / penalty if positive, * penalty if negative)."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from scone_amd import EmbeddingCache, SconeLanguageModel
    from scone_amd.inference import SconeInferenceEngine
    torch.manual_seed(3)
    rng = np.random.default_rng(14)
    vocab, H, n, max_n = 23, 768, 200, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()
    table = (rng.standard_normal((len(lens), H)) * 0.5).astype(np.float32)
    cache = EmbeddingCache(ex, H, table_format="fp32")
    cache.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
    base = GPT2LMHeadModel(GPT2Config(vocab_size=vocab, n_positions=32, n_embd=H, n_layer=2, n_head=4)).eval()
    engine = SconeInferenceEngine(SconeLanguageModel(base, None, cache).cuda().eval(), embedding_cache=cache)
    wte, wpe = base.transformer.wte.weight.detach().float().cpu(), base.transformer.wpe.weight.detach().float().cpu()

    def oracle_logp(seqs, penalty=1.0):
        ids = torch.tensor(seqs)
        x = _oracle_inputs_embeds("cover", keys, lens, max_n, table, ids, wte, wpe).cuda()
        with torch.no_grad():
            lg = base.lm_head(base.transformer(inputs_embeds=x, return_dict=True).last_hidden_state[:, -1, :]).float().cpu()
        if penalty != 1.0:
            for r, sq in enumerate(seqs):
                for t in set(sq):
                    lg[r, t] = lg[r, t] / penalty if lg[r, t] > 0 else lg[r, t] * penalty
        return torch.log_softmax(lg, dim=-1)

    prompt, L, nb = [3, 7, 7, 1], 10, 3
    for penalty in (1.0, 1.7):
        beams, scores = [list(prompt)], [0.0]                            # the test's own beam search
        while len(beams[0]) < L:
            lp = oracle_logp(beams, penalty)
            cand = sorted(((scores[b] + float(lp[b, t]), beams[b] + [t]) for b in range(len(beams)) for t in range(vocab)),
                          key=lambda c: -c[0])[:nb]
            scores, beams = [c[0] for c in cand], [c[1] for c in cand]
        seqs, sc = engine.beam_search_ids(torch.tensor([prompt]), max_length=L, num_beams=nb, num_return_sequences=nb,
                                          repetition_penalty=penalty)
        assert [q.tolist() for q in seqs] == beams
        assert np.allclose(sc, [x / L for x in scores], rtol=1e-4, atol=1e-4)
        one, _ = engine.beam_search_ids(torch.tensor([prompt]), max_length=L, num_beams=1, repetition_penalty=penalty)
        greedy = engine.generate_ids(torch.tensor([prompt]), max_length=L, do_sample=False, repetition_penalty=penalty)
        assert one[0].tolist() == greedy[0].tolist()
    out = engine.generate(prompt, max_length=L, num_beams=nb, num_return_sequences=2)
    assert len(out) == 2 and len(out[0]) == L


@pytest.mark.parametrize("use_mm", [False, True])
def test_engine_from_pretrained_reads_a_reference_checkpoint_layout(tmp_path, use_mm):
    """SconeInferenceEngine.from_pretrained (engine.py:129-190): config.json + weights under the reference's module names
    (base_model.* = GPT2LMHeadModel, f_gram_projection.weight, f_gram_model.* ignored), n_gram_extractor.npy, the
    reference-format embedding_cache.npy with rows of the f-gram model's size.  The projection is folded into the table;
    the engine's inputs_embeds equal wte + proj(mean(rows)) + wpe computed from the oracle."""
    import json
    import sys
    from safetensors.torch import save_file
    from transformers import GPT2Config, GPT2LMHeadModel
    from scone_amd import EmbeddingCache
    from scone_amd.inference import SconeInferenceEngine
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stub_tokenizer import StubTokenizer
    torch.manual_seed(2)
    rng = np.random.default_rng(6)
    vocab, H, df, n, max_n = 61, 768, 384, 300, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(2, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    _, first = np.unique(np.concatenate([keys, lens[:, None]], axis=1), axis=0, return_index=True)
    keys, lens = keys[np.sort(first)], lens[np.sort(first)]          # distinct f-grams: dense ids, as the reference's fit assigns them
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()
    assert len(ex.f_gram_to_id) == len(lens)
    d = str(tmp_path / "ckpt")
    os.makedirs(d)
    json.dump({"model_type": "scone", "vocab_size": vocab, "hidden_size": H, "num_hidden_layers": 2, "num_attention_heads": 4,
               "max_position_embeddings": 64, "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1,
               "layer_norm_eps": 1e-5, "use_f_gram_embeddings": True}, open(os.path.join(d, "config.json"), "w"))
    base = GPT2LMHeadModel(GPT2Config(vocab_size=vocab, n_positions=64, n_embd=H, n_layer=2, n_head=4, layer_norm_epsilon=1e-5)).eval()
    W = (torch.randn(H, df) * 0.05)
    state = {"base_model." + k: v.clone().contiguous() for k, v in base.state_dict().items() if k != "lm_head.weight"}   # tied: saved once
    state["f_gram_projection.weight"] = W
    state["f_gram_model.embeddings.word_embeddings.weight"] = torch.zeros(4, 4)     # the f-gram BERT: not needed to serve a cache
    save_file(state, os.path.join(d, "model.safetensors"))
    ex.save(os.path.join(d, "n_gram_extractor"))
    table = (rng.standard_normal((len(lens), df)) * 0.5).astype(np.float32)
    # the cache file in both of the reference's forms: rows inside the .npy, or a raw memory-mapped [N, d] file beside it
    host = EmbeddingCache(ex, df, cache_dir=str(tmp_path / "mm") if use_mm else None, use_memory_map=use_mm)
    host.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
    host.save(os.path.join(d, "embedding_cache"))
    engine = SconeInferenceEngine.from_pretrained(d, tokenizer=StubTokenizer(vocab), use_memory_map=use_mm)
    assert engine.embedding_cache.embedding_dim == H and engine.max_n == max_n
    ids = torch.from_numpy(rng.integers(2, vocab, size=(2, 11)))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, ids.numpy(), max_n))
    fg = torch.from_numpy(R.embed_numpy(table, ro, ri, "mean").reshape(2, 11, df))
    wte, wpe = base.transformer.wte.weight.detach().float(), base.transformer.wpe.weight.detach().float()
    ref = R.combine(ids, torch.nn.functional.linear(fg, W), wte, wpe)               # language_model.py:235-254 as written
    got = engine.embed(ids.cuda()).float().cpu()
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6   # proj(mean) vs mean(proj): fp32 rounding
    gen = engine.generate_ids(engine._encode("alpha beta gamma delta"), max_length=9, do_sample=False)   # text -> the loaded tokenizer
    assert tuple(gen.shape) == (1, 9)


def test_reference_written_files_give_reference_lookups_on_the_gpu(golden_dir):
    """the extractor and cache files the REFERENCE wrote (tests/golden/tiny_extractor.npy, tiny_cache.npy,
    n_gram_extractor.py:128-165, embedding_cache.py:183-243) loaded through this package's load() and looked up on the
    GPU give the reference's own outputs for that table (lookup.npz case c4, captured from the reference)."""
    from scone_amd import EmbeddingCache, NGramExtractor
    z = np.load(os.path.join(golden_dir, "lookup.npz"))
    ex = NGramExtractor.load(os.path.join(golden_dir, "tiny_extractor.npy"))
    cache = EmbeddingCache.load(os.path.join(golden_dir, "tiny_cache.npy"), ex, use_memory_map=False)
    c = "c4"
    tok = torch.from_numpy(z[f"{c}_tok"])[None, :]
    off, ids = cache.match(tok)
    assert np.array_equal(off.cpu().numpy(), z[f"{c}_off"]) and np.array_equal(ids.cpu().numpy(), z[f"{c}_ids"])
    assert np.array_equal(cache.get_embeddings(z[f"{c}_gather_ids"].tolist()).numpy(), z[f"{c}_gather_out"])
    te = cache.get_token_embeddings(z[f"{c}_tok"].tolist())
    assert sorted(te.keys()) == z[f"{c}_te_positions"].tolist()
    p = 0
    for pos, rows in zip(z[f"{c}_te_positions"], z[f"{c}_te_rows"]):
        assert np.array_equal(te[int(pos)].numpy(), z[f"{c}_te_stacks"][p:p + rows])
        p += rows
    assert np.array_equal(cache.embed_tokens(tok, out_dtype=torch.float32).cpu().numpy(), z[f"{c}_agg_f32"])
    assert np.array_equal(cache.embed_tokens(tok, out_dtype=torch.float16).cpu().numpy().view(np.uint16),
                          z[f"{c}_agg_f16"].view(np.uint16))
    tfg = ex.get_token_f_grams(z[f"{c}_tok"].tolist())               # the tuples, matched on the GPU
    f2id = ex.f_gram_to_id
    flat = [f2id[g] for pos in range(tok.shape[1]) for g in tfg[pos]]
    assert flat == z[f"{c}_ids"].tolist()


@pytest.mark.parametrize("fmt", ["int8", "int4", "fp16"])
@pytest.mark.parametrize("with_index", [False, True])
def test_native_table_file_round_trip_against_the_oracle(tmp_path, fmt, with_index):
    """save_native -> load_native, checked against the ORACLE on the dequantised table (not against the handle that
    wrote the file): whole table and a shard (rows [row_begin, row_end) only in the file), with and without the persisted
    index (slots + unigram table + presence bitmap copied back instead of rebuilt)."""
    from scone_amd import EmbeddingCache
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(78)
    vocab, n, d, max_n = 29, 900, 1024, 3
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()
    n = len(lens)
    table = rng.standard_normal((n, d)).astype(np.float32)
    deq = {"int8": lambda t: R.dequantize_i8(*R.quantize_i8(t)), "int4": lambda t: R.dequantize_i4(*R.quantize_i4(t)),
           "fp16": lambda t: t.astype(np.float16).astype(np.float32)}[fmt](table)
    tok = rng.integers(0, vocab, size=(3, 40))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
    want = R.embed_numpy(deq, ro, ri, "mean").reshape(3, 40, d)
    cache = EmbeddingCache(ex, d, table_format=fmt)
    cache.cache_embeddings(list(range(n)), torch.from_numpy(table), verbose=False)
    p = str(tmp_path / "table")
    cache.save_native(p, with_index=with_index)
    again = EmbeddingCache.load_native(p)
    assert np.array_equal(again.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy(), want)
    assert again.table.index_stats()[0] == cache.table.index_stats()[0]      # distinct keys (the vocabulary has duplicates)
    # a shard: only its own rows are in the file
    a, b = n // 3, 2 * n // 3
    shard = EmbeddingCache(ex, d, table_format=fmt)
    shard._table = SconeTable(max_n, n, dim=d, table_format=fmt, row_begin=a, row_end=b)
    ex.build_index(shard._table)
    shard._table.store_f32(torch.from_numpy(table[a:b]), row0=a)
    shard._dirty, shard.keep_host_copy = False, False
    ps = str(tmp_path / "shard")
    shard.save_native(ps, with_index=with_index)
    import json
    raw = np.load(ps + ".npy", mmap_mode="r")                         # a valid .npy of bytes; the json header says what is where
    meta = json.loads(bytes(raw[8:8 + int(np.frombuffer(bytes(raw[:8]), dtype=np.uint64)[0])]).decode())
    assert meta["sections"]["rows"]["shape"][0] == b - a and meta["sections"]["scales"]["shape"][0] == b - a
    assert ("index_slots" in meta["sections"]) == with_index and all(o % 4096 == 0 for o in meta["section_offsets"].values())
    del raw
    sh2 = EmbeddingCache.load_native(ps)
    assert (sh2.table.row_begin, sh2.table.row_end) == (a, b)
    partial, counts = sh2.table.embed_partial(torch.from_numpy(tok))
    own = (ri >= a) & (ri < b)
    seg = np.repeat(np.arange(len(ro) - 1), np.diff(ro))
    off_own = np.zeros(len(ro), dtype=np.int64)
    np.cumsum(np.bincount(seg[own], minlength=len(ro) - 1), out=off_own[1:])
    assert np.array_equal(partial.cpu().numpy(), R.embed_numpy(deq, off_own, ri[own], "sum"))
    assert np.array_equal(counts.cpu().numpy(), np.diff(ro))


def _rss_anon_kb():
    for ln in open("/proc/self/status"):
        if ln.startswith("RssAnon:"):
            return int(ln.split()[1])
    return 0


@pytest.mark.parametrize("with_index", [False, True])
def test_native_table_file_streams_a_4gb_shard_with_bounded_host_memory(tmp_path, with_index):
    """save_native / load_native at a shard's scale (round-2 VERDICT, weak #7): the LAST of 4 shards of a 33.6M-row INT4
    d = 1024 table -- 8.4M rows = 4.4 GB of rows + scales (with the index: + 1 GB of slots and bitmap) -- goes to one
    memory-mapped file and back in 256k-row chunks.  The process's ANONYMOUS memory (what is not the file's page cache)
    grows by less than 1 GB on the way out and on the way in -- the round-2 code held the whole shard twice -- the vocabulary
    has no key arrays (StructuredVocab: recorded by its parameters), and the loaded shard gives the oracle's partial sums."""
    import shutil
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    from scone_amd.distributed import ShardedEmbeddingCache
    N, W, d = 33_600_000, 4, 1024
    if shutil.disk_usage(tmp_path).free < 8e9:
        pytest.skip("needs 8 GB of scratch space")
    vocab = S.StructuredVocab(N)
    sh = ShardedEmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, rank=W - 1, world=W,
                                              n_rows=N)
    a, b = sh.row_begin, sh.row_end
    assert (b - a) * 528 > 4.3e9
    shard = EmbeddingCache(vocab, d, table_format="int4", keep_host_copy=False)
    shard._table, shard._dirty = sh.table, False
    p = str(tmp_path / "shard")
    torch.cuda.synchronize()
    before = _rss_anon_kb()
    shard.save_native(p, with_index=with_index)
    grown_save = _rss_anon_kb() - before
    assert os.path.getsize(p + ".npy") > (b - a) * 528
    before = _rss_anon_kb()
    again = EmbeddingCache.load_native(p)
    grown_load = _rss_anon_kb() - before
    assert grown_save < 1_000_000 and grown_load < 1_000_000, (grown_save, grown_load)       # kB
    t2 = again.table
    assert (t2.row_begin, t2.row_end, t2.n_rows) == (a, b, N) and isinstance(again.n_gram_extractor, S.StructuredVocab)
    ids = np.array([a, a + 1, (a + b) // 2, a + (1 << 18) - 1, a + (1 << 18), b - 2, b - 1], dtype=np.int64)   # chunk edges too
    assert np.array_equal(t2.gather_rows(torch.from_numpy(ids)).cpu().numpy(), R.dequantize_i4(*R.synth_rows_i4(7, ids, d, 0.02 / 127)))
    # lookups: partial sums over the rows of this shard equal the oracle's on the recomputed rows
    tok = S.stream_uniform_ids(vocab, None, 4, 512, 3)
    off, mids = t2.match_csr(torch.from_numpy(tok))
    ro, ri = off.cpu().numpy().astype(np.int64), mids.cpu().numpy().astype(np.int64)
    off0, mids0 = sh.table.match_csr(torch.from_numpy(tok))        # the handle that was saved (index built by insertion)
    assert torch.equal(off0, off) and torch.equal(mids0, mids)
    pos = np.repeat(np.arange(len(ro) - 1), np.diff(ro))
    own = (ri >= a) & (ri < b)
    assert own.any()
    uniq, inv = np.unique(ri[own], return_inverse=True)
    rows = R.dequantize_i4(*R.synth_rows_i4(7, uniq, d, 0.02 / 127))
    off_own = np.zeros(len(ro), dtype=np.int64)
    np.cumsum(np.bincount(pos[own], minlength=len(ro) - 1), out=off_own[1:])
    partial, counts = t2.embed_partial(torch.from_numpy(tok))
    assert np.array_equal(partial.cpu().numpy(), R.embed_numpy(rows, off_own, inv, "sum"))
    assert np.array_equal(counts.cpu().numpy(), np.diff(ro))
    os.remove(p + ".npy")


@pytest.mark.parametrize("fmt,d", [("int8", 768), ("int4", 1024), ("fp16", 1280)])
def test_staged_prefetch_matches_hbm(fmt, d):
    """SCONE_PLACE_PINNED_HOST with stage_tokens > 0 (side-stream match + de-duplicated host->HBM staging,
    double-buffered) gives the same bits as the HBM-resident table: several chunks, ragged last chunk,
    repeated calls (generation tags, buffer reuse), table modified between calls."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(55)
    vocab, n, max_n = 37, 3000, 3
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    ref = SconeTable(max_n, n, d, fmt)
    ref.index_build(keys, lens)
    ref.store_f32(torch.from_numpy(table))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32)).half().cuda()
    for hot, stage in ((0, 64), (100, 200), (2999, 40), (500, 10_000)):
        t = SconeTable(max_n, n, d, fmt, placement="pinned_host", hot_rows=hot, stage_tokens=stage)
        t.index_build(keys, lens)
        t.store_f32(torch.from_numpy(table))
        for B, T in ((7, 40), (1, 33), (300, 5), (64, 40)):
            tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
            for _ in range(2):
                assert torch.equal(t.embed(tok, wte=wte, wpe=wpe), ref.embed(tok, wte=wte, wpe=wpe)), (hot, stage, B, T)
            assert torch.equal(t.embed(tok, out_dtype=torch.float32), ref.embed(tok, out_dtype=torch.float32))
        # modify some rows (head and tail of the table), both tables alike
        upd = torch.from_numpy(rng.standard_normal((50, d)).astype(np.float32))
        for tb in (t, ref):
            tb.store_f32(upd[:25], row0=10)
            tb.store_f32(upd[25:], row0=n - 25)
        tok = torch.from_numpy(rng.integers(0, vocab, size=(9, 40)))
        assert torch.equal(t.embed(tok, wte=wte, wpe=wpe), ref.embed(tok, wte=wte, wpe=wpe))
        for tb in (ref,):
            tb.store_f32(torch.from_numpy(table[10:35]), row0=10)
            tb.store_f32(torch.from_numpy(table[n - 25:]), row0=n - 25)


def test_hipgraph_capture_of_the_fused_lookup():
    """scone_embed does no allocation or synchronisation once its workspace exists (scone_reserve), so the
    two-kernel step can be captured in a hipGraph and replayed (decode-size batches are launch-bound)."""
    from scone_amd import EmbeddingCache
    rng = np.random.default_rng(12)
    vocab, n, d = 50, 2000, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = EmbeddingCache.from_synthetic(_extractor(keys, lens, 3), d, table_format="int8", seed=2)
    wte = torch.randn(vocab, d, device="cuda").half()
    wpe = torch.randn(64, d, device="cuda").half()
    tok = torch.from_numpy(rng.integers(0, vocab, size=(8, 64))).to("cuda", torch.int32)
    out = torch.empty(8, 64, d, dtype=torch.float16, device="cuda")
    cache.table.reserve(tok.numel())
    want = cache.embed_tokens(tok, wte=wte, wpe=wpe).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)         # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    tok.copy_(torch.from_numpy(rng.integers(0, vocab, size=(8, 64))))   # new tokens, same graph
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, cache.embed_tokens(tok, wte=wte, wpe=wpe))


def test_abi_argument_errors():
    """Error behaviour of the boundary: negative sizes, wrong dtype, missing table -> exceptions, never a crash."""
    from scone_amd.hip_backend import SconeTable
    t = SconeTable(3, 10, 768, "int8")
    tok = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(KeyError):
        t.embed(tok, out_dtype=torch.float64)
    with pytest.raises(ValueError):
        t.embed(tok, wte=torch.zeros(5, 64, device="cuda"))            # wrong width
    idx_only = SconeTable(3, 10)
    with pytest.raises(Exception, match="no table"):
        idx_only.embed(tok)
    with pytest.raises(ValueError):
        SconeTable(3, 10, 100, "int8")                                  # dim not a multiple of 16
    with pytest.raises(ValueError):
        SconeTable(3, 10, 768, "int8", row_begin=8, row_end=4)
    with pytest.raises(IndexError):
        t.upload(np.zeros((4, 768), dtype=np.int8), np.ones(4, dtype=np.float16), row0=8)   # rows 8..11 of 10
    assert t.status() == 0


@pytest.mark.parametrize("head", [0, 37, 1200])
@pytest.mark.parametrize("fmt,d,max_n,world", [("int8", 768, 3, 3), ("int4", 1024, 4, 8), ("fp16", 1280, 3, 2)])
def test_row_exchange_between_shards_is_bit_exact(fmt, d, max_n, world, head):
    """The slice exchange for row-sharded tables (plan -> pack -> all-to-all of quantised rows -> embed), with the
    W shards living on one GPU and the all-to-all done by hand, default and caller-supplied position ids: every slice equals
    the unsharded table BIT FOR BIT (the receiver reduces the rows in the reference's order), and the wire carries one
    record per DISTINCT row outside the replicated head and destination (`head` rows kept on every shard, stored in two
    calls; 1200 spans more than one shard)."""
    from scone_amd.hip_backend import SconeTable
    from scone_amd.distributed import shard_range
    rng = np.random.default_rng(90 + world)
    vocab, n = 29, 2000
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
        if head:
            s.shard_set_head(head)
            s.shard_head_store_f32(torch.from_numpy(table[:head // 2]), row0=0)      # in two calls
            s.shard_head_store_f32(torch.from_numpy(table[head // 2:head]), row0=head // 2)
        shards.append(s)
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32)).half().cuda()
    rec = shards[0].shard_record_bytes()
    for B, T in ((11, 40), (2, 7), (64, 3), (1, 40)):
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        pos = torch.from_numpy(rng.integers(0, 40, size=(B, T)))
        for position_ids in (None, pos):
            want = full.embed(tok, wte=wte, wpe=wpe, position_ids=position_ids).reshape(B * T, d)
            off, ids = (x.cpu().numpy() for x in full.match_csr(tok))
            bper = (B + world - 1) // world
            ends = [s.shard_gather_plan_chunks(tok, world, dedup_across_chunks=False) for s in shards]   # chunk q = slice q
            cnt = [[e[0]] + [e[q] - e[q - 1] for q in range(1, world)] for e in ends]
            total = 0
            for q in range(world):                                   # one record per distinct row outside the head and slice
                t0, t1 = min(q * bper, B) * T, min(q * bper + bper, B) * T
                need = np.unique(ids[off[t0]:off[t1]])
                total += int((need >= head).sum())
            assert sum(e[-1] for e in ends) == total
            sends = []
            for r, s in enumerate(shards):
                buf = torch.empty((max(ends[r][-1], 1), rec), dtype=torch.uint8, device="cuda")
                s.shard_gather_pack_range(0, ends[r][-1], buf[:ends[r][-1]])
                sends.append(buf)
            for q, s in enumerate(shards):
                recv = torch.cat([sends[r][sum(cnt[r][:q]):sum(cnt[r][:q + 1])] for r in range(world)]).contiguous()
                b0, b1 = min(q * bper, B), min(q * bper + bper, B)
                got = torch.empty((max(b1 - b0, 1) * T, d), dtype=torch.float16, device="cuda")
                s.shard_gather_add_records(recv, 0, recv.shape[0])
                if b1 > b0:
                    s.shard_gather_embed_range(tok, b0, b1, recv, got, wte=wte, wpe=wpe, position_ids=position_ids, out_is_slice=True)
                    bad = (got[:(b1 - b0) * T] != want[b0 * T:b1 * T]).any(dim=1)
                    assert not bool(bad.any()), (B, T, q, int(bad.sum()), bad.nonzero().flatten()[:8].tolist(), s.status())
                assert s.status() == 0


@pytest.mark.parametrize("fmt,d,max_n,world,n,head", [("int8", 768, 3, 4, 60, 0), ("int4", 1024, 3, 8, 2000, 37),
                                                       ("fp16", 768, 4, 3, 500, 0), ("int8", 1024, 3, 8, 24, 5)])
def test_slice_exchange_one_record_per_distinct_row_and_destination(fmt, d, max_n, world, n, head):
    """The round-2 slice exchange (scone_shard_gather_plan_chunks with a claim generation per chunk, chunk q = rank q's
    slice): W shards on one GPU, the all-to-all by hand.  Every destination gets each DISTINCT row it references exactly
    once from its owner (the record counts equal the oracle's per-slice distinct-row counts; tiny tables where every slice
    needs nearly every row exercise the per-destination list capacity), every slice is bit-identical to the unsharded
    table, and the error paths of the new entry points return codes instead of touching memory."""
    from scone_amd.hip_backend import SconeTable
    from scone_amd.distributed import shard_range
    rng = np.random.default_rng(190 + world + n)
    vocab = 7 if n < 100 else 29
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
        if b > a:
            s.store_f32(torch.from_numpy(table[a:b]), row0=a)
        if head:
            s.shard_set_head(head)
            s.shard_head_store_f32(torch.from_numpy(table[:head]), row0=0)
        shards.append(s)
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((40, d)).astype(np.float32)).half().cuda()
    rec = shards[0].shard_record_bytes()
    for B, T in ((16, 40), (3, 7), (64, 3), (1, 40)):
        tok = torch.from_numpy(rng.integers(0, vocab, size=(B, T)))
        want = full.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        off, ids = (x.cpu().numpy() for x in full.match_csr(tok))
        bper = (B + world - 1) // world
        ends = [s.shard_gather_plan_chunks(tok, world, dedup_across_chunks=False) for s in shards]
        cnt = [[e[0]] + [e[q] - e[q - 1] for q in range(1, world)] for e in ends]
        for q in range(world):                                       # oracle: distinct rows of slice q outside the head, per owner
            t0, t1 = min(q * bper, B) * T, min(q * bper + bper, B) * T
            need = np.unique(ids[off[t0]:off[t1]])
            need = need[need >= head]
            for r in range(world):
                a, b = shard_range(n, r, world)
                assert cnt[r][q] == int(((need >= a) & (need < b)).sum()), (B, T, r, q)
        sends = []
        for r, s in enumerate(shards):
            buf = torch.empty((max(ends[r][-1], 1), rec), dtype=torch.uint8, device="cuda")
            s.shard_gather_pack_range(0, ends[r][-1], buf[:ends[r][-1]])
            sends.append(buf)
        for q, s in enumerate(shards):
            parts = [sends[r][sum(cnt[r][:q]):sum(cnt[r][:q + 1])] for r in range(world)]
            recv = torch.cat(parts).contiguous()
            b0, b1 = min(q * bper, B), min(q * bper + bper, B)
            out = torch.full(((b1 - b0) * T + 1, d), 7.0, dtype=torch.float16, device="cuda")
            s.shard_gather_add_records(recv, 0, recv.shape[0])
            if b1 > b0:
                s.shard_gather_embed_range(tok, b0, b1, recv, out, wte=wte, wpe=wpe, out_is_slice=True)
            assert torch.equal(out[:(b1 - b0) * T], want[b0 * T:b1 * T]), (B, T, q)
            assert bool((out[(b1 - b0) * T:] == 7.0).all())            # nothing written past the slice
            assert s.status() == 0
    # error paths
    s = shards[0]
    with pytest.raises((ValueError, IndexError, RuntimeError)):
        s.shard_gather_pack_range(10**9, 5, torch.empty((5, rec), dtype=torch.uint8, device="cuda"))
    with pytest.raises((ValueError, IndexError, RuntimeError)):
        s.shard_gather_plan_chunks(tok, 65)
    fresh = SconeTable(max_n, n, d, fmt, row_begin=0, row_end=n)
    fresh.index_build(keys, lens)
    fresh.shard_gather_plan_chunks(tok, 2)
    with pytest.raises((ValueError, IndexError, RuntimeError)):       # records were never added
        fresh.shard_gather_embed_range(tok, 0, 1, torch.empty((0, rec), dtype=torch.uint8, device="cuda"),
                                       torch.empty((tok.numel(), d), dtype=torch.float16, device="cuda"))
    with pytest.raises((ValueError, IndexError, RuntimeError)):       # another batch shape than the planned one
        s.shard_gather_embed_range(tok[:, :-1].contiguous(), 0, 1, recv, torch.empty((tok.numel(), d), dtype=torch.float16, device="cuda"))


def test_chunked_gather_whose_first_chunk_brings_no_records():
    """A pipelined `gather_rows` whose first chunk references head rows only: its records range is empty, so the NEXT chunk's
    records start at record 0 as well -- the row map is (re)started although the first chunk's lists were already rewritten
    (against an empty map: nothing in them depends on it).  Found by the split-phase soak; must not be refused as 'a new
    exchange on rewritten lists', and the output is the unsharded table's."""
    from scone_amd.hip_backend import SconeError, SconeTable
    rng = np.random.default_rng(3)
    vocab, n, d, head = 12, 200, 768, 12
    keys = np.zeros((n, 3), dtype=np.uint32)
    lens = np.ones(n, dtype=np.uint8)
    keys[:vocab, 0] = np.arange(vocab)                                   # ids 0..11: the unigrams = the replicated head
    pairs = [(a, b) for a in range(6, vocab) for b in range(6, vocab)] + [(a, b, c) for a in range(6, vocab) for b in range(6, vocab) for c in range(6, vocab)]
    for i, g in enumerate(pairs[:n - vocab]):                            # bigrams / trigrams over tokens 6..11 only
        keys[vocab + i, :len(g)] = g
        lens[vocab + i] = len(g)
    table = rng.standard_normal((n, d)).astype(np.float32)
    full = SconeTable(3, n, d, "int8")
    full.index_build(keys, lens)
    full.store_f32(torch.from_numpy(table))
    shards = []
    for r in range(2):
        s = SconeTable(3, n, d, "int8", row_begin=r * n // 2, row_end=(r + 1) * n // 2)
        s.index_build(keys, lens)
        s.store_f32(torch.from_numpy(table[r * n // 2:(r + 1) * n // 2]), row0=r * n // 2)
        s.shard_set_head(head)
        s.shard_head_store_f32(torch.from_numpy(table[:head]), row0=0)
        shards.append(s)
    B, T = 4, 24
    tok_np = np.empty((B, T), dtype=np.int64)
    tok_np[:2] = rng.integers(0, 6, size=(2, T))                         # chunk 0: tokens 0..5 -> unigram (head) rows only
    tok_np[2:] = rng.integers(6, vocab, size=(2, T))                     # chunk 1: bigrams / trigrams of both shards
    tok = torch.from_numpy(tok_np)
    want = full.embed(tok, out_dtype=torch.float32).reshape(B * T, d)
    ends = [s.shard_gather_plan_chunks(tok, 2) for s in shards]
    assert all(e[0] == 0 for e in ends) and ends[0][1] > 0
    rec = shards[0].shard_record_bytes()
    m = max(e[1] for e in ends)
    records = torch.empty((2 * m, rec), dtype=torch.uint8, device="cuda")
    for r, s in enumerate(shards):
        s.shard_gather_pack_range(0, ends[r][1], records[r * m:(r + 1) * m])
    for s in shards:
        out = torch.empty((B * T, d), dtype=torch.float32, device="cuda")
        s.shard_gather_add_records(records, 0, 0)                        # chunk 0: nothing arrived
        s.shard_gather_embed_range(tok, 0, 2, records, out)
        s.shard_gather_add_records(records, 0, 2 * m)                    # chunk 1's records start at record 0 too
        s.shard_gather_embed_range(tok, 2, 4, records, out)
        assert torch.equal(out, want) and s.status() == 0
        with pytest.raises(SconeError):                                  # but NOW the lists depend on the map: a restart is refused
            s.shard_gather_add_records(records, 0, 2 * m)


def test_sharded_cache_world1_row_exchange():
    from scone_amd import EmbeddingCache
    from scone_amd.distributed import ShardedEmbeddingCache
    rng = np.random.default_rng(8)
    vocab, n, d = 37, 800, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, 3)
    plain = EmbeddingCache.from_synthetic(ex, d, table_format="int8", seed=3)
    sharded = ShardedEmbeddingCache.from_synthetic(ex, d, table_format="int8", seed=3, rank=0, world=1)
    tok = torch.from_numpy(rng.integers(0, vocab, size=(3, 41)))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((41, d)).astype(np.float32)).half().cuda()
    a = plain.embed_tokens(tok, wte=wte, wpe=wpe)
    assert torch.equal(a, sharded.embed_tokens(tok, wte=wte, wpe=wpe, exchange="rows"))      # one shard: plain lookup
    assert torch.equal(a, sharded._embed_row_exchange_dedup(tok, "mean", wte, wpe, None, torch.float16, True))   # plan / pack / embed with itself
    assert torch.equal(a, sharded._embed_gather_rows(sharded.table._tok(tok), "mean", wte, wpe, None, torch.float16).view_as(a))
    assert torch.equal(a, sharded.embed_tokens(tok, wte=wte, wpe=wpe, exchange="partial_sums"))


def test_integration_stub_from_the_docs():
    """The ctypes binding shown in INTEGRATION.md section 2 (independent of scone_amd._lib) drives the library."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "scone_amd", "csrc", "libscone_hip.so"))

    class _Cfg(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_n", C.c_int32),
                    ("dim", C.c_int32), ("table_fmt", C.c_int32), ("placement", C.c_int32),
                    ("n_rows", C.c_uint64), ("row_begin", C.c_uint64), ("row_end", C.c_uint64),
                    ("index_capacity", C.c_uint64), ("hot_rows", C.c_uint64),
                    ("lookup_mode", C.c_uint32), ("stage_tokens", C.c_uint32), ("cache_rows", C.c_uint64)]

    lib.scone_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]
    lib.scone_index_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    lib.scone_table_store_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.scone_embed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.scone_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(0)
    f_gram_to_id = {(3,): 0, (3, 4): 1, (4,): 2, (3, 4, 5): 3}
    n, d, max_n = 4, 768, 3
    cfg = _Cfg(C.sizeof(_Cfg), torch.cuda.current_device(), max_n, d, 2, 0, n, 0, 0, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert lib.scone_create(C.byref(cfg), C.byref(h)) == 0
    keys = torch.zeros(n, max_n, dtype=torch.int32)
    lens = torch.zeros(n, dtype=torch.uint8)
    for g, i in f_gram_to_id.items():
        keys[i, :len(g)] = torch.tensor(g, dtype=torch.int32)
        lens[i] = len(g)
    assert lib.scone_index_build(h, keys.data_ptr(), lens.data_ptr(), n, 0) == 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    rows = torch.from_numpy(table).cuda()
    assert lib.scone_table_store_f32(h, rows.data_ptr(), 0, n, None) == 0
    torch.cuda.synchronize()
    wte = torch.randn(8, d, device="cuda").half()
    wpe = torch.randn(8, d, device="cuda").half()
    tok = torch.tensor([[3, 4, 5, 6]], dtype=torch.int32, device="cuda")
    out = torch.empty(1, 4, d, dtype=torch.float16, device="cuda")
    rc = lib.scone_embed(h, tok.data_ptr(), 1, 4, wte.data_ptr(), 8, wpe.data_ptr(), 8, None, 0, out.data_ptr(), 1,
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    keys_np, lens_np = keys.numpy().astype(np.uint32), lens.numpy()
    ro, ri = R.hits_to_csr(R.match_hits(keys_np, lens_np, tok.cpu().numpy(), max_n))
    deq = R.dequantize_i8(*R.quantize_i8(table))
    fg = torch.from_numpy(R.embed_numpy(deq, ro, ri, "mean").reshape(1, 4, d))
    ref = R.combine(tok.cpu().long(), fg, wte.float().cpu(), wpe.float().cpu()).numpy()
    assert _rel(out.float().cpu().numpy(), ref) < REL_TOL
    lib.scone_destroy(h)


def test_sync_free_plan_and_pack_do_not_block_the_host_and_pack_the_same_rows():
    """`scone_shard_gather_plan_async` + `scone_shard_cols_pack_cap` (round 4): the plan's count never comes back to the host.
    With 20 ms of lookups queued in front, both calls return while that work is still running (an event recorded behind it
    has not completed) -- the synchronous plan returns only after it --, and what they pack is what the synchronous plan +
    `scone_shard_cols_pack` pack: the same count in the header, the same rows under the same ids; a capacity below the count
    packs `cap` rows and raises the overflow flag."""
    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    keys, lens = S.make_keys(300_000, S.GPT2_VOCAB, 3, seed=11)
    ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
    cache = EmbeddingCache.from_synthetic(ex, 768, table_format="int8", seed=3)
    t = cache.table
    B, T = 1024, 512
    tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 5)).to("cuda", torch.int32)
    out = torch.empty(B, T, 768, dtype=torch.float16, device="cuda")
    n = t.shard_gather_plan_chunks(tok, 1)[0]                             # synchronous reference
    slots = t.cols_frag_slots(n)
    rows0 = torch.empty((n, t.payload_bytes()), dtype=torch.uint8, device="cuda")
    sc0 = torch.empty((n, t.scale_bytes()), dtype=torch.uint8, device="cuda")
    fr0 = torch.empty(slots, dtype=torch.int64, device="cuda")
    t.shard_cols_pack(0, n, rows0, sc0, fr0)
    torch.cuda.synchronize()

    def by_id(rows, sc, fr):
        f = fr.cpu().numpy()
        f = f[f != 0]
        ids, pos = (f >> 32) - 1, f & 0xFFFFFFFF
        o = np.argsort(ids)
        return ids[o], rows.cpu().numpy()[pos[o]], sc.cpu().numpy()[pos[o]]
    ids0, r0, s0 = by_id(rows0, sc0, fr0)
    assert ids0.size == n and n > 100_000
    for cap in (n + n // 8, n // 2):
        cslots = t.cols_frag_slots(cap)
        rows1 = torch.empty((cap, t.payload_bytes()), dtype=torch.uint8, device="cuda")
        sc1 = torch.empty((cap, t.scale_bytes()), dtype=torch.uint8, device="cuda")
        fr1 = torch.empty(cslots + 2, dtype=torch.int64, device="cuda")
        for _ in range(30):                                               # ~20 ms of work in front
            t.embed(tok, out=out, out_dtype=torch.float16)
        busy = torch.cuda.Event()
        busy.record()
        t0 = time.perf_counter()
        t.shard_gather_plan_async(tok)
        t.shard_cols_pack_cap(cap, rows1, sc1, fr1[:cslots], fr1[cslots:])
        host_ms = (time.perf_counter() - t0) * 1e3
        still_running = not busy.query()
        torch.cuda.synchronize()
        assert still_running and host_ms < 12.0, (still_running, host_ms)  # neither call waited for the ~20 ms of device work
        hdr = fr1[cslots:].cpu().tolist()
        assert hdr == [n, int(n > cap)]
        ids1, r1, s1 = by_id(rows1, sc1, fr1[:cslots])
        assert ids1.size == min(n, cap) and np.isin(ids1, ids0).all()
        sel = np.searchsorted(ids0, ids1)
        assert np.array_equal(r1, r0[sel]) and np.array_equal(s1, s0[sel])
        with pytest.raises(Exception):                                    # the count is on the device: the host-count pack is refused
            t.shard_cols_pack(0, 1, rows1, sc1, fr1[:cslots])
    for _ in range(30):
        t.embed(tok, out=out, out_dtype=torch.float16)
    busy = torch.cuda.Event()
    busy.record()
    t.shard_gather_plan_chunks(tok, 1)                                    # the synchronous plan DOES wait
    assert busy.query() and t.status() == 0


def test_row_exchange_missing_records_are_reported_not_read_out_of_bounds():
    """A referenced row whose record never arrived (the ranks disagreed about the batch: a caller error) is redirected to
    record 0 -- or to the zero row when nothing arrived -- and reported through status bit 1; the lookup never reads outside
    the receive buffer."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(2)
    vocab, n, d = 11, 300, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    t = SconeTable(3, n, d, "int8")
    t.index_build(keys, lens)
    t.store_f32(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)))
    tok = torch.from_numpy(rng.integers(0, vocab, size=(4, 16)))
    out = torch.empty((4 * 16, d), dtype=torch.float32, device="cuda")
    m = t.shard_gather_plan_chunks(tok, 1)[0]
    send = torch.empty((m, t.shard_record_bytes()), dtype=torch.uint8, device="cuda")
    t.shard_gather_pack_range(0, m, send)
    assert m > 4 and t.status() == 0
    half = send[: m // 2].contiguous()                                   # half of the records withheld
    t.shard_gather_add_records(half, 0, half.shape[0])
    t.shard_gather_embed_range(tok, 0, 4, half, out)
    assert t.status() & 2 and bool(torch.isfinite(out).all())
    t.shard_gather_plan_chunks(tok, 1)
    none = send[:0].contiguous()                                          # nothing arrived
    t.shard_gather_add_records(none, 0, 0)
    t.shard_gather_embed_range(tok, 0, 4, none, out)
    assert t.status() & 2 and bool(torch.isfinite(out).all())


# ------------------------------------------------------------------ BASELINE.json configs at their named sizes
def test_config_c1_100k_fp32_table_from_fit():
    """configs[0]: 100K f-grams fp32 d=768, vocabulary from fit on a 1M-token Zipf corpus (the reference's own
    CPU-runnable case), 8 x 512 Zipf tokens: ids bit-exact, fp32 mean bit-exact against the oracle."""
    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    rng = np.random.default_rng(1234)
    cdf = S.zipf_cdf(S.GPT2_VOCAB)
    corpus = [S.zipf_tokens(rng, cdf, 1000).tolist() for _ in range(1000)]
    ex = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100_000).fit_gpu(corpus, verbose=False)
    assert len(ex) == 100_000
    keys, lens = ex.key_arrays()
    d = 768
    table = rng.standard_normal((len(ex), d)).astype(np.float32)
    cache = EmbeddingCache(ex, d, table_format="fp32", keep_host_copy=False)
    cache.cache_embeddings(np.arange(len(ex)), torch.from_numpy(table), verbose=False)
    tok = S.zipf_tokens(rng, cdf, (8, 512))
    off, ids = cache.match(torch.from_numpy(tok))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, 3))
    assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(ids.cpu().numpy(), ri)
    assert 2.0 < len(ri) / tok.size < 3.2                      # the survey measured 2.59 hits per token on this setup
    out = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out, R.embed_numpy(table, ro, ri, "mean").reshape(8, 512, d))
    # the reference's per-position API on one sequence
    te = cache.get_token_embeddings(tok[0].tolist())
    assert all(np.array_equal(te[p].numpy(), table[ri[ro[p]:ro[p + 1]]]) for p in te)


def test_config_c3_10m_int8_d1024(lookup_form):
    """configs[2]: 10M f-grams INT8 d=1024 in HBM: index of 1e7 exact keys, spot check against the oracle."""
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    n, d = 10_000_000, 1024
    keys, lens = S.make_keys_structured(n)
    cache = EmbeddingCache.from_synthetic(_extractor(keys, lens, 3), d, table_format="int8", seed=7, base_scale=0.02 / 127)
    assert cache.table.index_stats()[0] == n
    tok = S.stream_uniform_ids(keys, lens, 4, 512, 3)
    off, ids = cache.match(torch.from_numpy(tok))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, 3))
    assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(ids.cpu().numpy(), ri)
    uniq, inv = np.unique(ri, return_inverse=True)
    rows = R.synth_rows_i8(7, uniq, d).astype(np.float32) * R.synth_scale_f16(7, uniq, 0.02 / 127).astype(np.float32)[:, None]
    out = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out, R.embed_numpy(rows, ro, inv, "mean").reshape(4, 512, d))


def test_config_c4_100m_int4_d1024_at_its_named_size(lookup_form):
    """configs[3] at its named size: 100M f-grams INT4 d=1024 (52.8 GB of rows, 4.3 GB index).  Rows at both ends,
    in the middle and around row 2^25 equal the host generator -- a one-wave-per-row fill of this table once
    stopped at row 33.5M because blocks x threads wrapped at 2^32 work-items, silently -- and lookups over the
    whole id range equal the oracle (ids bit-exact, fp32 result bit-exact on recomputed rows)."""
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    n, d = 100_000_000, 1024
    free, _ = torch.cuda.mem_get_info()
    if free < 70e9:
        pytest.skip("needs 70 GB of free HBM")
    keys, lens = S.make_keys_structured(n)
    cache = EmbeddingCache.from_synthetic(_extractor(keys, lens, 3), d, table_format="int4", seed=7, base_scale=0.02 / 127)
    table = cache.table
    assert table.index_stats()[0] == n
    ids = np.array([0, 1, 2**25 - 1, 2**25, 2**25 + 1, n // 2, 2**26 + 12345, n - 2, n - 1], dtype=np.int64)
    got = table.gather_rows(torch.from_numpy(ids)).cpu().numpy()
    assert np.array_equal(got, R.dequantize_i4(*R.synth_rows_i4(7, ids, d, 0.02 / 127)))
    tok = S.stream_uniform_ids(keys, lens, 4, 512, 3)
    off, mids = cache.match(torch.from_numpy(tok))
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, 3))
    assert np.array_equal(off.cpu().numpy(), ro) and np.array_equal(mids.cpu().numpy(), ri)
    assert ri.max() > 0.9 * n                                   # the stream does reach the end of the table
    uniq, inv = np.unique(ri, return_inverse=True)
    rows = R.dequantize_i4(*R.synth_rows_i4(7, uniq, d, 0.02 / 127))
    out = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
    assert np.array_equal(out, R.embed_numpy(rows, ro, inv, "mean").reshape(4, 512, d))


def test_config_c4_pinned_host_at_its_named_size():
    """configs[3] as BASELINE words it: the 100M-row INT4 d=1024 table in PINNED HOST DRAM (52.8 GB mapped into the
    GPU, 1M hot rows in HBM), read in place (zero-copy) and through the staged prefetch: rows at large offsets equal
    the host generator and both lookups equal the oracle on recomputed rows, bit for bit."""
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    n, d = 100_000_000, 1024
    keys, lens = S.make_keys_structured(n)
    ex = _extractor(keys, lens, 3)
    tok = S.stream_uniform_ids(keys, lens, 4, 512, 5)
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, 3))
    uniq, inv = np.unique(ri, return_inverse=True)
    want = R.embed_numpy(R.dequantize_i4(*R.synth_rows_i4(7, uniq, d, 0.02 / 127)), ro, inv, "mean").reshape(4, 512, d)
    ids = np.array([0, 999_999, 1_000_000, 2**25 + 3, n // 2, n - 1], dtype=np.int64)
    for stage_tokens in (0, 1024):
        try:
            cache = EmbeddingCache.from_synthetic(ex, d, table_format="int4", seed=7, base_scale=0.02 / 127,
                                                  placement="pinned_host", hot_rows=1_000_000, stage_tokens=stage_tokens)
        except (MemoryError, RuntimeError) as e:                 # hipHostMalloc refused (memlock limit, small host)
            pytest.skip(f"needs 52.8 GB of pinned host memory: {e}")
        got = cache.table.gather_rows(torch.from_numpy(ids)).cpu().numpy()
        assert np.array_equal(got, R.dequantize_i4(*R.synth_rows_i4(7, ids, d, 0.02 / 127))), stage_tokens
        out = cache.embed_tokens(torch.from_numpy(tok), out_dtype=torch.float32).cpu().numpy()
        assert np.array_equal(out, want), stage_tokens
        del cache
        torch.cuda.empty_cache()


def test_config_c5_shard_rows_at_its_named_size():
    """configs[4]: the last of 8 shards of the 1e9-row INT4 d=1024 table (125M rows = 66 GB on this GPU; the index
    is left empty here): the shard's first, middle and last rows equal the host generator."""
    from scone_amd.hip_backend import SconeTable
    n, d, world = 1_000_000_000, 1024, 8
    free, _ = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs 80 GB of free HBM")
    lo, hi = (world - 1) * n // world, n
    t = SconeTable(3, n, d, "int4", row_begin=lo, row_end=hi, index_capacity=64)
    t.fill_synthetic(7, 0.02 / 127)
    ids = np.array([lo, lo + 1, lo + 2**25, lo + 2**26 + 7, (lo + hi) // 2, hi - 2, hi - 1], dtype=np.int64)
    got = t.gather_rows(torch.from_numpy(ids)).cpu().numpy()
    assert np.array_equal(got, R.dequantize_i4(*R.synth_rows_i4(7, ids, d, 0.02 / 127)))
    z = t.gather_rows(torch.tensor([lo - 1]))                   # not this shard's row: zeros + status bit, no fault
    assert not bool(z.any()) and t.status() & 2


def test_config_c5_last_shard_lookup_at_its_named_size():
    """configs[4]: the LAST of 8 shards of the 1e9-row INT4 d=1024 table with the full replicated index (2^31 slots,
    34 GB; ids up to 1e9 - 1).  A stream laid out from f-grams of this shard must match to exactly those ids at their
    first positions (the index is exact, so any other hit there would be a wrong id), and the shard's partial sums
    equal the rows recomputed on the host."""
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import SconeTable
    n, d, world = 1_000_000_000, 1024, 8
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs 120 GB of free HBM")
    import psutil
    if psutil.virtual_memory().available < 60e9:
        pytest.skip("needs 60 GB of host memory for the 1e9 keys")
    keys, lens = S.make_keys_structured(n)
    lo, hi = (world - 1) * n // world, n
    t = SconeTable(3, n, d, "int4", row_begin=lo, row_end=hi)
    t.index_build(keys, lens)
    assert t.index_stats()[0] == n
    t.fill_synthetic(7, 0.02 / 127)
    rng = np.random.default_rng(8)
    pick = np.concatenate([rng.integers(lo, hi, size=60), [lo, hi - 1, hi - 2]]).astype(np.int64)
    T = 4                                                        # one f-gram per sequence, padded with a token outside the vocabulary
    tok = np.full((len(pick), T), S.GPT2_VOCAB + 5, dtype=np.int64)
    for r, i in enumerate(pick):
        tok[r, :lens[i]] = keys[i, :lens[i]]
    hits = t.match(torch.from_numpy(tok)).cpu().numpy()          # [max_n, B, T]
    for r, i in enumerate(pick):
        assert hits[lens[i] - 1, r, 0] == i, (r, i)
    partial, counts = t.embed_partial(torch.from_numpy(tok))
    partial = partial.cpu().numpy().reshape(len(pick), T, d)
    rows = R.dequantize_i4(*R.synth_rows_i4(7, pick, d, 0.02 / 127))
    for r, i in enumerate(pick):
        # position 0 is covered by f-gram i (owned by this shard) and by f-grams of other shards (unigram, prefixes):
        # the shard's partial sum there is exactly row i
        own = [h for h in (hits[0, r, 0], hits[1, r, 0], hits[2, r, 0]) if lo <= h < hi]
        assert own == [i]
        assert np.array_equal(partial[r, 0], rows[r]), (r, i)


# ------------------------------------------------------------------ callers of the match step
def test_fgram_tokenizer_and_dataset_ids_match_the_reference(golden_dir, tmp_path):
    """FGramTokenizer.tokenize / batch_tokenize (f_gram_tokenizer.py:38-126) and the f-gram id vector of
    SconeDataset.__getitem__ (dataset.py:117-147) against outputs captured from the reference."""
    import sys
    sys.path.insert(0, os.path.dirname(golden_dir))
    from stub_tokenizer import StubTokenizer
    from scone_amd import FGramTokenizer
    z = np.load(os.path.join(golden_dir, "callers.npz"))
    ex = _extractor(z["keys"], z["lens"], int(z["max_n"]))
    ft = FGramTokenizer(StubTokenizer(), ex)
    assert FGramTokenizer(tokenizer=ft.base_tokenizer, n_gram_extractor=ex).n_gram_extractor is ex   # train.py:290-293
    texts = [str(t) for t in z["texts"]]
    f2id = ex.f_gram_to_id

    def csr_of(tfg, n):
        off, ids = [0], []
        for pos in range(n):
            ids.extend(f2id[g] for g in tfg[pos])
            off.append(len(ids))
        return np.asarray(off), np.asarray(ids, dtype=np.int64)

    for i, t in enumerate(texts):
        for tag, kw in (("plain", {}), ("trunc", {"max_length": 8, "truncation": True})):
            r = ft.tokenize(t, **kw)
            assert r["input_ids"] == z[f"tok{i}_{tag}_input_ids"].tolist()
            assert r["attention_mask"] == z[f"tok{i}_{tag}_mask"].tolist()
            off, ids = csr_of(r["token_f_grams"], len(r["input_ids"]))
            assert np.array_equal(off, z[f"tok{i}_{tag}_off"]) and np.array_equal(ids, z[f"tok{i}_{tag}_ids"])
        assert "token_f_grams" not in ft.tokenize(t, return_f_grams=False)
    r = ft.batch_tokenize(texts, max_length=24)
    assert np.array_equal(r["input_ids"].numpy(), z["batch_input_ids"])
    T = r["input_ids"].shape[1]
    flat = []
    for b in range(len(texts)):
        off, ids = csr_of(r["token_f_grams"][b], T)
        assert np.array_equal(off, z["batch_off"][b])
        flat.append(ids)
    assert np.array_equal(np.concatenate(flat), z["batch_ids_flat"])
    off_d, ids_d = ft.batch_f_gram_ids(r["input_ids"])                      # device CSR form, no tuples
    assert np.array_equal(ids_d.cpu().numpy(), z["batch_ids_flat"]) and int(off_d[-1]) == len(z["batch_ids_flat"])
    for max_length in (16, 4):
        for i, t in enumerate(texts):
            ids = ft.base_tokenizer(t, max_length=max_length, truncation=True, return_tensors="pt")["input_ids"].squeeze(0)
            got_ids, got_mask = ft.flat_f_gram_ids(ids.tolist(), max_length=max_length)
            assert np.array_equal(got_ids.numpy(), z[f"ds{max_length}_f_gram_ids"][i])
            assert np.array_equal(got_mask.numpy(), z[f"ds{max_length}_f_gram_mask"][i])
    # extract_f_grams (preprocessing.py:12-50): tokenise + fit on the GPU, same f-grams and ids as the reference
    from scone_amd.data import extract_f_grams
    max_n, min_freq, max_f = (int(x) for x in z["xf_args"])
    xf = extract_f_grams([str(t) for t in z["xf_texts"]], StubTokenizer(), max_n=max_n, min_freq=min_freq,
                         max_f_grams=max_f, verbose=False)
    k2, l2 = xf.key_arrays()
    assert np.array_equal(l2, z["xf_lens"]) and np.array_equal(k2, z["xf_keys"])
    # save_pretrained / from_pretrained round trip (n_gram_extractor.npy in the reference's format)
    ft.save_pretrained(str(tmp_path / "ft"))
    back = FGramTokenizer.from_pretrained(str(tmp_path / "ft"), base_tokenizer=StubTokenizer())
    assert back.n_gram_extractor.f_gram_to_id == f2id
    assert back.tokenize(texts[0])["token_f_grams"] == ft.tokenize(texts[0])["token_f_grams"]


def test_interleaved_handles_streams_and_growing_batches():
    """Two handles on two HIP streams, batch sizes that grow and shrink (workspaces are re-allocated on the way), fused
    small-batch and two-kernel paths interleaved with match_csr and get_token_embeddings: every result equals the
    result of the same call made alone."""
    rng = np.random.default_rng(123)
    vocab, d = 41, 768
    caches, refs = [], []
    for seed, fmt, max_n in ((1, "int8", 3), (2, "fp16", 4)):
        r = np.random.default_rng(seed)
        n = 900
        lens = r.integers(1, max_n + 1, size=n).astype(np.uint8)
        keys = r.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
        keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
        table = r.standard_normal((n, d)).astype(np.float32)
        caches.append(_cache(keys, lens, max_n, table, fmt))
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((64, d)).astype(np.float32)).half().cuda()
    shapes = [(1, 5), (3, 64), (70, 64), (2, 9), (300, 64), (1, 1), (40, 64), (8, 3), (500, 64), (2, 64)]
    toks = [torch.from_numpy(rng.integers(0, vocab, size=s)).to("cuda", torch.int32) for s in shapes]
    alone = [[c.embed_tokens(t, wte=wte, wpe=wpe).clone() for t in toks] for c in caches]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        outs = [[None] * len(toks) for _ in caches]
        for k in rng.permutation(len(toks)):
            for ci, c in enumerate(caches):
                with torch.cuda.stream(streams[ci]):
                    outs[ci][k] = c.embed_tokens(toks[k], wte=wte, wpe=wpe)
                    if k % 3 == 0:
                        c.match(toks[k])                             # uses the same handle's workspaces in between
            if k % 4 == 1:
                caches[0].get_token_embeddings(toks[k][0].tolist())
        torch.cuda.synchronize()
        for ci in range(len(caches)):
            for k in range(len(toks)):
                assert torch.equal(outs[ci][k], alone[ci][k]), (rep, ci, k)


@pytest.mark.parametrize("fmt,d", [("int8", 768), ("int4", 1024), ("fp16", 1280), ("fp32", 768), ("int8", 256)])
def test_gather_reduce_csr_entry_point(fmt, d):
    """scone_gather_reduce: caller-supplied per-token id lists (empty, short, exactly 10, longer than any list the
    reference can produce), optional base rows, mean / sum, a shard that owns only part of the ids, an id outside the
    table (status bit, skipped, still counted in K).  fp32 out bit-exact against the oracle on the dequantised table."""
    from scone_amd.hip_backend import SconeTable
    rng = np.random.default_rng(77)
    n = 3000
    table = rng.standard_normal((n, d)).astype(np.float32)
    keys = np.zeros((n, 3), dtype=np.uint32)
    keys[:, 0] = np.arange(n)
    lens = np.ones(n, dtype=np.uint8)
    ks = [0, 1, 2, 3, 6, 10, 11, 17, 40, 5, 0, 10, 1]
    ks = ks + rng.integers(0, 13, size=500).tolist()
    off = np.zeros(len(ks) + 1, dtype=np.int64)
    np.cumsum(ks, out=off[1:])
    ids = rng.integers(0, n, size=int(off[-1])).astype(np.int64)
    deq = {"fp32": table, "fp16": table.astype(np.float16).astype(np.float32),
           "int8": R.dequantize_i8(*R.quantize_i8(table)), "int4": R.dequantize_i4(*R.quantize_i4(table))}[fmt]
    base = rng.standard_normal((len(ks), d)).astype(np.float32)
    for lo, hi in ((0, n), (1000, 2200)):
        t = SconeTable(3, n, d, fmt, row_begin=lo, row_end=hi)
        t.index_build(keys, lens)
        t.store_f32(torch.from_numpy(table[lo:hi]), row0=lo)
        own = (ids >= lo) & (ids < hi)
        seg = np.repeat(np.arange(len(ks)), ks)
        off_own = np.zeros(len(ks) + 1, dtype=np.int64)
        np.cumsum(np.bincount(seg[own], minlength=len(ks)), out=off_own[1:])
        sums = R.embed_numpy(deq, off_own, ids[own], "sum")
        kf = np.asarray(ks, dtype=np.float32)[:, None]
        mean = np.where(kf > 1, sums / np.maximum(kf, 1), sums).astype(np.float32)       # K counts every listed id
        for reduce, want in (("sum", sums), ("mean", mean)):
            got = t.gather_reduce(torch.from_numpy(off), torch.from_numpy(ids), reduce).cpu().numpy()
            assert np.array_equal(got, want), (fmt, d, lo, reduce)
        got = t.gather_reduce(torch.from_numpy(off), torch.from_numpy(ids), "mean", base=torch.from_numpy(base)).cpu().numpy()
        assert np.array_equal(got, base + mean)
        got16 = t.gather_reduce(torch.from_numpy(off), torch.from_numpy(ids), "mean", base=torch.from_numpy(base),
                                out_dtype=torch.float16).float().cpu().numpy()
        b16 = torch.from_numpy(base).half().float().numpy()
        assert _rel(got16, b16 + mean) < REL_TOL
        assert t.status() == 0
        bad = ids.copy()
        bad[off[5]] = n + 5                                        # token 5 (K = 10): one id outside the table
        got = t.gather_reduce(torch.from_numpy(off), torch.from_numpy(bad), "sum").cpu().numpy()
        assert t.status() & 2
        keep = own.copy()
        keep[off[5]] = False
        off_b = np.zeros(len(ks) + 1, dtype=np.int64)
        np.cumsum(np.bincount(seg[keep], minlength=len(ks)), out=off_b[1:])
        assert np.array_equal(got, R.embed_numpy(deq, off_b, bad[keep], "sum"))


@pytest.mark.parametrize("fmt", ["fp32", "int8", "int4"])
def test_embed_tokens_with_a_dense_base_vs_oracle(fmt):
    """embed_tokens(base=[B, T, d]) (SURVEY 8b's additive API: the caller's own inputs_embeds): base + mean of the rows, fp32
    bit-exact against the oracle on the dequantised table, fp16 within 1e-3; and equal to wte[tok] passed as the base."""
    from scone_amd import EmbeddingCache
    rng = np.random.default_rng(12)
    vocab, n, d, max_n = 40, 900, 1024, 3
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    ex = _extractor(keys, lens, max_n)
    keys, lens = ex.key_arrays()
    table = (rng.standard_normal((len(lens), d)) * 0.3).astype(np.float32)
    cache = EmbeddingCache(ex, d, table_format=fmt)
    cache.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
    deq = {"fp32": table, "int8": R.dequantize_i8(*R.quantize_i8(table)), "int4": R.dequantize_i4(*R.quantize_i4(table))}[fmt]
    B, T = 6, 37
    tok = rng.integers(0, vocab + 2, size=(B, T))
    base = rng.standard_normal((B, T, d)).astype(np.float32)
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok, max_n))
    fg = R.embed_numpy(deq, ro, ri, "mean").reshape(B, T, d)
    got = cache.embed_tokens(torch.from_numpy(tok), base=torch.from_numpy(base).cuda())
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), base + fg)
    got16 = cache.embed_tokens(torch.from_numpy(tok), base=torch.from_numpy(base).cuda().half())
    assert got16.dtype == torch.float16
    assert _rel(got16.float().cpu().numpy(), torch.from_numpy(base).half().float().numpy() + fg) < REL_TOL
    sums = cache.embed_tokens(torch.from_numpy(tok), base=torch.zeros(B, T, d).cuda(), reduce="sum")
    assert np.array_equal(sums.cpu().numpy(), R.embed_numpy(deq, ro, ri, "sum").reshape(B, T, d))
    with pytest.raises(ValueError):
        cache.embed_tokens(torch.from_numpy(tok), base=torch.zeros(B, T, d + 8).cuda())
    with pytest.raises(ValueError):
        cache.embed_tokens(torch.from_numpy(tok), base=torch.zeros(B, T, d).cuda(), wte=torch.zeros(vocab + 2, d).cuda())


def test_one_handle_from_two_host_threads():
    """Batches that take the one-launch kernel touch no handle state: two host threads, each on its own HIP stream,
    hammer ONE handle concurrently; every result equals the single-threaded answer."""
    import threading
    rng = np.random.default_rng(31)
    vocab, n, d = 53, 1500, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = _cache(keys, lens, 3, rng.standard_normal((n, d)).astype(np.float32), "int8")
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((128, d)).astype(np.float32)).half().cuda()
    toks = [torch.from_numpy(rng.integers(0, vocab, size=s)).to("cuda", torch.int32) for s in ((1, 7), (4, 128), (64, 128), (2, 33))]
    want = [cache.embed_tokens(t, wte=wte, wpe=wpe).clone() for t in toks]
    torch.cuda.synchronize()
    errors = []

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(300):
                    k = int(r.integers(len(toks)))
                    out = cache.embed_tokens(toks[k], wte=wte, wpe=wpe)
                    if it % 25 == 0:
                        stream.synchronize()
                        if not torch.equal(out, want[k]):
                            errors.append((seed, it, k))
            stream.synchronize()
        except Exception as e:          # surface in the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("shared_stream,reserve", [(False, 0), (True, 0), (False, 16)], ids=["own_streams", "one_stream", "own_streams_cu_reserve"])
def test_one_handle_large_batches_from_host_threads(shared_stream, reserve):
    """SURVEY 8b "lookups are thread-safe and stream-ordered" for batches ABOVE the one-launch limit: 64k-token batches
    use a workspace (per-token id records) that belongs to the stream of the call and is locked while the match that
    writes it and the lookup that reads it are enqueued.  Three host threads -- each on its own stream, or all on ONE
    stream -- run scone_embed (two batch shapes) and scone_match_csr on one handle; every result equals the
    single-threaded answer.  Round 4, `reserve`: with a CU reserve set the lookups of all three threads hop to the handle's one
    masked stream between two events (scone_lookup_enter / _leave hold the handle's lock across the hop): same results."""
    import threading
    rng = np.random.default_rng(32)
    vocab, n, d = 61, 4000, 768
    lens = rng.integers(1, 4, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, 3)).astype(np.uint32)
    keys[np.arange(3)[None, :] >= lens[:, None]] = 0
    cache = _cache(keys, lens, 3, rng.standard_normal((n, d)).astype(np.float32), "int8")
    wte = torch.from_numpy(rng.standard_normal((vocab, d)).astype(np.float32)).half().cuda()
    wpe = torch.from_numpy(rng.standard_normal((512, d)).astype(np.float32)).half().cuda()
    toks = [torch.from_numpy(rng.integers(0, vocab, size=s)).to("cuda", torch.int32) for s in ((128, 512), (160, 448), (96, 512))]
    assert all(t.numel() > 32768 for t in toks)
    want = [cache.embed_tokens(t, wte=wte, wpe=wpe).clone() for t in toks]
    want_csr = [tuple(x.clone() for x in cache.table.match_csr(t)) for t in toks]
    torch.cuda.synchronize()
    if reserve:
        cache.table.set_cu_reserve(reserve)
        assert cache.table.cu_reserve()[0] == reserve
    errors = []
    one = torch.cuda.Stream()

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            stream = one if shared_stream else torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(60):
                    k = int(r.integers(len(toks)))
                    if it % 7 == 3:
                        off, ids = cache.table.match_csr(toks[k])            # synchronises its stream
                        if not (torch.equal(off, want_csr[k][0]) and torch.equal(ids, want_csr[k][1])):
                            errors.append((seed, it, k, "csr"))
                        continue
                    out = cache.embed_tokens(toks[k], wte=wte, wpe=wpe)
                    if it % 5 == 0:
                        stream.synchronize()
                        if not torch.equal(out, want[k]):
                            errors.append((seed, it, k))
            stream.synchronize()
        except Exception as e:          # surface in the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (1, 2, 3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
