"""GPU: the blocks bench.py adds around the headline (round 6) -- the `latency` block (the reference's own benchmark grid on a
table, scone/configs/benchmark_config.json:79-82) and config C1 (the reference's CPU-runnable case through the GPU path) --
run on small inputs, so that a broken block is found here and not in the driver's one bench run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def test_latency_block_covers_the_reference_grid_and_both_forms():
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    d = 768
    vocab_obj, keys, lens = bench.make_vocabulary(200_000, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    lat = bench.latency_block(cache, wte, wpe, d, extra_shapes=((80, 512),), calls=30)
    assert list(lat) == ["1x512", "1x1024", "4x512", "4x1024", "8x512", "8x1024", "80x512"]
    for shape, e in lat.items():
        assert e["call_us"] > 0 and e["sync_us"] >= e["call_us"] * 0.5 and e["kernel_us"] > 0
        if shape == "80x512":
            assert e["form"] == "two" and "graph_us" not in e          # 40,960 tokens: match + gather, not launch-bound
        else:
            assert e["form"] == "fused" and e.get("graph_us", 0) > 0, e  # captured in a hipGraph and replayed
    # the table still answers correctly after the captures (a failed capture would leave a sticky HIP error behind)
    tok = torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, 2, 64, 1)).to("cuda", torch.int32)
    a = cache.embed_tokens(tok, wte=wte, wpe=wpe).clone()
    assert torch.equal(a, cache.embed_tokens(tok, wte=wte, wpe=wpe)) and cache.table.status() == 0
    line = bench.compact_record({"metric": "m", "value": 1.0, "latency": lat, "roofline": {}}, None)
    assert set(line["latency"]["1x512"]) >= {"call_us", "sync_us", "kernel_us", "graph_us", "form"}


def test_config_c1_record_is_bit_exact_against_the_oracle():
    import bench
    rec = bench.c1_record(bench.c1_oracle_leg, steps=10, seconds=0.5)
    assert rec["gpu_vs_oracle_bit_exact"] is True and rec["gpu_vs_oracle_max_rel_err"] == 0.0
    assert rec["gpu_vs_oracle_tokens"] == 8 * 512 and 2.0 < rec["mean_hits_per_token"] < 3.2
    assert rec["tokens_per_s"] > 1e6 and rec["cpu_port_1core_tokens_per_s"] > 1e3 and rec["status_bits"] == 0
    assert rec["positions_returned"] > 400 and rec["get_token_embeddings_ms_per_512_token_sequence"] > 0


def test_alloc_output_keeps_the_fastest_candidate_and_changes_no_result():
    """EmbeddingCache.alloc_output (round 6: the lookup kernel's time follows the physical placement of the buffer it writes,
    profiles/r06m): N candidate allocations, timed lookups into each, the fastest kept; the lookup into the chosen buffer is
    the lookup."""
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    d = 768
    vocab_obj, keys, lens = bench.make_vocabulary(200_000, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    tok = torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, 96, 512, 3)).to("cuda", torch.int32)
    out, rep = cache.alloc_output(tok, wte=wte, wpe=wpe, candidates=3, trials=2)
    assert out.shape == (96, 512, d) and out.dtype == torch.float16 and out.is_contiguous()
    assert rep["candidates"] == 3 and len(rep["kernel_ms"]) == 3 and all(t > 0 for t in rep["kernel_ms"])
    assert rep["kept"] in rep["finalists_kernel_ms"] and len(rep["finalists_kernel_ms"]) == 3
    assert rep["kept"] == min(rep["finalists_kernel_ms"], key=rep["finalists_kernel_ms"].get)
    want = cache.embed_tokens(tok, wte=wte, wpe=wpe).clone()
    assert torch.equal(cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out), want)
    plain, rep1 = cache.alloc_output(tok, wte=wte, wpe=wpe, candidates=1)
    assert rep1 == {"candidates": 1, "kernel_ms": [None], "kept": 0} and plain.shape == out.shape
    f32, _ = cache.alloc_output(tok[:2], candidates=2, trials=1)                 # no wte / wpe: fp32 out
    assert f32.dtype == torch.float32 and cache.table.status() == 0
