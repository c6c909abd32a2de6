"""Pin oracle/ref_port.py against the fixtures captured from the real reference
(tests/golden/make_golden.py).  CPU only."""

import os

import numpy as np
import pytest
import torch

from oracle import ref_port as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _dict_from(keys, lens):
    return {tuple(int(x) for x in keys[i, :lens[i]]): i for i in range(len(lens))}


def test_match_python_port_bit_exact(golden_dir):
    z = _load(golden_dir, "match.npz")
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        d = _dict_from(keys, lens)
        for si in range(int(z["n_streams"])):
            tok = z[f"{c}_s{si}_tok"].tolist()
            off, ids = R.match_csr_python(d, max_n, tok)
            assert np.array_equal(off, z[f"{c}_s{si}_off"]), (c, si)
            assert np.array_equal(ids, z[f"{c}_s{si}_ids"]), (c, si)


def test_match_numpy_port_bit_exact(golden_dir):
    z = _load(golden_dir, "match.npz")
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        for si in range(int(z["n_streams"])):
            tok = z[f"{c}_s{si}_tok"][None, :]
            hits = R.match_hits(keys, lens, tok, max_n)
            off, ids = R.hits_to_csr(hits)
            assert np.array_equal(off, z[f"{c}_s{si}_off"]), (c, si)
            assert np.array_equal(ids, z[f"{c}_s{si}_ids"]), (c, si)


def test_multiplicity_case(golden_dir):
    """[7,7,7,7] vs {(7,),(7,7),(7,7,7)} gives position 1 ids [0,1,1,2,2]."""
    d = {(7,): 0, (7, 7): 1, (7, 7, 7): 2}
    off, ids = R.match_csr_python(d, 3, [7, 7, 7, 7])
    assert ids[off[1]:off[2]].tolist() == [0, 1, 1, 2, 2]
    assert ids[off[0]:off[1]].tolist() == [0, 1, 2]


def test_fit_order_matches_reference(golden_dir):
    z = _load(golden_dir, "match.npz")
    for c in z["cases"]:
        flat, cl = z[f"{c}_corpus_flat"], z[f"{c}_corpus_lens"]
        min_freq, max_f = (int(x) for x in z[f"{c}_fit_args"])
        corpus, p = [], 0
        for n in cl:
            corpus.append(flat[p:p + n].tolist())
            p += n
        grams = R.fit(corpus, int(z[f"{c}_max_n"]), min_freq, max_f)
        keys, lens = z[f"{c}_keys"], z[f"{c}_lens"]
        assert len(grams) == len(lens)
        for i, g in enumerate(grams):
            assert tuple(int(x) for x in keys[i, :lens[i]]) == g


@pytest.mark.parametrize("use_mm", [False, True])
def test_lookup_port_bit_exact(golden_dir, use_mm):
    z = _load(golden_dir, "lookup.npz")
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        table = z[f"{c}_table"]
        n, d = table.shape
        cache = R.RefCache(_dict_from(keys, lens), max_n, d, use_memory_map=use_mm)
        cache.cache_embeddings(list(range(n)), torch.from_numpy(table))
        # a4
        g = cache.get_embeddings(z[f"{c}_gather_ids"].tolist())
        assert np.array_equal(g.numpy(), z[f"{c}_gather_out"])
        # a5
        te = cache.get_token_embeddings(z[f"{c}_tok"].tolist())
        assert sorted(te.keys()) == z[f"{c}_te_positions"].tolist()
        p = 0
        for pos, rows in zip(z[f"{c}_te_positions"], z[f"{c}_te_rows"]):
            assert np.array_equal(te[int(pos)].numpy(), z[f"{c}_te_stacks"][p:p + rows])
            p += rows
        # a6
        agg = R.aggregate(cache, z[f"{c}_tok"].tolist(), d)
        assert np.array_equal(agg.numpy(), z[f"{c}_agg_f32"]), c
        agg16 = R.aggregate(cache, z[f"{c}_tok"].tolist(), d, half=True)
        assert np.array_equal(agg16.numpy().view(np.uint16), z[f"{c}_agg_f16"].view(np.uint16)), c


def test_embed_numpy_matches_reference_aggregate(golden_dir):
    """The vectorised oracle (sequential fp32 sum in list order, / K) against engine.py:250."""
    z = _load(golden_dir, "lookup.npz")
    for c in z["cases"]:
        keys, lens, max_n = z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"])
        hits = R.match_hits(keys, lens, z[f"{c}_tok"][None, :], max_n)
        off, ids = R.hits_to_csr(hits)
        assert np.array_equal(off, z[f"{c}_off"]) and np.array_equal(ids, z[f"{c}_ids"])
        out = R.embed_numpy(z[f"{c}_table"], off, ids, "mean")
        ref = z[f"{c}_agg_f32"][0]
        # torch.mean may divide or multiply by 1/K; allow 1 ulp, report exactness
        np.testing.assert_allclose(out, ref, rtol=3e-7, atol=1e-7)


def test_combine_port_matches_reference(golden_dir):
    z = _load(golden_dir, "combine.npz")
    for c in z["cases"]:
        pos = z[f"{c}_pos"]
        x = R.combine(torch.from_numpy(z[f"{c}_input_ids"]), torch.from_numpy(z[f"{c}_fg"]),
                      torch.from_numpy(z[f"{c}_wte"]), torch.from_numpy(z[f"{c}_wpe"]),
                      torch.from_numpy(z[f"{c}_proj"]),
                      torch.from_numpy(pos) if pos.size else None)
        assert np.array_equal(x.numpy(), z[f"{c}_embeds"]), c


def test_quantizers_round_trip():
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((37, 256)).astype(np.float32)
    rows[3] = 0.0
    q, s = R.quantize_i8(rows)
    dq = R.dequantize_i8(q, s)
    assert np.all(q[3] == 0) and np.all(np.abs(q) <= 127)
    assert np.max(np.abs(dq - rows)) <= 0.51 * float(s.astype(np.float32).max()) + 1e-6
    p, s4 = R.quantize_i4(rows)
    dq4 = R.dequantize_i4(p, s4)
    assert p.shape == (37, 128) and s4.shape == (37, 2)
    assert np.max(np.abs(dq4 - rows)) <= 0.51 * float(s4.astype(np.float32).max()) + 1e-6
    # idempotence: re-quantising a dequantised table reproduces the codes
    q2, s2 = R.quantize_i8(dq)
    assert np.array_equal(R.dequantize_i8(q2, s2), dq)


def test_synth_generators_are_deterministic():
    a = R.synth_rows_i8(7, np.array([0, 1, 2**33 + 5], dtype=np.uint64), 64)
    b = R.synth_rows_i8(7, np.array([0, 1, 2**33 + 5], dtype=np.uint64), 64)
    assert a.dtype == np.int8 and a.shape == (3, 64) and np.array_equal(a, b)
    assert not np.array_equal(a[0], a[1])
    s = R.synth_scale_f16(7, np.arange(100), 0.02 / 127)
    assert s.dtype == np.float16 and np.all(s > 0)


def test_paper_lookup_known_answers():
    """Algorithm 2 (assets/algorithm.png): longest f-gram of length >= 2 ending at the token; unigrams never match."""
    d = {(5,): 0, (5, 6): 1, (4, 5, 6): 2, (6, 7): 3, (9, 9): 4, (9, 9, 9): 5}
    assert R.paper_lookup(d, 3, [4, 5, 6, 7]) == [-1, -1, 2, 3]
    assert R.paper_lookup(d, 2, [4, 5, 6, 7]) == [-1, -1, 1, 3]
    assert R.paper_lookup(d, 3, [9, 9, 9, 9]) == [-1, 4, 5, 5]
    assert R.paper_lookup(d, 3, [5]) == [-1] and R.paper_lookup(d, 3, []) == []
    table = np.arange(6 * 4, dtype=np.float32).reshape(6, 4)
    wte = 100 + np.arange(10 * 4, dtype=np.float32).reshape(10, 4)
    e = R.paper_embed(d, 3, np.array([[4, 5, 6, 7]]), table, wte=wte)
    assert np.array_equal(e[0, 0], wte[4]) and np.array_equal(e[0, 2], table[2]) and np.array_equal(e[0, 3], table[3])


def test_c_oracle_bit_exact_vs_golden(golden_dir):
    """oracle/oracle.c (second, independent restatement): ids and fp32 means equal the reference's."""
    from oracle.c_oracle import COracle
    z = _load(golden_dir, "match.npz")
    for c in z["cases"]:
        co = COracle(z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"]))
        for si in range(int(z["n_streams"])):
            off, ids = co.match_csr(z[f"{c}_s{si}_tok"])
            assert np.array_equal(off, z[f"{c}_s{si}_off"]) and np.array_equal(ids, z[f"{c}_s{si}_ids"]), (c, si)
    z = _load(golden_dir, "lookup.npz")
    for c in z["cases"]:
        co = COracle(z[f"{c}_keys"], z[f"{c}_lens"], int(z[f"{c}_max_n"]))
        out, total = co.embed(z[f"{c}_table"], z[f"{c}_tok"], "mean", nthreads=2)
        assert total == len(z[f"{c}_ids"])
        assert np.array_equal(out, z[f"{c}_agg_f32"]), c


def test_c_oracle_under_asan_ubsan_vs_golden(golden_dir, tmp_path):
    """SURVEY section 5 "race detection / sanitizers" (CPU only): oracle/oracle.c built with
    -fsanitize=address,undefined (oracle/Makefile target `asan`) runs every golden match / lookup case plus the edge
    shapes (T = 0, T < max_n, all-miss, ids beyond 32 bits) through index build, match and the OpenMP batch mean; the
    sanitizers abort on any finding, leaks included, and the outputs still equal the reference's."""
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], check=True, capture_output=True)
    cases, expect = [], []
    zm, zl = _load(golden_dir, "match.npz"), _load(golden_dir, "lookup.npz")
    rng = np.random.default_rng(0)
    for c in zm["cases"]:
        keys, lens, max_n = zm[f"{c}_keys"], zm[f"{c}_lens"], int(zm[f"{c}_max_n"])
        for si in range(int(zm["n_streams"])):
            tok = zm[f"{c}_s{si}_tok"].astype(np.int64).reshape(1, -1)
            table = rng.standard_normal((len(lens), 4)).astype(np.float32)
            cases.append((keys, lens, max_n, table, tok))
            expect.append((zm[f"{c}_s{si}_off"], zm[f"{c}_s{si}_ids"], None))
    for c in zl["cases"]:
        tok = zl[f"{c}_tok"].astype(np.int64).reshape(1, -1)
        cases.append((zl[f"{c}_keys"], zl[f"{c}_lens"], int(zl[f"{c}_max_n"]), zl[f"{c}_table"], tok))
        expect.append((zl[f"{c}_off"], zl[f"{c}_ids"], zl[f"{c}_agg_f32"]))
    k = np.array([[5, 0, 0], [5, 6, 0], [5, 6, 7]], dtype=np.uint32)
    l = np.array([1, 2, 3], dtype=np.uint8)
    t4 = np.eye(3, 4, dtype=np.float32)
    for tok in (np.zeros((2, 0), np.int64), np.array([[5, 6]]), np.array([[9, 9, 9, 9]]), np.array([[5, 2**33, -1, 5]])):
        cases.append((k, l, 3, t4, tok.astype(np.int64)))
        expect.append(None)
    src, dst = str(tmp_path / "cases.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as f:
        f.write(struct.pack("<q", len(cases)))
        for keys, lens, max_n, table, tok in cases:
            f.write(struct.pack("<5q", len(lens), max_n, table.shape[1], tok.shape[0], tok.shape[1]))
            f.write(np.ascontiguousarray(keys, dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(lens, dtype=np.uint8).tobytes())
            f.write(np.ascontiguousarray(table, dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(tok, dtype=np.int64).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:exitcode=24")
    p = subprocess.run([os.path.join(root, "oracle", "_build", "oracle_asan"), src, dst], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    buf = open(dst, "rb").read()
    pos = 0
    for (keys, lens, max_n, table, tok), exp in zip(cases, expect):
        B, T, d = tok.shape[0], tok.shape[1], table.shape[1]
        total = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        off = np.frombuffer(buf, dtype=np.int64, count=B * (T + 1), offset=pos)
        pos += 8 * B * (T + 1)
        ids = np.frombuffer(buf, dtype=np.int64, count=total, offset=pos)
        pos += 8 * total
        mean = np.frombuffer(buf, dtype=np.float32, count=B * T * d, offset=pos).reshape(B, T, d)
        pos += 4 * B * T * d
        if exp is None:
            ro, ri = R.hits_to_csr(R.match_hits(keys, lens, np.where((tok < 0) | (tok >= 2**32 - 1), 2**31 - 1, tok), max_n)) \
                if T else (np.zeros(1, np.int64), np.zeros(0, np.int64))
            assert total == len(ri)
            continue
        assert np.array_equal(off, exp[0]) and np.array_equal(ids, exp[1])
        if exp[2] is not None:
            assert np.array_equal(mean, exp[2])
    assert pos == len(buf)


def test_callers_fixture_is_reproduced_by_the_oracle(golden_dir):
    """callers.npz (FGramTokenizer.tokenize / batch_tokenize and SconeDataset's id vector, captured from the
    reference with tests/stub_tokenizer.py): the stub still produces the recorded token ids and the oracle's
    match reproduces every recorded id list."""
    import sys
    sys.path.insert(0, os.path.dirname(golden_dir))
    from stub_tokenizer import StubTokenizer
    z = _load(golden_dir, "callers.npz")
    keys, lens, max_n = z["keys"], z["lens"], int(z["max_n"])
    f2id = R._key_dict(keys, lens)
    tok = StubTokenizer()
    texts = [str(t) for t in z["texts"]]
    for i, t in enumerate(texts):
        for tag, kw in (("plain", {}), ("trunc", {"max_length": 8, "truncation": True})):
            ids = tok(t, return_tensors="pt", **kw)["input_ids"].squeeze(0).tolist()
            assert ids == z[f"tok{i}_{tag}_input_ids"].tolist()
            off, flat = R.match_csr_python(f2id, max_n, ids)
            assert np.array_equal(off, z[f"tok{i}_{tag}_off"]) and np.array_equal(flat, z[f"tok{i}_{tag}_ids"])
    enc = tok(texts, max_length=24, padding=True, truncation=True, return_tensors="pt")
    assert np.array_equal(enc["input_ids"].numpy(), z["batch_input_ids"])
    assert np.array_equal(enc["attention_mask"].numpy(), z["batch_mask"])
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, z["batch_input_ids"], max_n))
    T = z["batch_input_ids"].shape[1]
    assert np.array_equal(ri, z["batch_ids_flat"])
    for b in range(len(texts)):
        assert np.array_equal(ro[b * T:(b + 1) * T + 1] - ro[b * T], z["batch_off"][b])
    for max_length in (16, 4):
        for i, t in enumerate(texts):
            ids = tok(t, max_length=max_length, truncation=True, return_tensors="pt")["input_ids"].squeeze(0).tolist()
            _, flat = R.match_csr_python(f2id, max_n, ids)
            want = np.zeros(10, dtype=np.int64)
            want[:min(10, len(flat))] = flat[:10]
            assert np.array_equal(want, z[f"ds{max_length}_f_gram_ids"][i])
            assert int(z[f"ds{max_length}_f_gram_mask"][i].sum()) == min(10, len(flat))


def test_extract_f_grams_fixture_is_reproduced_by_the_oracle(golden_dir):
    """extract_f_grams (preprocessing.py:12-50) captured from the reference with the stub tokenizer: the oracle's fit
    on the stub's token ids gives the same f-grams in the same id order, and so does the product's host fit."""
    import sys
    sys.path.insert(0, os.path.dirname(golden_dir))
    from stub_tokenizer import StubTokenizer
    from scone_amd.data import extract_f_grams
    z = _load(golden_dir, "callers.npz")
    max_n, min_freq, max_f = (int(x) for x in z["xf_args"])
    tok = StubTokenizer()
    texts = [str(t) for t in z["xf_texts"]]
    grams = R.fit([tok(t, add_special_tokens=False)["input_ids"] for t in texts], max_n, min_freq, max_f)
    want = [tuple(int(x) for x in z["xf_keys"][i, :z["xf_lens"][i]]) for i in range(len(z["xf_lens"]))]
    assert grams == want
    ex = extract_f_grams(texts, tok, max_n=max_n, min_freq=min_freq, max_f_grams=max_f, verbose=False, use_gpu=False)
    assert [ex.id_to_f_gram[i] for i in range(len(want))] == want
