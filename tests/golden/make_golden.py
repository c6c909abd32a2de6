#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (``/root/reference`` does not exist on the GPU
box and never travels):

    python tests/golden/make_golden.py

It imports the reference's hot-path modules unmodified (one import shim for the
missing ``scone.utils.cloud`` module), drives them on seeded
inputs and stores inputs + the reference's outputs as ``.npz`` data files.  The
fixtures hold data only -- no reference source text.

Reference functions exercised (paths relative to /root/reference):
  * NGramExtractor.fit / get_token_f_grams   scone/tokenization/n_gram_extractor.py:72-126
  * EmbeddingCache.cache_embeddings / get_embeddings / get_token_embeddings / save / load
                                              scone/inference/embedding_cache.py:56-243
  * the aggregate lines of SconeInferenceEngine.generate, re-executed here verbatim
    (they are inline in a method that needs an HF model)   scone/inference/engine.py:247-266
  * SconeLanguageModel.forward (unbound, on a stub holding a locally built GPT-2)
                                              scone/models/language_model.py:181-289
"""

import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("SCONE_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.path.insert(0, REF)
    shim = types.ModuleType("scone.utils.cloud")

    class CloudStorage:  # placeholder for the module the reference forgot to ship
        pass

    shim.CloudStorage = CloudStorage
    sys.modules["scone.utils.cloud"] = shim
    from scone.tokenization.n_gram_extractor import NGramExtractor
    from scone.inference.embedding_cache import EmbeddingCache
    from scone.models.language_model import SconeLanguageModel
    return NGramExtractor, EmbeddingCache, SconeLanguageModel


def zipf_tokens(rng, vocab, size, s=1.1):
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = ranks ** (-s)
    p /= p.sum()
    return rng.choice(vocab, size=size, p=p).astype(np.int64)


def keys_arrays(f_gram_to_id, max_n):
    n = len(f_gram_to_id)
    keys = np.zeros((n, max(max_n, 1)), dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint8)
    for g, i in f_gram_to_id.items():
        keys[i, :len(g)] = g
        lens[i] = len(g)
    return keys, lens


def csr_from_reference(ex, token_ids):
    tfg = ex.get_token_f_grams(list(token_ids))
    offsets = [0]
    ids = []
    for pos in range(len(token_ids)):
        ids.extend(ex.f_gram_to_id[g] for g in tfg[pos])
        offsets.append(len(ids))
    return np.asarray(offsets, dtype=np.int64), np.asarray(ids, dtype=np.int64)


def make_match(NGramExtractor):
    out = {}
    case = 0
    cases = []
    for seed, vocab, max_f in [(11, 40, 300), (12, 50257, 4000), (13, 7, 60)]:
        for max_n in (1, 2, 3, 4):
            rng = np.random.default_rng(seed * 100 + max_n)
            corpus = [zipf_tokens(rng, vocab, int(rng.integers(5, 400))).tolist() for _ in range(40)]
            ex = NGramExtractor(max_n=max_n, min_freq=2, max_f_grams=max_f)
            ex.fit(corpus, verbose=False)
            keys, lens = keys_arrays(ex.f_gram_to_id, max_n)
            name = f"c{case}"
            out[f"{name}_keys"] = keys
            out[f"{name}_lens"] = lens
            out[f"{name}_max_n"] = np.int64(max_n)
            # fit pin: corpus (ragged -> flat + lengths) and the id-ordered f-gram list is keys/lens
            out[f"{name}_corpus_flat"] = np.concatenate([np.asarray(c, dtype=np.int64) for c in corpus])
            out[f"{name}_corpus_lens"] = np.asarray([len(c) for c in corpus], dtype=np.int64)
            out[f"{name}_fit_args"] = np.asarray([2, max_f], dtype=np.int64)
            for si, T in enumerate((0, 1, 2, 3, 17, 512)):
                toks = zipf_tokens(rng, vocab, T)
                off, ids = csr_from_reference(ex, toks.tolist())
                out[f"{name}_s{si}_tok"] = toks
                out[f"{name}_s{si}_off"] = off
                out[f"{name}_s{si}_ids"] = ids
            cases.append(name)
            case += 1
    # the multiplicity case of SURVEY section 0: [7,7,7,7] with {(7,),(7,7),(7,7,7)}
    ex = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100)
    ex.fit([[7, 7, 7, 7]], verbose=False)
    keys, lens = keys_arrays(ex.f_gram_to_id, 3)
    name = f"c{case}"
    out[f"{name}_keys"], out[f"{name}_lens"], out[f"{name}_max_n"] = keys, lens, np.int64(3)
    out[f"{name}_corpus_flat"] = np.asarray([7, 7, 7, 7], dtype=np.int64)
    out[f"{name}_corpus_lens"] = np.asarray([4], dtype=np.int64)
    out[f"{name}_fit_args"] = np.asarray([1, 100], dtype=np.int64)
    for si, toks in enumerate(([7, 7, 7, 7], [7, 7], [7], [], [7, 8, 7, 7, 7, 9], [8, 9])):
        off, ids = csr_from_reference(ex, toks)
        out[f"{name}_s{si}_tok"] = np.asarray(toks, dtype=np.int64)
        out[f"{name}_s{si}_off"], out[f"{name}_s{si}_ids"] = off, ids
    cases.append(name)
    out["cases"] = np.asarray(cases)
    out["n_streams"] = np.int64(6)
    np.savez_compressed(os.path.join(HERE, "match.npz"), **out)
    print("match.npz:", len(cases), "cases")


def make_lookup(NGramExtractor, EmbeddingCache):
    """gather (a4), get_token_embeddings (a5), aggregate (a6), save/load (a9)."""
    out = {}
    cases = []
    for ci, (seed, vocab, max_n, d, max_f, T) in enumerate([
            (21, 30, 3, 64, 1024, 64),
            (22, 30, 4, 64, 700, 33),
            (23, 200, 3, 768, 96, 17),
            (24, 200, 3, 1024, 64, 17),
            (25, 12, 2, 16, 40, 5),
    ]):
        rng = np.random.default_rng(seed)
        corpus = [zipf_tokens(rng, vocab, int(rng.integers(20, 300))).tolist() for _ in range(30)]
        ex = NGramExtractor(max_n=max_n, min_freq=1, max_f_grams=max_f)
        ex.fit(corpus, verbose=False)
        n = len(ex.f_grams)
        torch.manual_seed(seed)
        table = torch.randn(n, d, dtype=torch.float32)
        keys, lens = keys_arrays(ex.f_gram_to_id, max_n)
        toks = zipf_tokens(rng, vocab + 3, T)      # a few out-of-vocabulary tokens -> K = 0 positions
        name = f"c{ci}"
        with tempfile.TemporaryDirectory() as tmp:
            mem = EmbeddingCache(ex, d)
            mem.cache_embeddings(list(range(n)), table, verbose=False)
            mm = EmbeddingCache(ex, d, cache_dir=tmp, use_memory_map=True)
            mm.cache_embeddings(list(range(n)), table, verbose=False)
            # a4: gather
            idlist = rng.integers(0, n, size=23).tolist()
            g_mem = mem.get_embeddings(idlist)
            g_mm = mm.get_embeddings(idlist)
            assert torch.equal(g_mem, g_mm)
            # a5: per-position stacks; K = 0 positions omitted
            te = mem.get_token_embeddings(toks.tolist())
            te_mm = mm.get_token_embeddings(toks.tolist())
            assert sorted(te) == sorted(te_mm) and all(torch.equal(te[p], te_mm[p]) for p in te)
            positions = np.asarray(sorted(te.keys()), dtype=np.int64)
            stacks = torch.cat([te[int(p)] for p in positions], dim=0) if len(positions) else torch.zeros(0, d)
            stack_rows = np.asarray([te[int(p)].shape[0] for p in positions], dtype=np.int64)
            # a6: the engine's aggregate lines, re-executed verbatim (engine.py:234-266)
            token_f_grams = ex.get_token_f_grams(toks.tolist())
            token_embeddings = {}
            for pos, f_grams in token_f_grams.items():
                if not f_grams:
                    continue
                f_gram_ids = [ex.f_gram_to_id[f_gram] for f_gram in f_grams]
                embeddings = mem.get_embeddings(f_gram_ids, torch.device("cpu"))
                token_embeddings[pos] = embeddings.mean(dim=0)
            f_gram_embeddings = torch.zeros((1, len(toks), d), device="cpu")
            for pos, embedding in token_embeddings.items():
                f_gram_embeddings[0, pos] = embedding
            agg_half = f_gram_embeddings.half()
            # a9: in-memory save/load round trip (np.save appends .npy)
            p = os.path.join(tmp, "cache")
            mem.save(p)
            re = EmbeddingCache.load(p + ".npy", ex)
            assert torch.equal(re.get_embeddings(idlist), g_mem)
            if ci == 4:
                # keep one tiny saved cache + extractor as on-disk format fixtures (data files)
                ex.save(os.path.join(HERE, "tiny_extractor"))
                mem.save(os.path.join(HERE, "tiny_cache"))
        off, ids = csr_from_reference(ex, toks.tolist())
        out[f"{name}_keys"], out[f"{name}_lens"] = keys, lens
        out[f"{name}_max_n"] = np.int64(max_n)
        out[f"{name}_table"] = table.numpy()
        out[f"{name}_tok"] = toks
        out[f"{name}_off"], out[f"{name}_ids"] = off, ids
        out[f"{name}_gather_ids"] = np.asarray(idlist, dtype=np.int64)
        out[f"{name}_gather_out"] = g_mem.numpy()
        out[f"{name}_te_positions"] = positions
        out[f"{name}_te_rows"] = stack_rows
        out[f"{name}_te_stacks"] = stacks.numpy()
        out[f"{name}_agg_f32"] = f_gram_embeddings.numpy()
        out[f"{name}_agg_f16"] = agg_half.numpy()
        cases.append(name)
    out["cases"] = np.asarray(cases)
    np.savez_compressed(os.path.join(HERE, "lookup.npz"), **out)
    print("lookup.npz:", len(cases), "cases")


def make_combine(SconeLanguageModel):
    """a7: the tensor handed to transformer(inputs_embeds=...) by the reference forward."""
    from transformers import GPT2Config, GPT2LMHeadModel
    out = {}
    cases = []
    for ci, (seed, n_embd, d_f, B, T, with_pos) in enumerate([
            (31, 16, 8, 2, 9, False), (32, 64, 64, 1, 17, True), (33, 768, 32, 1, 5, False)]):
        torch.manual_seed(seed)
        cfg = GPT2Config(vocab_size=101, n_positions=32, n_embd=n_embd, n_layer=1, n_head=2)
        base = GPT2LMHeadModel(cfg).eval()
        proj = torch.nn.Linear(d_f, n_embd, bias=False)
        stub = types.SimpleNamespace(use_f_gram_embeddings=True, f_gram_model=None,
                                     f_gram_projection=proj, base_model=base)
        captured = {}

        def hook(module, args, kwargs):
            captured["x"] = kwargs["inputs_embeds"].detach().clone()

        h = base.transformer.register_forward_pre_hook(hook, with_kwargs=True)
        input_ids = torch.randint(0, 101, (B, T))
        fg = torch.randn(B, T, d_f)
        pos = torch.randint(0, 32, (B, T)) if with_pos else None
        with torch.no_grad():
            SconeLanguageModel.forward(stub, input_ids=input_ids, f_gram_embeddings=fg, position_ids=pos)
        h.remove()
        name = f"c{ci}"
        out[f"{name}_wte"] = base.transformer.wte.weight.detach().numpy()
        out[f"{name}_wpe"] = base.transformer.wpe.weight.detach().numpy()
        out[f"{name}_proj"] = proj.weight.detach().numpy()
        out[f"{name}_input_ids"] = input_ids.numpy()
        out[f"{name}_fg"] = fg.numpy()
        out[f"{name}_pos"] = pos.numpy() if pos is not None else np.zeros((0,), dtype=np.int64)
        out[f"{name}_embeds"] = captured["x"].numpy()
        cases.append(name)
    out["cases"] = np.asarray(cases)
    np.savez_compressed(os.path.join(HERE, "combine.npz"), **out)
    print("combine.npz:", len(cases), "cases")


CALLER_WORDS = ("the of and to in a is that for it as was with be by on not he this are or his from at which "
                "but have an had they you were their one all we can her has there been if more when will would who so no").split()


def caller_texts(rng, n, lo, hi):
    return [" ".join(rng.choice(CALLER_WORDS, size=int(rng.integers(lo, hi)), p=None).tolist()) for _ in range(n)]


def make_callers(NGramExtractor):
    """The two callers of the match step, driven with a deterministic stub tokenizer (tests/stub_tokenizer.py):
    FGramTokenizer.tokenize / batch_tokenize (scone/tokenization/f_gram_tokenizer.py:38-126) and the f-gram id
    vector of SconeDataset.__getitem__ (scone/data/dataset.py:117-147)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from stub_tokenizer import StubTokenizer
    from scone.tokenization.f_gram_tokenizer import FGramTokenizer
    from scone.data.dataset import SconeDataset
    rng = np.random.default_rng(77)
    tok = StubTokenizer()
    corpus_texts = caller_texts(rng, 60, 5, 60)
    ex = NGramExtractor(max_n=3, min_freq=2, max_f_grams=400)
    ex.fit([tok.encode(t) for t in corpus_texts], verbose=False)
    ft = FGramTokenizer(tok, ex)
    keys, lens = keys_arrays(ex.f_gram_to_id, 3)
    out = {"keys": keys, "lens": lens, "max_n": np.int64(3)}
    texts = caller_texts(rng, 6, 1, 40) + ["the of and to", "zzz qqq"]      # the last one has no f-gram words in the corpus
    out["texts"] = np.asarray(texts)

    def csr_of(tfg, n):
        offsets, ids = [0], []
        for pos in range(n):
            ids.extend(ex.f_gram_to_id[g] for g in tfg[pos])
            offsets.append(len(ids))
        return np.asarray(offsets, dtype=np.int64), np.asarray(ids, dtype=np.int64)

    for i, t in enumerate(texts):                                              # tokenize, one text at a time
        for tag, kw in (("plain", {}), ("trunc", {"max_length": 8, "truncation": True})):
            r = ft.tokenize(t, **kw)
            out[f"tok{i}_{tag}_input_ids"] = np.asarray(r["input_ids"], dtype=np.int64)
            out[f"tok{i}_{tag}_mask"] = np.asarray(r["attention_mask"], dtype=np.int64)
            out[f"tok{i}_{tag}_off"], out[f"tok{i}_{tag}_ids"] = csr_of(r["token_f_grams"], len(r["input_ids"]))
    r = ft.batch_tokenize(texts, max_length=24)                                 # padded + truncated batch
    ids = r["input_ids"].numpy()
    out["batch_input_ids"], out["batch_mask"] = ids, r["attention_mask"].numpy()
    offs, flat = [], []
    for b in range(ids.shape[0]):
        o, f = csr_of(r["token_f_grams"][b], ids.shape[1])
        offs.append(o)
        flat.append(f)
    out["batch_off"] = np.stack(offs)
    out["batch_ids_flat"] = np.concatenate(flat)
    for max_length in (16, 4):                                                  # dataset items (causal_lm)
        ds = SconeDataset(texts, tok, ft, max_length=max_length, task="causal_lm")
        items = [ds[i] for i in range(len(texts))]
        out[f"ds{max_length}_input_ids"] = np.stack([it["input_ids"].numpy() for it in items])
        out[f"ds{max_length}_f_gram_ids"] = np.stack([it["f_gram_ids"].numpy() for it in items])
        out[f"ds{max_length}_f_gram_mask"] = np.stack([it["f_gram_attention_mask"].numpy() for it in items])
    # extract_f_grams (scone/data/preprocessing.py:12-50): tokenise + fit, driven with the same stub
    from scone.data.preprocessing import extract_f_grams
    xf = extract_f_grams(corpus_texts, tok, max_n=3, min_freq=3, max_f_grams=250, verbose=False)
    out["xf_texts"] = np.asarray(corpus_texts)
    out["xf_keys"], out["xf_lens"] = keys_arrays(xf.f_gram_to_id, 3)
    out["xf_args"] = np.asarray([3, 3, 250], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "callers.npz"), **out)
    print("callers.npz:", len(texts), "texts,", len(lens), "f-grams")


def main():
    NGramExtractor, EmbeddingCache, SconeLanguageModel = import_reference()
    make_match(NGramExtractor)
    make_lookup(NGramExtractor, EmbeddingCache)
    make_combine(SconeLanguageModel)
    make_callers(NGramExtractor)


if __name__ == "__main__":
    main()
