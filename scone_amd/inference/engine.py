"""Inference glue around the fused f-gram lookup.

Mirrors the constructor and the ``generate`` / ``benchmark_inference`` entry points of
``scone/inference/engine.py`` of the reference, with three differences that follow from what the
reference does around the lookup path:

* the per-position Python loop of ``engine.py:234-266`` (ids -> rows -> mean -> scatter) is ONE
  fused GPU pass, ``EmbeddingCache.embed_tokens``, batched over ``[B, T]``;
* the embeddings are actually consumed: the reference hands ``f_gram_embeddings`` to HF
  ``generate``, which never routes them back into ``SconeLanguageModel.forward``
  (``language_model.py:349-376``); here every decoding step makes the call that ``forward`` makes,
  ``transformer(inputs_embeds=wte + f_gram + wpe)`` (``language_model.py:257-264``), so the logits of a
  step equal ``SconeLanguageModel.forward(input_ids)["logits"][:, -1]``;
* with the paper's causal lookup (``lookup_mode="longest_suffix"``) decoding is incremental: a new
  token only needs its own embedding (computed from the last ``max_n`` tokens) and the KV cache.
  With the reference code's covering lookup (``"cover"``) a new token changes the f-gram sets of
  the ``max_n - 1`` positions before it: the KV cache is rolled back over those positions and the last
  ``max_n`` positions are recomputed from a lookup over the last ``2 max_n - 1`` tokens (O(1) positions per
  step; the reference would re-run everything).

The transformer itself is any HF-style causal LM (``.transformer``, ``.lm_head``); tokenisation is
delegated to the tokenizer object the caller passes (``encode`` / ``decode``).
"""

import time
from typing import Dict, List, Optional, Sequence, Union

import torch


class SconeInferenceEngine:
    """``SconeInferenceEngine(model, tokenizer, f_gram_tokenizer, embedding_cache, device, quantization)``
    (reference: engine.py:33-67).  ``model`` is a ``scone_amd.models.SconeLanguageModel`` (or any
    object with ``.base_model.transformer`` / ``.base_model.lm_head``); ``embedding_cache`` must hold a
    table in hidden size (``fold_projection``).  ``f_gram_tokenizer`` is accepted for signature
    compatibility; matching happens on the GPU inside the cache."""

    def __init__(self, model, tokenizer=None, f_gram_tokenizer=None, embedding_cache=None,
                 device: Optional[torch.device] = None, quantization: Optional[str] = None) -> None:
        if embedding_cache is None:
            raise ValueError("SconeInferenceEngine needs an embedding_cache")
        self.model = model
        self.tokenizer = tokenizer
        self.f_gram_tokenizer = f_gram_tokenizer
        self.embedding_cache = embedding_cache
        self.quantization = quantization
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if self.device.type != "cuda":
            raise RuntimeError("scone_amd: the lookup path runs on the GPU (no CPU fallback)")
        if quantization == "fp16":                         # engine.py:118-121
            self.model.half()
        elif quantization not in (None, "int8", "int4"):
            raise ValueError(f"Unknown quantization mode: {quantization}")
        # int8 / int4 in the reference quantise the transformer's nn.Linear layers (engine.py:75-116),
        # which is outside this layer; the f-gram TABLE format is chosen on the cache (table_format=).
        self.model.to(self.device)
        self.base = model.base_model
        self.causal = getattr(embedding_cache, "lookup_mode", "cover") == "longest_suffix"
        self.max_n = embedding_cache.n_gram_extractor.max_n

    @classmethod
    def from_pretrained(cls, model_path: str, tokenizer_path: Optional[str] = None,
                        f_gram_tokenizer_path: Optional[str] = None, embedding_cache_path: Optional[str] = None,
                        device: Optional[torch.device] = None, quantization: Optional[str] = None,
                        use_memory_map: bool = True, *, tokenizer=None, table_format: str = "fp32",
                        lookup_mode: str = "cover") -> "SconeInferenceEngine":
        """``SconeInferenceEngine.from_pretrained`` of the reference (engine.py:129-190): same arguments, same default
        paths (``tokenizer_path`` / ``f_gram_tokenizer_path`` = ``model_path``, the cache at
        ``{model_path}/embedding_cache.npy``), same loading order -- model, tokenizer, f-gram tokenizer
        (``n_gram_extractor.npy``), embedding cache (``EmbeddingCache.load(path, extractor, use_memory_map=...)``, the
        kwarg the reference passes at :180 and its own ``load`` rejects).  Additive: ``tokenizer=`` supplies the base
        tokenizer object when ``AutoTokenizer`` cannot reach its files; ``table_format`` / ``lookup_mode`` choose the
        device table's row format and the lookup.

        The cache file holds rows of the f-gram model's size; the checkpoint's bias-free ``f_gram_projection`` is folded
        into them here (``fold_projection``: ``proj(mean(rows)) == mean(proj(rows))``), so that the fused lookup adds
        hidden-size rows straight to ``wte`` -- the result ``SconeLanguageModel.forward`` computes with its GEMM."""
        import numpy as np
        from scone_amd.inference.embedding_cache import EmbeddingCache
        from scone_amd.models.language_model import SconeLanguageModel, fold_projection
        from scone_amd.tokenization.f_gram_tokenizer import FGramTokenizer
        if tokenizer_path is None:
            tokenizer_path = model_path
        if f_gram_tokenizer_path is None:
            f_gram_tokenizer_path = model_path
        if embedding_cache_path is None:
            embedding_cache_path = f"{model_path}/embedding_cache.npy"
        model = SconeLanguageModel.from_pretrained(model_path)
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer_path)
        f_gram_tokenizer = FGramTokenizer.from_pretrained(f_gram_tokenizer_path, base_tokenizer=tokenizer)
        ex = f_gram_tokenizer.n_gram_extractor
        loaded = EmbeddingCache.load(embedding_cache_path, ex, use_memory_map=use_memory_map)
        hidden = model.base_model.transformer.wte.weight.shape[1]
        # rows of the file, in id order, projected to hidden size and stored in the device table's format
        if loaded.use_memory_map:
            ids = np.arange(loaded.memory_mapped_embeddings.shape[0], dtype=np.int64)
            rows = torch.from_numpy(np.array(loaded.memory_mapped_embeddings, dtype=np.float32))     # a writable copy of the read-only map
        else:
            ids = np.asarray(sorted(loaded.embeddings.keys()), dtype=np.int64)
            rows = torch.from_numpy(np.stack([np.asarray(loaded.embeddings[int(i)], dtype=np.float32).reshape(-1) for i in ids])
                                    if len(ids) else np.zeros((0, loaded.embedding_dim), dtype=np.float32))
        if model.f_gram_projection is not None and rows.shape[1] == model.f_gram_projection.weight.shape[1]:
            rows = fold_projection(rows, model.f_gram_projection.weight.detach().cpu())
        if rows.shape[1] != hidden:
            raise ValueError(f"embedding cache rows have {rows.shape[1]} dims; the model needs {hidden} "
                             "(or rows of the f_gram_projection's input size)")
        cache = EmbeddingCache(ex, hidden, table_format=table_format, lookup_mode=lookup_mode)
        cache.cache_embeddings(ids.tolist(), rows, verbose=False)
        model.embed.embedding_cache = cache
        return cls(model=model, tokenizer=tokenizer, f_gram_tokenizer=f_gram_tokenizer, embedding_cache=cache, device=device,
                   quantization=quantization)

    # ------------------------------------------------------------------ lookup
    def _weights(self):
        wte = self.base.transformer.wte.weight.detach().contiguous()
        wpe = self.base.transformer.wpe.weight.detach().contiguous()
        return wte, wpe

    def embed(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``inputs_embeds [B, T, H]`` for the transformer: fused match + gather + mean + wte + wpe."""
        wte, wpe = self._weights()
        return self.embedding_cache.embed_tokens(input_ids, wte=wte, wpe=wpe, position_ids=position_ids,
                                                 out_dtype=wte.dtype)

    def _encode(self, text: Union[str, Sequence[int], torch.Tensor]) -> torch.Tensor:
        if isinstance(text, str):
            if self.tokenizer is None:
                raise ValueError("a tokenizer is needed to generate from text")
            ids = self.tokenizer.encode(text)
        else:
            ids = text
        ids = torch.as_tensor(ids, dtype=torch.long)
        return ids.reshape(1, -1) if ids.dim() == 1 else ids

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate_ids(self, input_ids: torch.Tensor, max_length: int = 50, min_length: int = 0, do_sample: bool = True,
                     temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0, eos_token_id: Optional[int] = None,
                     generator: Optional[torch.Generator] = None, repetition_penalty: float = 1.0) -> torch.Tensor:
        """Decode until ``max_length`` total tokens.  Batch of prompts of equal length ``[B, T0]``.
        ``repetition_penalty`` as HF's logits processor: the logit of every token already in the sequence is divided by
        it when positive, multiplied when negative."""
        ids = input_ids.to(self.device)
        B = ids.shape[0]
        past = None
        finished = torch.zeros(B, dtype=torch.bool, device=self.device)
        while ids.shape[1] < max_length:
            T = ids.shape[1]
            if self.causal and past is not None:
                # only the new position: its f-gram is decided by the last max_n tokens
                tail = ids[:, -self.max_n:]
                pos = torch.arange(T - tail.shape[1], T, device=self.device).unsqueeze(0).expand(B, -1)
                x = self.embed(tail, position_ids=pos)[:, -1:, :]
                out = self.base.transformer(inputs_embeds=x, past_key_values=past, position_ids=pos[:, -1:],
                                            use_cache=True, return_dict=True)
            elif past is not None and hasattr(past, "crop"):
                # covering lookup (the reference code): the new token joins f-grams that also cover the max_n - 1
                # positions before it, so THEIR input embeddings change -- roll the KV cache back over them and
                # recompute the last max_n positions; every f-gram covering one of those lies inside the last
                # 2 max_n - 1 tokens, so that window is all the lookup needs.  O(1) positions per step instead of T.
                redo = min(self.max_n, T)
                if redo > 1:
                    past.crop(-(redo - 1))
                win = min(T, 2 * self.max_n - 1)
                pos = torch.arange(T - win, T, device=self.device).unsqueeze(0).expand(B, -1)
                x = self.embed(ids[:, -win:], position_ids=pos)[:, -redo:, :]
                out = self.base.transformer(inputs_embeds=x, past_key_values=past, position_ids=pos[:, -redo:],
                                            use_cache=True, return_dict=True)
            else:
                pos = torch.arange(T, device=self.device).unsqueeze(0).expand(B, -1)
                x = self.embed(ids, position_ids=pos)
                # the same call SconeLanguageModel.forward makes (language_model.py:257-264)
                out = self.base.transformer(inputs_embeds=x, position_ids=pos, use_cache=True, return_dict=True)
            past = out.past_key_values
            logits = self.base.lm_head(out.last_hidden_state[:, -1, :]).float()
            logits = self._penalise(logits, ids, repetition_penalty)
            if eos_token_id is not None and T < min_length:
                logits[:, eos_token_id] = -float("inf")
            nxt = self._pick(logits, do_sample, temperature, top_k, top_p, generator)
            if eos_token_id is not None:
                nxt = torch.where(finished, torch.full_like(nxt, eos_token_id), nxt)
                finished |= nxt == eos_token_id
            ids = torch.cat([ids, nxt.unsqueeze(1)], dim=1)
            if eos_token_id is not None and bool(finished.all()):
                break
        return ids

    @staticmethod
    def _penalise(logits: torch.Tensor, ids: torch.Tensor, penalty: float) -> torch.Tensor:
        if penalty == 1.0:
            return logits
        seen = torch.gather(logits, 1, ids)
        seen = torch.where(seen > 0, seen / penalty, seen * penalty)
        return logits.scatter(1, ids, seen)

    @torch.no_grad()
    def _step_logits(self, ids: torch.Tensor) -> torch.Tensor:
        """Next-token logits ``[N, V]`` for ``N`` sequences of equal length: the lookup over the whole prefix and a full
        transformer pass (what ``SconeLanguageModel.forward`` does) -- used where sequences are re-ordered between steps."""
        T = ids.shape[1]
        pos = torch.arange(T, device=self.device).unsqueeze(0).expand(ids.shape[0], -1)
        h = self.base.transformer(inputs_embeds=self.embed(ids, position_ids=pos), position_ids=pos, return_dict=True)
        return self.base.lm_head(h.last_hidden_state[:, -1, :]).float()

    @torch.no_grad()
    def beam_search_ids(self, input_ids: torch.Tensor, max_length: int = 50, min_length: int = 0, num_beams: int = 4,
                        num_return_sequences: int = 1, eos_token_id: Optional[int] = None, repetition_penalty: float = 1.0,
                        length_penalty: float = 1.0):
        """Beam search over the f-gram-augmented model for ONE prompt ``[1, T0]``: every step scores all beams with the
        fused lookup over their whole prefix (beams are re-ordered between steps, so no KV cache is carried), keeps the
        ``num_beams`` best continuations by summed log-probability; a beam that emits ``eos_token_id`` is finished and
        ranked by ``score / length ** length_penalty`` (HF's convention).  Returns ``(sequences, scores)``, best first."""
        ids = input_ids.to(self.device)
        if ids.shape[0] != 1:
            raise ValueError("beam search takes one prompt")
        if not 1 <= num_return_sequences <= num_beams:
            raise ValueError("num_return_sequences must be in 1..num_beams")
        beams = ids                                          # [n, T]
        scores = torch.zeros(1, device=self.device)
        done = []                                            # (normalised score, sequence)
        while beams.shape[1] < max_length and beams.shape[0] > 0:
            T = beams.shape[1]
            logits = self._penalise(self._step_logits(beams), beams, repetition_penalty)
            if eos_token_id is not None and T < min_length:
                logits[:, eos_token_id] = -float("inf")
            logp = torch.log_softmax(logits, dim=-1) + scores[:, None]
            V = logp.shape[1]
            top = torch.topk(logp.reshape(-1), min(2 * num_beams, logp.numel()))
            nb, ns = [], []
            for sc, flat in zip(top.values.tolist(), top.indices.tolist()):
                b, tok = divmod(flat, V)
                seq = torch.cat([beams[b], torch.tensor([tok], device=self.device)])
                if eos_token_id is not None and tok == eos_token_id:
                    done.append((sc / (seq.numel() ** length_penalty), seq))
                else:
                    nb.append(seq)
                    ns.append(sc)
                if len(nb) == num_beams:
                    break
            if len(done) >= num_beams and nb and max(d[0] for d in done) >= ns[0] / ((T + 1) ** length_penalty):
                nb = []                                      # no live beam can still beat the finished ones (scores only fall)
            beams = torch.stack(nb) if nb else beams[:0]
            scores = torch.tensor(ns, device=self.device)
        for sc, seq in zip(scores.tolist(), beams):
            done.append((sc / (seq.numel() ** length_penalty), seq))
        done.sort(key=lambda x: -x[0])
        return [d[1] for d in done[:num_return_sequences]], [d[0] for d in done[:num_return_sequences]]

    @staticmethod
    def _pick(logits, do_sample, temperature, top_k, top_p, generator):
        if not do_sample:
            return logits.argmax(dim=-1)
        logits = logits / max(temperature, 1e-6)
        if top_k and top_k > 0:
            kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[:, -1:]
            logits = logits.masked_fill(logits < kth, -float("inf"))
        if top_p < 1.0:
            srt, idx = torch.sort(logits, descending=True, dim=-1)
            cum = torch.softmax(srt, dim=-1).cumsum(dim=-1)
            drop = cum - torch.softmax(srt, dim=-1) > top_p
            srt = srt.masked_fill(drop, -float("inf"))
            logits = torch.full_like(logits, -float("inf")).scatter(-1, idx, srt)
        return torch.multinomial(torch.softmax(logits, dim=-1), 1, generator=generator).squeeze(1)

    def generate(self, text, max_length: int = 50, min_length: int = 0, do_sample: bool = True, num_beams: int = 1,
                 temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0, repetition_penalty: float = 1.0,
                 num_return_sequences: int = 1) -> List:
        """Same arguments as the reference's ``generate`` (engine.py:192-204).  ``num_beams > 1`` runs
        :meth:`beam_search_ids` (deterministic; sampling applies to ``num_beams == 1`` only)."""
        eos = getattr(self.tokenizer, "eos_token_id", None)
        if num_beams > 1:
            seqs, _ = self.beam_search_ids(self._encode(text)[:1], max_length=max_length, min_length=min_length,
                                           num_beams=num_beams, num_return_sequences=num_return_sequences, eos_token_id=eos,
                                           repetition_penalty=repetition_penalty)
            out = seqs
        else:
            ids = self._encode(text).repeat(num_return_sequences, 1) if num_return_sequences > 1 else self._encode(text)
            out = self.generate_ids(ids, max_length=max_length, min_length=min_length, do_sample=do_sample,
                                    temperature=temperature, top_k=top_k, top_p=top_p, eos_token_id=eos,
                                    repetition_penalty=repetition_penalty)
        if self.tokenizer is None:
            return [o.tolist() for o in out]
        return [self.tokenizer.decode(o.tolist()) for o in out]

    def benchmark_inference(self, text, max_length: int = 50, num_runs: int = 10, warmup_runs: int = 2) -> Dict[str, float]:
        """engine.py:292-395: greedy decoding timed with synchronisation around the runs."""
        ids = self._encode(text)
        for _ in range(warmup_runs):
            self.generate_ids(ids, max_length=max_length, do_sample=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(num_runs):
            self.generate_ids(ids, max_length=max_length, do_sample=False)
            torch.cuda.synchronize()
        total = time.time() - t0
        avg = total / num_runs
        return {"total_time_seconds": total, "avg_time_seconds": avg, "tokens_per_second": max_length / avg,
                "quantization": self.quantization}
