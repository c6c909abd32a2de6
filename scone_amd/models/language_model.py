"""Embedding stage of the SCONE language model, MI355X-native.

Mirrors the hot part of ``scone/models/language_model.py`` of the reference:
``SconeLanguageModel.forward(input_ids, ..., f_gram_embeddings=...)`` keeps its
signature and its returned dict; what is new is *where* the tensor handed to the
transformer as ``inputs_embeds`` (language_model.py:234-258) comes from:

* with an attached :class:`scone_amd.inference.EmbeddingCache` and no explicit
  ``f_gram_embeddings`` the whole chain -- n-gram match, row gather, dequantise,
  mean, ``+ wte(input_ids)``, ``+ wpe(position_ids)`` -- is one fused pass of the
  HIP kernels (``scone_embed``), the bias-free projection
  (language_model.py:172-176) having been folded into the table offline
  (:func:`fold_projection`; ``proj(mean(rows)) == mean(proj(rows))``);
* with explicit ``f_gram_embeddings`` the reference arithmetic is applied as is
  (projection GEMM through rocBLAS, two adds) on the device the tensors live on.

The transformer body is not part of this layer: any HF-style causal LM exposing
``.transformer(inputs_embeds=...)``, ``.transformer.wte`` / ``.wpe`` and ``.lm_head``
(GPT-2 on PyTorch-ROCm) is wrapped unchanged.
"""

from typing import Dict, Optional

import torch
from torch import nn


def fold_projection(table_f32: torch.Tensor, projection_weight: torch.Tensor) -> torch.Tensor:
    """Pre-project f-gram rows ``[N, d_f]`` with the bias-free ``nn.Linear(d_f, H)`` weight ``[H, d_f]``
    (language_model.py:172-176, :235-236) so that lookups return hidden-size rows."""
    return torch.nn.functional.linear(table_f32.to(torch.float32), projection_weight.to(torch.float32))


class SconeEmbedding(nn.Module):
    """``inputs_embeds = wte(input_ids) + proj(f_gram_embeddings) + wpe(position_ids)``
    (language_model.py:234-254) with an optional fused lookup through an ``EmbeddingCache``."""

    def __init__(self, wte: nn.Embedding, wpe: nn.Embedding, f_gram_projection: Optional[nn.Linear] = None,
                 embedding_cache=None) -> None:
        super().__init__()
        self.wte = wte
        self.wpe = wpe
        self.f_gram_projection = f_gram_projection
        self.embedding_cache = embedding_cache     # rows must already be in hidden size (fold_projection)

    def forward(self, input_ids: torch.Tensor, f_gram_embeddings: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, use_f_gram_embeddings: bool = True) -> torch.Tensor:
        if use_f_gram_embeddings and f_gram_embeddings is None and self.embedding_cache is not None:
            if self.embedding_cache.embedding_dim != self.wte.weight.shape[1]:
                raise ValueError("fused lookup needs a table in hidden size: fold the projection into it "
                                 "(scone_amd.models.language_model.fold_projection)")
            return self.embedding_cache.embed_tokens(
                input_ids, reduce="mean", wte=self.wte.weight.detach().contiguous(),
                wpe=self.wpe.weight.detach().contiguous(), position_ids=position_ids,
                out_dtype=self.wte.weight.dtype)
        if use_f_gram_embeddings and f_gram_embeddings is not None and self.f_gram_projection is not None:
            f_gram_embeddings = self.f_gram_projection(f_gram_embeddings)           # :235-236
        base_embeddings = self.wte(input_ids)                                        # :239
        if use_f_gram_embeddings and f_gram_embeddings is not None:
            combined = base_embeddings + f_gram_embeddings                           # :242-243
        else:
            combined = base_embeddings
        if position_ids is None:                                                     # :248-251
            position_ids = torch.arange(0, input_ids.size(1), dtype=torch.long,
                                        device=input_ids.device).unsqueeze(0)
        return combined + self.wpe(position_ids)                                     # :253-254


class SconeLanguageModel(nn.Module):
    """Reference-compatible wrapper: same ``forward`` signature and outputs as
    ``scone.models.SconeLanguageModel`` (language_model.py:181-289) around any GPT-2-style
    ``base_model``; the embedding stage is :class:`SconeEmbedding`."""

    def __init__(self, base_model: nn.Module, f_gram_projection: Optional[nn.Linear] = None,
                 embedding_cache=None, use_f_gram_embeddings: bool = True, f_gram_model: Optional[nn.Module] = None):
        super().__init__()
        self.base_model = base_model
        self.f_gram_model = f_gram_model
        self.f_gram_projection = f_gram_projection
        self.use_f_gram_embeddings = use_f_gram_embeddings
        self.embed = SconeEmbedding(base_model.transformer.wte, base_model.transformer.wpe, f_gram_projection,
                                    embedding_cache)

    @classmethod
    def from_pretrained(cls, model_path: str, embedding_cache=None, **kwargs) -> "SconeLanguageModel":
        """Load a checkpoint directory written by the reference's ``SconeLanguageModel.save_pretrained`` (a HF
        ``PreTrainedModel``: ``config.json`` with ``SconeConfig``'s fields, language_model.py:12-98, and the weights as
        ``model.safetensors`` or ``pytorch_model.bin`` with the module names of language_model.py:125-176 --
        ``base_model.*`` = ``GPT2LMHeadModel``, ``f_gram_projection.weight``, ``f_gram_model.*``).

        The transformer is rebuilt from the config exactly as ``__init__`` builds it (:125-139: GPT-2 with the config's
        sizes; no hub access needed), ``base_model.*`` and ``f_gram_projection.weight`` are loaded; the BERT-style
        ``f_gram_model`` (:141-170) produces the table offline and is not needed to serve a precomputed cache, so its
        weights are skipped."""
        import json
        import os
        from transformers import GPT2Config, GPT2LMHeadModel
        cfg = json.load(open(os.path.join(model_path, "config.json")))
        hidden = int(cfg.get("hidden_size", 768))
        drop = float(cfg.get("hidden_dropout_prob", 0.1))
        base = GPT2LMHeadModel(GPT2Config(
            vocab_size=int(cfg.get("vocab_size", 30522)), n_positions=int(cfg.get("max_position_embeddings", 512)),
            n_embd=hidden, n_layer=int(cfg.get("num_hidden_layers", 12)), n_head=int(cfg.get("num_attention_heads", 12)),
            resid_pdrop=drop, embd_pdrop=drop, attn_pdrop=float(cfg.get("attention_probs_dropout_prob", 0.1)),
            layer_norm_epsilon=float(cfg.get("layer_norm_eps", 1e-12))))
        st = os.path.join(model_path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            state = load_file(st)
        else:
            state = torch.load(os.path.join(model_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        base_state = {k[len("base_model."):]: v for k, v in state.items() if k.startswith("base_model.")}
        if "lm_head.weight" not in base_state and "transformer.wte.weight" in base_state:
            base_state["lm_head.weight"] = base_state["transformer.wte.weight"]      # tied weights are saved once
        missing, unexpected = base.load_state_dict(base_state, strict=False)
        missing = [k for k in missing if not k.endswith((".attn.bias", ".attn.masked_bias"))]   # buffers, not weights
        if missing or unexpected:
            raise ValueError(f"checkpoint does not match the config: missing {missing[:4]}, unexpected {list(unexpected)[:4]}")
        use_fg = bool(cfg.get("use_f_gram_embeddings", True))
        proj = None
        if use_fg and "f_gram_projection.weight" in state:
            w = state["f_gram_projection.weight"]
            proj = nn.Linear(w.shape[1], w.shape[0], bias=False)
            proj.weight.data.copy_(w)
        return cls(base.eval(), proj, embedding_cache, use_f_gram_embeddings=use_fg)

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                token_type_ids: Optional[torch.Tensor] = None, position_ids: Optional[torch.Tensor] = None,
                f_gram_ids: Optional[torch.Tensor] = None, f_gram_attention_mask: Optional[torch.Tensor] = None,
                f_gram_embeddings: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        if (self.use_f_gram_embeddings and f_gram_embeddings is None and f_gram_ids is not None
                and self.f_gram_model is not None):
            f_gram_embeddings = self.f_gram_model(                                   # :218-232
                input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                position_ids=position_ids, f_gram_ids=f_gram_ids, f_gram_attention_mask=f_gram_attention_mask,
                output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                return_dict=True)["f_gram_embeddings"]
        embeddings = self.embed(input_ids, f_gram_embeddings, position_ids, self.use_f_gram_embeddings)
        outputs = self.base_model.transformer(                                       # :257-264
            inputs_embeds=embeddings, attention_mask=attention_mask, token_type_ids=token_type_ids,
            output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=True)
        hidden_states = outputs.last_hidden_state
        logits = self.base_model.lm_head(hidden_states)                              # :267-268
        loss = None
        if labels is not None:                                                       # :271-282
            shift_logits = logits[..., :-1, :].contiguous()
            shift_labels = labels[..., 1:].contiguous()
            loss = nn.CrossEntropyLoss()(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))
        return {"loss": loss, "logits": logits, "hidden_states": outputs.hidden_states,
                "attentions": outputs.attentions}
