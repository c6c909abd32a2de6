"""F-gram tokenizer: the caller of the match step, MI355X-native.

Mirrors ``scone/tokenization/f_gram_tokenizer.py`` of the reference: same class name, constructor
(``base_tokenizer, n_gram_extractor``), ``tokenize`` / ``batch_tokenize`` / ``save_pretrained`` /
``from_pretrained`` signatures and return dictionaries.  The base tokenizer is any HF-style callable
(``tokenizer(text, max_length=, padding=, truncation=, return_tensors="pt")`` returning ``input_ids``
and ``attention_mask``); text -> ids stays on the CPU exactly as in the reference, the f-gram match
(``NGramExtractor.get_token_f_grams``, reference lines 77 and 122-123) runs on the GPU.

Differences that follow from batching on a device:

* ``batch_tokenize`` matches the whole ``[B, T]`` batch in ONE ``scone_match_csr`` call (windows never
  cross a sequence boundary) where the reference loops over sequences in Python (``:121-123``);
* the keyword form ``FGramTokenizer(tokenizer=..., n_gram_extractor=...)`` that the reference's own
  callers use (``train.py:290-293``, ``tests/test_language_model.py:43``) is accepted as well;
* additive: :meth:`batch_f_gram_ids` returns the id lists as device CSR without building Python
  tuples, and :meth:`flat_f_gram_ids` is the fixed-length id vector of ``SconeDataset.__getitem__``
  (``scone/data/dataset.py:117-147``).
"""

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from scone_amd.tokenization.n_gram_extractor import NGramExtractor


class FGramTokenizer:
    """Tokenizer wrapper that adds f-gram information (reference: f_gram_tokenizer.py:11-36)."""

    def __init__(self, base_tokenizer=None, n_gram_extractor: Optional[NGramExtractor] = None, *, tokenizer=None) -> None:
        if base_tokenizer is None:
            base_tokenizer = tokenizer
        if base_tokenizer is None or n_gram_extractor is None:
            raise TypeError("FGramTokenizer(base_tokenizer, n_gram_extractor)")
        self.base_tokenizer = base_tokenizer
        self.n_gram_extractor = n_gram_extractor

    # ------------------------------------------------------------------ reference API
    def tokenize(self, text: str, return_f_grams: bool = True, max_length: Optional[int] = None,
                 padding: bool = False, truncation: bool = False) -> Dict[str, Union[List[int], Dict[int, List[Tuple[int, ...]]]]]:
        """f_gram_tokenizer.py:38-80: ``input_ids``, ``attention_mask`` (lists) and ``token_f_grams``
        (position -> list of f-gram tuples, n ascending then window start ascending, duplicates kept)."""
        encoding = self.base_tokenizer(text, max_length=max_length, padding=padding, truncation=truncation,
                                       return_tensors="pt")
        result = {
            "input_ids": torch.as_tensor(encoding["input_ids"]).squeeze(0).tolist(),
            "attention_mask": torch.as_tensor(encoding["attention_mask"]).squeeze(0).tolist(),
        }
        if return_f_grams:
            result["token_f_grams"] = self.n_gram_extractor.get_token_f_grams(result["input_ids"])
        return result

    def batch_tokenize(self, texts: List[str], return_f_grams: bool = True, max_length: Optional[int] = None,
                       padding: bool = True, truncation: bool = True):
        """f_gram_tokenizer.py:82-126: ``input_ids`` / ``attention_mask`` tensors and, per sequence, the
        ``token_f_grams`` dictionary.  Pad tokens take part in the match exactly as in the reference
        (no pad masking there, ``:121-123``)."""
        encodings = self.base_tokenizer(texts, max_length=max_length, padding=padding, truncation=truncation,
                                        return_tensors="pt")
        result = {"input_ids": encodings["input_ids"], "attention_mask": encodings["attention_mask"]}
        if return_f_grams:
            ids = torch.as_tensor(encodings["input_ids"])
            result["token_f_grams"] = self.n_gram_extractor.get_token_f_grams_batch(ids)
        return result

    def save_pretrained(self, save_directory: str) -> None:
        """f_gram_tokenizer.py:128-136."""
        self.base_tokenizer.save_pretrained(save_directory)
        self.n_gram_extractor.save(f"{save_directory}/n_gram_extractor.npy")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, n_gram_extractor_path: Optional[str] = None,
                        base_tokenizer=None) -> "FGramTokenizer":
        """f_gram_tokenizer.py:138-162.  ``base_tokenizer`` may be passed in when the HF hub is not
        reachable (the reference always calls ``AutoTokenizer.from_pretrained``)."""
        if base_tokenizer is None:
            from transformers import AutoTokenizer
            base_tokenizer = AutoTokenizer.from_pretrained(pretrained_model_name_or_path)
        if n_gram_extractor_path is None:
            n_gram_extractor_path = f"{pretrained_model_name_or_path}/n_gram_extractor.npy"
        return cls(base_tokenizer, NGramExtractor.load(n_gram_extractor_path))

    # ------------------------------------------------------------------ additive, device-side
    def batch_f_gram_ids(self, input_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-position f-gram ID lists of a ``[B, T]`` batch as device CSR ``(offsets[B*T+1], ids)`` --
        ``get_token_f_grams`` + the id map of ``embedding_cache.py:173`` without Python tuples."""
        return self.n_gram_extractor.device_index().match_csr(torch.as_tensor(input_ids))

    def flat_f_gram_ids(self, input_ids: Sequence[int], max_length: Optional[int] = None,
                        max_f_grams: int = 10) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(f_gram_ids [max_f_grams] long, f_gram_attention_mask [max_f_grams] float)`` as
        ``SconeDataset.__getitem__`` builds them (dataset.py:117-147): the ids of positions
        ``< max_length`` in position order, cut / zero-padded to ``max_f_grams``."""
        ids = np.asarray(list(input_ids), dtype=np.int64)
        out_ids = torch.zeros(max_f_grams, dtype=torch.long)
        out_mask = torch.zeros(max_f_grams, dtype=torch.float)
        if ids.size == 0:
            return out_ids, out_mask
        off, flat = self.n_gram_extractor.device_index().match_csr(
            torch.as_tensor(ids.clip(-1, 2**31 - 1), dtype=torch.int32))
        stop = ids.size if max_length is None else min(ids.size, int(max_length))
        n = min(int(off[stop].item()), max_f_grams)
        out_ids[:n] = flat[:n].to("cpu", torch.long)
        out_mask[:n] = 1.0
        return out_ids, out_mask
