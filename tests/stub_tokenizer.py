"""A deterministic whitespace tokenizer with the HF call convention the reference's FGramTokenizer /
SconeDataset use (``tokenizer(text | [texts], max_length=, padding=, truncation=, return_tensors="pt")``).
Test infrastructure shared by tests/golden/make_golden.py (drives the REAL reference with it) and the
parity tests (drive scone_amd with it), so both see identical token ids.  No network, no HF hub."""

import zlib

import torch


class StubTokenizer:
    pad_token_id = 0
    mask_token_id = 1

    def __init__(self, vocab_size: int = 61):
        self.vocab_size = vocab_size

    def _ids(self, text):
        return [2 + zlib.crc32(w.encode()) % (self.vocab_size - 2) for w in text.split()]

    def encode(self, text):
        return self._ids(text)

    def __call__(self, text, max_length=None, padding=False, truncation=False, return_tensors=None, add_special_tokens=True):
        single = isinstance(text, str)
        seqs = [self._ids(t) for t in ([text] if single else text)]
        if truncation and max_length is not None:
            seqs = [s[:max_length] for s in seqs]
        if padding == "max_length" and max_length is not None:
            width = max_length
        elif padding:
            width = max(len(s) for s in seqs)
        else:
            width = None
            if len({len(s) for s in seqs}) > 1:
                raise ValueError("ragged batch without padding")
        mask = [[1] * len(s) for s in seqs]
        if width is not None:
            mask = [m + [0] * (width - len(m)) for m in mask]
            seqs = [s + [self.pad_token_id] * (width - len(s)) for s in seqs]
        if return_tensors is None:          # HF returns plain lists (flat for a single text) without return_tensors
            return {"input_ids": seqs[0] if single else seqs, "attention_mask": mask[0] if single else mask}
        return {"input_ids": torch.tensor(seqs, dtype=torch.long), "attention_mask": torch.tensor(mask, dtype=torch.long)}

    def save_pretrained(self, directory):
        import json
        import os
        os.makedirs(directory, exist_ok=True)
        json.dump({"vocab_size": self.vocab_size}, open(os.path.join(directory, "stub_tokenizer.json"), "w"))
