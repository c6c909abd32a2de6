"""The f-gram vocabulary (index + GPU match, GPU ``fit``) and the tokenizer wrapper that calls it.

Same public names as ``scone.tokenization``; matching runs on the device index behind ``include/scone_hip.h``.
"""

from scone_amd.tokenization import f_gram_tokenizer as _ft
from scone_amd.tokenization import n_gram_extractor as _ng

NGramExtractor = _ng.NGramExtractor
FGramTokenizer = _ft.FGramTokenizer

__all__ = ["NGramExtractor", "FGramTokenizer"]
