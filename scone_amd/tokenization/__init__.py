"""Mirror of ``scone.tokenization`` (the f-gram vocabulary, the match step and its caller)."""

from scone_amd.tokenization.n_gram_extractor import NGramExtractor
from scone_amd.tokenization.f_gram_tokenizer import FGramTokenizer

__all__ = ["NGramExtractor", "FGramTokenizer"]
