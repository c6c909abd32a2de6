"""Mirror of ``scone.tokenization`` (hot-path part)."""

from scone_amd.tokenization.n_gram_extractor import NGramExtractor

__all__ = ["NGramExtractor"]
