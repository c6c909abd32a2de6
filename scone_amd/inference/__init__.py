"""Mirror of ``scone.inference`` (hot-path part)."""

from scone_amd.inference.embedding_cache import EmbeddingCache

__all__ = ["EmbeddingCache"]
