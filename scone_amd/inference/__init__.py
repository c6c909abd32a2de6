"""Mirror of ``scone.inference`` (hot-path part + inference glue)."""

from scone_amd.inference.embedding_cache import EmbeddingCache
from scone_amd.inference.engine import SconeInferenceEngine

__all__ = ["EmbeddingCache", "SconeInferenceEngine"]
