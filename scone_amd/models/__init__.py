"""Mirror of ``scone.models`` (embedding stage of the language model)."""

from scone_amd.models.language_model import SconeEmbedding, SconeLanguageModel

__all__ = ["SconeEmbedding", "SconeLanguageModel"]
