"""scone_amd -- MI355X-native f-gram embedding lookup / aggregation layer.

Drop-in for the hot path of llmsresearch/scone (``scone.tokenization.NGramExtractor``,
``scone.inference.EmbeddingCache`` and the embedding stage of
``scone.models.SconeLanguageModel.forward``); the work is done by hand-written
gfx950 kernels behind the C ABI in ``include/scone_hip.h``.
"""

from scone_amd.tokenization.n_gram_extractor import NGramExtractor
from scone_amd.tokenization.f_gram_tokenizer import FGramTokenizer
from scone_amd.inference.embedding_cache import EmbeddingCache
from scone_amd.models.language_model import SconeEmbedding, SconeLanguageModel

__all__ = ["NGramExtractor", "FGramTokenizer", "EmbeddingCache", "SconeEmbedding", "SconeLanguageModel"]
__version__ = "0.1.0"
