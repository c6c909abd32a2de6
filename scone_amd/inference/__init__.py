"""The f-gram table + lookup (``EmbeddingCache``) and the generation glue around it (``SconeInferenceEngine``).

Same public names as ``scone.inference``; every lookup is served by the HIP kernels behind ``include/scone_hip.h`` from a
device copy of the table (HBM, or pinned host DRAM), there is no CPU path.
"""

from scone_amd.inference import embedding_cache as _ec
from scone_amd.inference import engine as _en

EmbeddingCache = _ec.EmbeddingCache
SconeInferenceEngine = _en.SconeInferenceEngine

__all__ = ["EmbeddingCache", "SconeInferenceEngine"]
