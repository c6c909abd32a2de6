"""Embedding stage of the SCONE language model on MI355X.

What lives here is the part of ``scone.models`` that touches the f-gram lookup: the module that builds
``inputs_embeds = wte + f_gram + wpe`` (one fused HIP pass when an ``EmbeddingCache`` is attached) and a thin
``forward``-compatible wrapper around any GPT-2 style causal LM.  The BERT f-gram encoder of the reference is an
offline table producer and is not part of this package.
"""

from scone_amd.models import language_model as _lm

SconeEmbedding = _lm.SconeEmbedding
SconeLanguageModel = _lm.SconeLanguageModel
fold_projection = _lm.fold_projection

__all__ = ["SconeEmbedding", "SconeLanguageModel", "fold_projection"]
