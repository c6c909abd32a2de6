"""The step in front of the lookup path: vocabulary extraction (tokenise a corpus, fit the f-gram vocabulary on the GPU).

Only ``extract_f_grams`` of ``scone.data`` is mirrored; the training dataset class is outside the lookup layer (its f-gram
id vector is available as ``FGramTokenizer.flat_f_gram_ids``).
"""

from scone_amd.data import preprocessing as _pre

extract_f_grams = _pre.extract_f_grams

__all__ = ["extract_f_grams"]
