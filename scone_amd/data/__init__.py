"""Mirror of ``scone.data`` (the part that produces the f-gram vocabulary the lookup path consumes)."""

from scone_amd.data.preprocessing import extract_f_grams

__all__ = ["extract_f_grams"]
