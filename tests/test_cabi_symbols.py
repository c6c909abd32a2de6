"""CPU: the C-ABI library loads and exports every symbol include/scone_hip.h declares;
error behaviour that needs no GPU."""

import ctypes as C
import os
import re

import pytest

from scone_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "scone_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scone_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.SIGNATURES.keys()) == declared


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.scone_abi_version() == _lib.ABI_VERSION
    assert lib.scone_strerror(0) == b"ok"
    assert lib.scone_strerror(_lib.EINVAL) == b"invalid argument"


def test_enum_values_match_header():
    text = open(os.path.join(ROOT, "include", "scone_hip.h")).read()
    for name, val in re.findall(r"\b(SCONE_[A-Z0-9_]+)\s*=\s*(-?\d+)", text):
        py = name.replace("SCONE_", "")
        if hasattr(_lib, py):
            assert getattr(_lib, py) == int(val), name


def test_cfg_struct_layout():
    assert C.sizeof(_lib.SconeCfg) == 80
    assert _lib.SconeCfg.n_rows.offset == 24 and _lib.SconeCfg.index_capacity.offset == 48
    assert _lib.SconeCfg.hot_rows.offset == 56 and _lib.SconeCfg.lookup_mode.offset == 64
    assert _lib.SconeCfg.stage_tokens.offset == 68 and _lib.SconeCfg.cache_rows.offset == 72


def test_create_rejects_bad_arguments_without_gpu_work():
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.scone_create(None, C.byref(h)) == _lib.EINVAL
    cfg = _lib.SconeCfg(4, 0, 3, 768, _lib.FMT_I8, 0, 10, 0, 0, 0, 0, 0, 0, 0)       # wrong struct_size
    assert lib.scone_create(C.byref(cfg), C.byref(h)) == _lib.EINVAL
    cfg = _lib.SconeCfg(C.sizeof(_lib.SconeCfg), 0, 7, 768, _lib.FMT_I8, 0, 10, 0, 0, 0, 0, 0, 0)   # max_n = 7
    assert lib.scone_create(C.byref(cfg), C.byref(h)) == _lib.EINVAL
    assert b"max_n" in lib.scone_last_error(None)
    lib.scone_destroy(None)                                                # no-op


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="HIP extension not built"):
        _lib.lib()


def test_lookup_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scone_amd import EmbeddingCache, NGramExtractor
    ex = NGramExtractor(max_n=2, min_freq=1).fit([[1, 2, 3]], verbose=False)
    cache = EmbeddingCache(ex, 16)
    cache.cache_embeddings(list(range(len(ex))), torch.zeros(len(ex), 16), verbose=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cache.get_embeddings([0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ex.get_token_f_grams([1, 2])
