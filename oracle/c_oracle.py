"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/oracle.c (the plain-C restatement of the
reference's lookup path).  Allowed users: tests/, __graft_entry__.smoke(), bench.py's baseline leg."""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = C.CDLL(_LIB)
        _lib.oracle_index_new.restype = C.c_void_p
        _lib.oracle_index_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        _lib.oracle_index_free.argtypes = [C.c_void_p]
        _lib.oracle_match_csr.restype = C.c_int64
        _lib.oracle_match_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.oracle_embed_batch.restype = C.c_int64
        _lib.oracle_embed_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                            C.c_void_p, C.c_int]
    return _lib


class COracle:
    def __init__(self, keys: np.ndarray, lens: np.ndarray, max_n: int):
        self.keys = np.ascontiguousarray(keys, dtype=np.uint32)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint8)
        self.max_n = int(max_n)
        self._ix = C.c_void_p(lib().oracle_index_new(self.keys.ctypes.data, self.lens.ctypes.data, len(self.lens), self.max_n))

    def __del__(self):
        try:
            lib().oracle_index_free(self._ix)
        except Exception:
            pass

    def match_csr(self, tok):
        tok = np.ascontiguousarray(tok, dtype=np.int64).reshape(-1)
        T = tok.shape[0]
        off = np.zeros(T + 1, dtype=np.int64)
        ids = np.zeros(T * self.max_n * (self.max_n + 1) // 2 + 1, dtype=np.int64)
        n = lib().oracle_match_csr(self._ix, tok.ctypes.data, T, off.ctypes.data, ids.ctypes.data)
        return off, ids[:n]

    def embed(self, table_f32: np.ndarray, tok: np.ndarray, reduce: str = "mean", nthreads: int = 1):
        table = np.ascontiguousarray(table_f32, dtype=np.float32)
        tok = np.ascontiguousarray(tok, dtype=np.int64)
        if tok.ndim == 1:
            tok = tok[None, :]
        B, T = tok.shape
        out = np.empty((B, T, table.shape[1]), dtype=np.float32)
        total = lib().oracle_embed_batch(self._ix, table.ctypes.data, table.shape[1], tok.ctypes.data, B, T,
                                         1 if reduce == "mean" else 0, out.ctypes.data, int(nthreads))
        return out, total
