"""ctypes binding of the host rANS coder in libcrdr_hip.so (stands in for `compressai.ans`, see csrc/rans.cpp)."""
from __future__ import annotations

import numpy as np

from crdr_amd.hip import lib as L


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _tables(cdfs, sizes, offsets):
    cdfs, sizes, offsets = _i32(cdfs), _i32(sizes), _i32(offsets)
    assert cdfs.ndim == 2 and len(sizes) == len(offsets) == cdfs.shape[0]
    return cdfs, sizes, offsets


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets) -> bytes:
    lib = L.load()
    symbols, indexes = _i32(symbols), _i32(indexes)
    assert symbols.shape == indexes.shape
    cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    cap = 4 * len(symbols) + 64
    for _ in range(2):
        out = np.empty(cap, dtype=np.uint8)
        n = lib.crdr_rans_encode_with_indexes(symbols.ctypes.data, indexes.ctypes.data, len(symbols), cdfs.ctypes.data,
                                              cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, cdfs.shape[0],
                                              out.ctypes.data, cap)
        if n >= 0:
            return out[:n].tobytes()
        if n < -(1 << 39):
            L.check(-1, "rans_encode_with_indexes")
        cap = -n
    raise L.CrdrHipError("rans encode: buffer sizing failed")


def decode_with_indexes(data: bytes, indexes, cdfs, cdf_sizes, offsets) -> np.ndarray:
    lib = L.load()
    indexes = _i32(indexes)
    cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(len(indexes), dtype=np.int32)
    L.check(lib.crdr_rans_decode_with_indexes(buf.ctypes.data, len(buf), indexes.ctypes.data, len(indexes), cdfs.ctypes.data,
                                              cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, cdfs.shape[0],
                                              out.ctypes.data), "rans_decode_with_indexes")
    return out


class RansDecoder:
    """Streaming decoder: `set_stream` once, then one `decode_stream` per Charm slice
    (minnen20_charm_context_model.py:201-224)."""

    def __init__(self):
        self._lib = L.load()
        self._h = self._lib.crdr_rans_decoder_create()
        self._buf = None

    def set_stream(self, data: bytes) -> None:
        self._buf = np.frombuffer(data, dtype=np.uint8)
        L.check(self._lib.crdr_rans_decoder_set_stream(self._h, self._buf.ctypes.data, len(self._buf)), "rans set_stream")

    def decode_stream(self, indexes, cdfs, cdf_sizes, offsets) -> np.ndarray:
        indexes = _i32(indexes)
        cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
        out = np.empty(len(indexes), dtype=np.int32)
        L.check(self._lib.crdr_rans_decoder_decode_stream(self._h, indexes.ctypes.data, len(indexes), cdfs.ctypes.data,
                                                          cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, cdfs.shape[0],
                                                          out.ctypes.data), "rans decode_stream")
        return out

    def decode_stream_into(self, indexes: np.ndarray, cdfs, cdf_sizes, offsets, out: np.ndarray) -> np.ndarray:
        """decode_stream without allocations: `indexes` and `out` are caller-owned contiguous int32 arrays (pinned host staging)."""
        assert indexes.dtype == np.int32 and out.dtype == np.int32 and indexes.flags.c_contiguous and out.flags.c_contiguous
        assert out.size >= indexes.size
        cdfs, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
        L.check(self._lib.crdr_rans_decoder_decode_stream(self._h, indexes.ctypes.data, indexes.size, cdfs.ctypes.data,
                                                          cdfs.shape[1], sizes.ctypes.data, offs.ctypes.data, cdfs.shape[0],
                                                          out.ctypes.data), "rans decode_stream")
        return out

    def __del__(self):
        try:
            if self._h:
                self._lib.crdr_rans_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass
