"""Quantised CDF tables (compressai 1.2.4 `_pmf_to_cdf` + C++ `pmf_to_quantized_cdf`, called by the reference via
`codec_setup`: hyperprior_model.py:120-124). The quantiser itself is crdr_pmf_to_quantized_cdf in the library."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from crdr_amd.hip import lib as L


def std_cdf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def pmf_to_quantized_cdf(pmf: np.ndarray, precision: int = 16) -> np.ndarray:
    lib = L.load()
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(len(pmf) + 1, dtype=np.uint32)
    L.check(lib.crdr_pmf_to_quantized_cdf(pmf.ctypes.data, len(pmf), precision, out.ctypes.data), "pmf_to_quantized_cdf")
    return out.astype(np.int32)


def pmf_to_cdf_table(pmf: np.ndarray, tail_mass: np.ndarray, pmf_length: np.ndarray, max_length: int, precision: int = 16) -> np.ndarray:
    table = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i][: int(pmf_length[i])], np.asarray([tail_mass[i]], dtype=np.float32)]).astype(np.float32)
        cdf = pmf_to_quantized_cdf(prob, precision)
        table[i, : len(cdf)] = cdf
    return table
