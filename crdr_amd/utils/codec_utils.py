"""Bitstream container of the codec (byte-compatible with src/utils/codec_utils.py:12-143).

Header: u16-LE H, u16-LE W, u8 floor(max|y_hat|) [, u8 floor(16*rate_ind) for the multi-rate model].
File:   for each of (header, z string, y string): u32-LE length followed by the payload.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple, Union

import torch


class HeaderHandler:
    def __init__(self, use_non_zero_ind: bool = False):
        if use_non_zero_ind:
            raise NotImplementedError("non-zero channel index is not used by any shipped CRDR config")
        self.use_non_zero_ind = False

    @staticmethod
    def check_img_size(img_size) -> None:
        assert len(img_size) == 2 and all(isinstance(v, int) for v in img_size), img_size

    @staticmethod
    def _max_sample(y_hat: torch.Tensor) -> int:
        v = int(torch.max(torch.abs(y_hat)))
        if not 0 <= v <= 255:  # numpy>=2 raises for out-of-range uint8 in the reference too
            raise OverflowError(f"max |y_hat| = {v} does not fit the u8 header field")
        return v

    def encode(self, img_size: Tuple[int, int], y_hat: torch.Tensor) -> bytes:
        self.check_img_size(img_size)
        return struct.pack("<HHB", img_size[0], img_size[1], self._max_sample(y_hat))

    def decode(self, header: bytes) -> Dict:
        h, w, mx = struct.unpack("<HHB", header[:5])
        return {"img_size": (h, w), "max_sample": mx}


class MultiRateHeaderHandler(HeaderHandler):
    def encode(self, img_size: Tuple[int, int], y_hat: torch.Tensor, rate_ind: Union[torch.Tensor, float]) -> bytes:
        if isinstance(rate_ind, torch.Tensor):
            assert rate_ind.numel() == 1
            rate_ind = float(rate_ind.item())
        q16 = int(rate_ind * 16)
        self.check_img_size(img_size)
        return struct.pack("<HHBB", img_size[0], img_size[1], self._max_sample(y_hat), q16)

    def decode(self, header: bytes) -> Dict:
        h, w, mx, q16 = struct.unpack("<HHBB", header[:6])
        return {"img_size": (h, w), "max_sample": mx, "rate_ind": float(q16) / 16}


def save_byte_strings(save_path: str, string_list: List[bytes]) -> None:
    with open(save_path, "wb") as f:
        for s in string_list:
            f.write(struct.pack("<I", len(s)))
            f.write(s)


def load_byte_strings(load_path: str) -> List[bytes]:
    out = []
    with open(load_path, "rb") as f:
        while True:
            head = f.read(4)
            if not head:
                break
            (n,) = struct.unpack("<I", head)
            out.append(f.read(n))
    return out
