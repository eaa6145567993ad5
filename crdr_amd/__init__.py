"""crdr_amd: MI355X-native (gfx950) implementation of the CRDR codec hot path.

Arithmetic lives in hand-written HIP behind the C ABI of include/crdr_hip.h (crdr_amd/csrc); this package is the
Python host that mirrors the reference's registry / config / model / trainer surface on top of it.
"""
__version__ = "0.1.0"
