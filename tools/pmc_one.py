"""Run one conv shape a few times (for rocprofv3 --pmc). Usage: [CRDR_PRECISION=bf16x6|bf16x3] python3 tools/pmc_one.py CIN H COUT K STRIDE TRANSPOSED [BS] [ALGO]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crdr_amd.hip import ops, lib as L  # noqa: E402
import ctypes as C  # noqa: E402

ci, h, co, k, s, tr = [int(v) for v in sys.argv[1:7]]
bs = int(sys.argv[7]) if len(sys.argv) > 7 else 16
algo = int(sys.argv[8]) if len(sys.argv) > 8 else 0
dev = torch.device("cuda:0")
ops.MATRIX_BF16X6 = os.environ.get("CRDR_PRECISION") == "bf16x6"
ops.MATRIX_BF16X3 = os.environ.get("CRDR_PRECISION") == "bf16x3"
p = k // 2
oh = ops.conv_out_size(h, k, s, p, bool(tr), out_pad=(1 if (tr and s == 2) else 0))
x = torch.randn(bs, ci, h, h, device=dev)
x, _ = ops.nhwc(x.contiguous(memory_format=torch.channels_last) if ci % 4 == 0 else x)
w = torch.randn(*((ci, co, k, k) if tr else (co, ci, k, k)), device=dev) * 0.02
wf = ops.pack_weight(w, transpose=bool(tr))
y = ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh))
if algo < 0:   # autotune this shape first (trial launches come before the measured ones)
    ops.AUTOTUNE = True
elif algo:
    ops._algo_cache.clear()
    ops.AUTOTUNE = True
    key_hook = {}
    orig = ops._autotune
    ops._autotune = lambda key, *a, **k: (ops._algo_cache.__setitem__(key, algo) or algo)
for _ in range(10):
    ops.conv2d_raw(x, wf, co, (k, k), s, p, bool(tr), (oh, oh), out=y)
torch.cuda.synchronize()
print("done")
