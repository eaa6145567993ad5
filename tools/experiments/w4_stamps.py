"""Phase times of the F(4x4) kernel's tiles from the diagnostic build (-DW4X_STAMPS): python tools/experiments/w4_stamps.py CIN COUT HW [K]
(CRDR_HIP_LIB must point at the stamps build).  Prints, per tile index of a workgroup, median shader cycles of: set-up (tile start -> K loop
call), prologue (-> first sub-step), loop, next-tile requests, epilogue, and the gap to the next tile's start."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from crdr_amd.hip import lib as L, ops  # noqa: E402

ci, co, hw = (int(v) for v in sys.argv[1:4])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = L.load()
raw = C.CDLL(L.LIB_PATH)
wid = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs() + 2
dev = torch.device("cuda:0")
x = torch.randn(16, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, k, k, device=dev) * (ci * k * k) ** -0.5
b = torch.randn(co, device=dev)
wp = ops.pack_weight(w, transpose=False)
for _ in range(5):
    ops.conv2d_raw(x, wp, co, (k, k), 1, k // 2, False, (hw, hw), bias=b, flags=3, algo=wid)
torch.cuda.synchronize()
buf = np.zeros(256 * 16 * 8, dtype=np.uint64)
assert raw.crdr_w4_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.reshape(256, 16, 8).astype(np.int64)
seg = st[:, 8]
if (seg > 0).any():
    nsub = (ci + 3) // 4 * (4 if k == 5 else 1)
    names = ["top (row 5 pass)", "slots 0-14", "15-26 (+vpass 0)", "27-38 (+vpass 1)", "39-50 (+vpass 2, hpass 0-2)", "51-64 (+hpass 3)", "65 (wait + barrier)", "66-71 (+hpass 4 at 62.. see code)"]
    print("in-loop segments, median cycles per sub-step (second tile of each workgroup, wave 0):")
    for i, nm in enumerate(names):
        print(f"   {nm:34s} {float(np.median(seg[:, i])) / nsub:8.1f}")
    print(f"   {'sum':34s} {float(np.median(seg.sum(1))) / nsub:8.1f}")
ep = st[:, 9]
if (ep[:, 4] > 0).any():
    e = ep[ep[:, 4] > 0]
    med = lambda a: float(np.median(a))
    print(f"epilogue of the second tile: output transform of half 0 (+ next tile's requests) {med(e[:, 1]):.0f}, its element-wise part + 16 stores "
          f"{med(e[:, 2] - e[:, 1]):.0f}, output transform of half 1 {med(e[:, 3] - e[:, 2]):.0f}, its element-wise part + stores {med(e[:, 4] - e[:, 3]):.0f}")
print(f"{ci}->{co} k{k} @{hw}: median shader cycles per phase (wave 0 of each workgroup)")
print(" tile   setup  prologue      loop  next-req  epilogue   to-next")
for t in range(16):
    v = st[:, t]
    ok = v[:, 5] > 0
    if not ok.any():
        break
    v = v[ok]
    nxt = st[ok, t + 1, 0] - v[:, 5] if t + 1 < 16 else np.zeros(len(v))
    nxt = nxt[nxt > 0] if (nxt > 0).any() else np.array([0])
    med = lambda a: float(np.median(a))
    if (v[:, 6] > 0).all():
        print(f"      next-req split: tile+vectors {med(v[:, 6] - v[:, 3]):.0f}, src {med(v[:, 7] - v[:, 6]):.0f}, dma issue {med(v[:, 4] - v[:, 7]):.0f}")
    print(f"{t:5d} {med(v[:, 1] - v[:, 0]):7.0f} {med(v[:, 2] - v[:, 1]):9.0f} {med(v[:, 3] - v[:, 2]):9.0f} {med(v[:, 4] - v[:, 3]):9.0f} "
          f"{med(v[:, 5] - v[:, 4]):9.0f} {med(nxt):9.0f}   (n={len(v)})")
