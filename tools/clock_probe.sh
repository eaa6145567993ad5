#!/bin/bash
# sample the shader clock / power while a kernel loop runs (is the fp32 MFMA rate clock-limited?)
python3 - <<'PY' &
import sys, torch
sys.path.insert(0, ".")
from crdr_amd.hip import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
wid = lib.crdr_conv2d_num_configs() + 1 + lib.crdr_conv2d_num_stream_configs()
x = torch.randn(16, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(128, 128, 3, 3, device=dev) * 0.03
b = torch.randn(128, device=dev)
wp = ops.pack_weight(wt, transpose=False)
import time
for name, algo in (("direct", 24), ("winograd", wid), ("idle", None)):
    t0 = time.time()
    print("PHASE", name, flush=True)
    while time.time() - t0 < 6:
        if algo is None:
            time.sleep(0.1)
        else:
            for _ in range(50):
                ops.conv2d_raw(x, wp, 128, (3, 3), 1, 1, False, (128, 128), bias=b, flags=3, algo=algo)
            torch.cuda.synchronize()
PY
PID=$!
sleep 12   # (import + first phase start)
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $PID
