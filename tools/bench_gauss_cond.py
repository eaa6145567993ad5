"""Device time of crdr_gauss_cond_fwd2 / bwd2 at the step's and the codec's sizes (10 calls per HIP graph replay, HIP events):
python tools/bench_gauss_cond.py"""
import sys

import torch

sys.path.insert(0, ".")
from crdr_amd.hip import functional as HF  # noqa: E402
from crdr_amd.hip import lib as L  # noqa: E402
from crdr_amd.hip import ops  # noqa: E402


def timed(fn, reps=10):
    side = torch.cuda.Stream()
    fn()
    torch.cuda.synchronize()
    ops.reserve_workspace(torch.device("cuda:0"), side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    for name, n, c, h, w, noisy, want_lik, bpe in (("training 16x320x16x16 (Philox, both bit sums)", 16, 320, 16, 16, True, False, 16.0),
                                                    ("codec 1x320x128x86 (y_hat + likelihood)", 1, 320, 128, 86, False, True, 20.0),
                                                    ("training, one slice 16x32x16x16", 16, 32, 16, 16, True, False, 16.0)):
        m = n * h * w
        y, mu, sg = (torch.randn(m, c, device=dev) for _ in range(3))
        sg = sg.abs() + 0.05
        yh, lik = torch.empty(m, c, device=dev), (torch.empty(m, c, device=dev) if want_lik else None)
        bn, bq = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ph = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
        d = L.GcDesc2(N=n, HW=h * w, C=c, ldy=c, ldmu=c, ldsigma=c, ldyhat=c, ldlik=c, Ctot=c, c0=0, scale_bound=0.11, likelihood_bound=1e-9)
        io = L.GcIO(y=y.data_ptr(), mu=mu.data_ptr(), sigma=sg.data_ptr(), philox=ph.data_ptr() if noisy else None, yhat=yh.data_ptr(),
                    lik_quant=None if lik is None else lik.data_ptr(), bits_noisy=bn.data_ptr() if noisy else None, bits_quant=bq.data_ptr())
        t = timed(lambda: HF.gauss_cond_fwd2(d, io, dev))
        nb = bpe * m * c
        print(f"fwd {name}: {t * 1e6:7.1f} us  {nb / t / 1e9:7.1f} GB/s  {nb / t / 8e12:.3f} of 8 TB/s", flush=True)
        if noisy:
            gb = torch.ones(n, device=dev)
            dy, dmu, dsg = (torch.empty(m, c, device=dev) for _ in range(3))
            d2 = L.GcDesc2(N=n, HW=h * w, C=c, ldy=c, ldmu=c, ldsigma=c, ldyhat=c, ldgrad=c, Ctot=c, c0=0, scale_bound=0.11, likelihood_bound=1e-9)
            io2 = L.GcIO(y=y.data_ptr(), mu=mu.data_ptr(), sigma=sg.data_ptr(), philox=ph.data_ptr(), gbits=gb.data_ptr(), dy=dy.data_ptr(),
                         dmu=dmu.data_ptr(), dsigma=dsg.data_ptr())
            t = timed(lambda: L.check(lib.crdr_gauss_cond_bwd2(__import__("ctypes").byref(d2), __import__("ctypes").byref(io2), ops._stream()), "bwd"))
            nb = 24.0 * m * c
            print(f"bwd {name}: {t * 1e6:7.1f} us  {nb / t / 1e9:7.1f} GB/s  {nb / t / 8e12:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
