"""Parity of the plan set the headline bench and scripts/train.py actually run.

bench.py and scripts/train.py switch `ops.AUTOTUNE` on and load the shipped perf database `crdr_amd/hip/tune_gfx950.json`
(per-shape tile configuration / split depth / streaming-1x1 / Winograd choices at bs 16 and bs 8, 256x256 crops -- the analogue of
the reference's `cudnn.benchmark = True`, base_trainer.py:20).  Every other oracle comparison of the suite runs the library's
built-in plans at 64x64, so this file replays EVERY entry of the database at its own shape through the product's launch
wrappers (`ops.conv2d_raw / conv_group / conv_multi`, `ops.conv2d_wgrad_raw / wgrad_group / wgrad_multi / wgrad_split`, with
the deferred weight-gradient reduce and the column-sum partial rows the training step uses) and compares

 (i)  with the built-in plan's result on the same operands: max |tuned - builtin| <= 1e-5 of the output scale (two correct fp32
      summation orders), and
 (ii) with float64 stock torch on a strided sample of >= 4 096 output pixels that includes all four borders and the first / last image
      (weight gradients: a 32 x 32 sample of (out, in) channel pairs, all taps, full reduction).  The float64 products run where the
      operands live (ATen's float64 matmul on the device); for every 8th entry they run a second time entirely on the CPU and the two
      references must agree to 1e-11 (round 4 spent 190 s of the suite's 1 056 s in CPU dgemm here).

The autotuner is not allowed to run here: an entry whose reconstructed key misses the database fails the test.
"""
import ast
import ctypes as C
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "crdr_amd", "hip", "tune_gfx950.json")
BF16 = 16384
BF16X6 = 32768
# measured over all 759 entries of the v371 database (profiles/r4_plan_replay.json): tuned vs built-in <= 5.5e-6 (column sums 5.6e-6),
# vs float64 <= 3.7e-6 (exact fp32) / 5.4e-6 (bf16x3 entries); the gates sit at ~3x those
TOL_PLAN = 1e-5      # tuned plan against the built-in plan (fp32 summation order only)
TOL_F64 = 2e-5       # against float64, exact-fp32 entries
TOL_F64_BF16X3 = 2e-5  # opt-in bf16x3 entries (split-bf16 triples: per-product error 3 * 2^-16, random in sign)
TOL_F64_BF16X6 = 4e-6  # opt-in bf16x6 entries (fp32-equivalent: three exact pieces, six products) on DIRECT kernels: the exact-fp32 direct kernels' own
#                        worst over the round-5 database (profiles/r5_plan_replay.json); entries of that mode whose plan is a Winograd id run the
#                        exact-fp32 Winograd kernels and keep those kernels' gate
RESULTS = {}


def _entries(kind):
    db = json.load(open(DB))
    out = []
    for k, v in db["algos"].items():
        key = ast.literal_eval(k)
        if key[0] == kind:
            out.append((key, int(v)))
    return out


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a HIP device"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.rand(shape, generator=g, device="cuda") * 2 - 1) * scale


class _Replay:
    """AUTOTUNE on with the shipped database and the tuner disarmed / AUTOTUNE off (built-in plan)."""

    def __enter__(self):
        from crdr_amd.hip import ops
        self.ops = ops
        self.keep = (ops.AUTOTUNE, dict(ops._algo_cache), ops._autotune, ops.MATRIX_BF16X3, ops.WGRAD_DEFER)
        self.keep6 = ops.MATRIX_BF16X6
        ops._algo_cache.clear()
        assert ops.load_tune_cache(DB) > 0, "the shipped perf database is not of this library build (signature mismatch)"

        def refuse(key, *a, **k):
            raise AssertionError(f"replay reconstructed a key the shipped database does not hold: {key}")
        ops._autotune = refuse
        ops.WGRAD_DEFER = ops.DeferredWgrad(_dev(), arena_bytes=1 << 30)
        return self

    def __exit__(self, *exc):
        ops = self.ops
        ops.AUTOTUNE, cache, ops._autotune, ops.MATRIX_BF16X3, ops.WGRAD_DEFER = self.keep
        ops.MATRIX_BF16X6 = self.keep6
        ops._algo_cache.clear()
        ops._algo_cache.update(cache)

    def mode(self, tuned: bool, prec: int):
        """prec: 0 exact fp32, 3 / 6 the split-bf16 modes the entry was tuned in"""
        self.ops.AUTOTUNE = tuned
        self.ops.MATRIX_BF16X3, self.ops.MATRIX_BF16X6 = prec == 3, prec == 6


def _sel(n, cap):
    """<= cap indices of range(n): both ends, their neighbours, and an even spread in between"""
    if n <= cap:
        return list(range(n))
    s = {0, 1, n - 2, n - 1}
    m = cap - len(s)
    s |= {2 + int((i + 0.5) * (n - 4) / m) for i in range(m)}
    return sorted(s)


def _sample_pixels(n, oh, ow, work=1):
    """~4 096 output pixels (borders + an even spread, every image where there are few); `work` = float64 multiply-adds per sampled
    pixel: the sample shrinks (never below 1 024) where 4 096 of them would cost more than 2e10 on the CPU (the hoisted convs)"""
    want = max(1024, min(4096, int(2e10 // max(work, 1))))
    side = 32 if want >= 4096 else 16
    ys, xs = _sel(oh, side), _sel(ow, side)
    need = -(-want // (len(ys) * len(xs)))
    ns = _sel(n, max(2, min(n, need)))
    idx = torch.cartesian_prod(torch.tensor(ns), torch.tensor(ys), torch.tensor(xs))
    return idx[:, 0], idx[:, 1], idx[:, 2]


def _conv_ref64(xb, wp, geo, pix, on_cpu=False):
    """float64 accumulator of the convolution at the sampled output pixels.  xb: [N,H,W,ld] float32; wp: pack [T][wrows][wcols]
    (wlayout 0) or [1][wrows][4T..] (wlayout 1), on any device: the sampled input rows are gathered, converted to float64 and
    multiplied by stock torch WHERE THE TENSORS LIVE (float64 matmul of ATen: an implementation that shares nothing with the kernels
    under test; on the GPU box 140 grouped entries took 118 s of CPU dgemm, round 4) -- `on_cpu` forces every product and sum onto
    the CPU instead (the replay does that for every 8th entry and requires the two float64 results to agree to 1e-11);
    -> [P][OC] float64 on the CPU"""
    n, h, w, c, oh, ow, oc, kh, kw, stride, pad, transposed, wlayout = geo
    pn, py, px = pix
    dev = torch.device("cpu") if on_cpu else xb.device
    acc = torch.zeros((pn.numel(), oc), dtype=torch.float64, device=dev)
    x2d = xb.reshape(n * h * w, xb.shape[-1])
    for r in range(kh):
        for s in range(kw):
            if transposed:   # out[i*stride - pad + r] += in[i] * w[r]
                ny, nx = py + pad - r, px + pad - s
                ok = (ny % stride == 0) & (nx % stride == 0)
                iy, ix = torch.div(ny, stride, rounding_mode="floor"), torch.div(nx, stride, rounding_mode="floor")
            else:            # out[o] = sum_r in[o*stride - pad + r] * w[r]
                iy, ix = py * stride - pad + r, px * stride - pad + s
                ok = torch.ones_like(iy, dtype=torch.bool)
            ok = ok & (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w)
            if not ok.any():
                continue
            t = r * kw + s
            rows = ((pn[ok] * h + iy[ok]) * w + ix[ok]).to(x2d.device)
            patch = x2d.index_select(0, rows)[:, :c].to(dev).double()          # [P'][c]
            if wlayout == 1:
                wt = wp[0, :oc, 4 * t:4 * t + c].to(dev).double()              # [oc][c]
            else:
                wt = wp[t, :oc, :c].to(dev).double()
            acc[ok.to(dev)] += patch @ wt.t()
    return acc.cpu()


def _epilogue64(v, flags, o, pix, oc):
    """the fused epilogue (include/crdr_hip.h CRDR_EPI_*, in the kernel's order) in float64 on the sampled pixels.
    o: dict of CPU operands; -> (value written, sigmoid or None)"""
    from crdr_amd.hip import lib as L
    pn, py, px = pix

    def at(name):   # (gathered on the operand's device, then float64 on the CPU)
        t = o[name]
        rows = ((pn * t.shape[1] + py) * t.shape[2] + px).to(t.device)
        return t.reshape(-1, t.shape[-1]).index_select(0, rows)[:, :oc].cpu().double()

    def vec(name):
        return o[name][:oc].cpu().double().view(1, -1)
    sig = None
    if flags & L.EPI_PREADD:
        v = v + at("pre")
    if flags & L.EPI_BIAS:
        v = v + vec("bias")
    if flags & L.EPI_RELU:
        v = v.clamp_min(0)
    if flags & L.EPI_LRELU:
        v = torch.where(v > 0, v, 0.2 * v)
    if flags & L.EPI_VEC2:
        v = v + vec("vec2")
    if flags & L.EPI_RES:
        v = v + at("res")
    if flags & L.EPI_GATE:
        sig = torch.sigmoid(v)
        v = at("gx") + at("gt") * sig
    if flags & L.EPI_AFFINE:
        v = v * vec("scale") + vec("shift")
    if flags & (L.EPI_RELUMASK | L.EPI_LRELUMASK):
        mv = at("mask") - (vec("vec2") if flags & L.EPI_MASKOFF else 0.0)
        v = torch.where(mv > 0, v, 0.2 * v if flags & L.EPI_LRELUMASK else torch.zeros_like(v))
    if flags & L.EPI_ACCUM:
        v = v + at("y0")
    return v, sig


def _nchw(buf, c):
    """[N,H,W,ld] buffer -> logical NCHW tensor of its first c channels (memory stays NHWC)"""
    return buf.permute(0, 3, 1, 2)[:, :c]


def _replay_conv(rp, key, algo, seed):
    """-> dict of measured errors for one database entry of kind c / g / m"""
    from crdr_amd.hip import lib as L, ops
    dev = _dev()
    kind = key[0]
    if kind == "c":
        _, n, h, w, c, oh, ow, oc, k, stride, pad, tr, ldx, ldy, flags, ldres, ldg, wlayout = key
        G, ldpre, ldmask = 1, 0, 0
        wrows = ops.round32(oc)
        wcols = ops.round32(4 * k[0] * k[1]) if wlayout else ops.round32(c)
    elif kind == "g":
        _, G, n, h, w, c, oc, k, pad, tr, ldx, ldy, flags, ldpre, ldmask, wrows, wcols = key
        oh, ow, stride, wlayout, ldres, ldg = h, w, 1, 0, 0, 0
    else:
        _, G, n, h, w, oh, ow, c, oc, k, stride, pad, tr, ldx, ldy, flags, ldres, ldpre, ldmask, wrows, wcols, wlayout = key
        ldg = 0
    bf = 3 if (flags & BF16) else (6 if (flags & BF16X6) else 0)
    f = flags & ~(BF16 | BF16X6)
    self_res = kind == "g" and (f & L.EPI_RES)      # conv_group: pure accumulation = residual epilogue with the output as its operand
    T = k[0] * k[1]
    fan = c * T / (stride * stride if tr else 1)
    xs = [_rand((n, h, w, ldx), seed + 10 * g + 1) for g in range(G)]
    wps = []
    for g in range(G):
        wp = _rand((1 if wlayout else T, wrows, wcols), seed + 10 * g + 2, fan ** -0.5)
        wp[:, oc:] = 0
        wp[:, :, (4 * T if wlayout else c):] = 0
        wps.append(wp)
    y0 = [_rand((n, oh, ow, ldy), seed + 10 * g + 3) for g in range(G)]
    opn = {}
    if f & L.EPI_BIAS:
        opn["bias"] = [_rand((oc,), seed + 10 * g + 4) for g in range(G)]
    if f & (L.EPI_VEC2 | L.EPI_MASKOFF):
        opn["vec2"] = [_rand((oc,), seed + 5, 0.3)]
    if f & L.EPI_AFFINE:
        opn["scale"], opn["shift"] = [_rand((oc,), seed + 6) + 1.5], [_rand((oc,), seed + 7)]
    alias_pre = bool(f & L.EPI_PREADD) and ldpre == ldy
    if (f & L.EPI_PREADD) and not alias_pre:
        opn["pre"] = [_rand((n, oh, ow, ldpre), seed + 10 * g + 8) for g in range(G)]
    if (f & L.EPI_RES) and not self_res:
        opn["res"] = [_rand((n, oh, ow, ldres), seed + 10 * g + 9) for g in range(G)]
    if f & (L.EPI_RELUMASK | L.EPI_LRELUMASK):
        opn["mask"] = [_rand((n, oh, ow, ldmask), seed + 10 * g + 11) for g in range(G)]
    if f & L.EPI_GATE:
        opn["gx"], opn["gt"] = [_rand((n, oh, ow, ldg), seed + 12)], [_rand((n, oh, ow, ldg), seed + 13)]

    def launch(tuned):
        rp.mode(tuned, bf)
        ys = [t.clone() for t in y0]
        sig, cs = None, None
        if kind == "c":
            kw = {}
            if "res" in opn:
                kw["res"] = _nchw(opn["res"][0], oc)
            if "scale" in opn:
                kw["scale"], kw["shift"] = opn["scale"][0], opn["shift"][0]
            if f & L.EPI_GATE:
                sig = torch.zeros((n, oh, ow, ldg), device=dev)
                kw.update(gate_x=_nchw(opn["gx"][0], oc), gate_t=_nchw(opn["gt"][0], oc), sig_out=sig)
            ops.conv2d_raw(_nchw(xs[0], c), wps[0], oc, k, stride, pad, bool(tr), (oh, ow), bias=opn["bias"][0] if "bias" in opn else None,
                           vec2=opn["vec2"][0] if "vec2" in opn else None, flags=f, out=_nchw(ys[0], oc), wlayout=wlayout, **kw)
        else:
            xv = [ops.view(t, 0, c) for t in xs]
            yv = [ops.view(t, 0, oc) for t in ys]
            biases = [t.data_ptr() for t in opn["bias"]] if "bias" in opn else None
            pres = None
            if f & L.EPI_PREADD:
                pres = yv if alias_pre else [ops.view(t, 0, oc) for t in opn["pre"]]
            masks = [ops.view(t, 0, oc) for t in opn["mask"]] if "mask" in opn else None
            wa = [t.data_ptr() for t in wps]
            if kind == "g":
                ops.conv_group(n, h, w, xv, wa, yv, oc, k, pad, bool(tr), wrows=wrows, wcols=wcols, biases=biases, pres=pres,
                               masks=masks, flags=(L.EPI_ACCUM if self_res else f & (L.EPI_RELU | L.EPI_ACCUM)), device=dev)
            else:
                q = ops.colsum_queue(dev)
                r = ops.conv_multi(n, h, w, oh, ow, xv, wa, yv, oc, k, stride, pad, bool(tr), wrows=wrows, wcols=wcols, biases=biases,
                                   pres=pres, masks=masks, ress=[ops.view(t, 0, oc) for t in opn["res"]] if "res" in opn else None,
                                   vec2=opn["vec2"][0].data_ptr() if "vec2" in opn else None,
                                   scale=opn["scale"][0].data_ptr() if "scale" in opn else None,
                                   shift=opn["shift"][0].data_ptr() if "shift" in opn else None,
                                   flags=f & ~(L.EPI_BIAS | L.EPI_PREADD | L.EPI_RES | L.EPI_AFFINE | L.EPI_COLSUM),
                                   colsum=bool(f & L.EPI_COLSUM), wlayout=wlayout, device=dev)
                if r is not None:   # partial column-sum rows live in the queue's arena: add them up (what colsum_finish does)
                    cs = []
                    for ptr, rows, ld in r:
                        off = ptr - q.arena.data_ptr()
                        part = q.arena[off:off + rows * 2 * ld * 4].view(torch.float32).view(rows, 2, ld)
                        cs.append(part.double().sum(0)[:, :oc].clone())
                    q.off, q.jobs, q.scratch_off = 0, [], 0
        torch.cuda.synchronize()
        return ys, sig, cs

    yb, sigb, csb = launch(False)
    yt, sigt, cst = launch(True)
    out = {}
    scale = max(float(t[..., :oc].abs().max()) for t in yb) + 1e-20
    out["vs_builtin"] = max(float((a[..., :oc] - b[..., :oc]).abs().max()) for a, b in zip(yt, yb)) / scale
    oc4 = (oc + 3) // 4 * 4
    for a, b, z in zip(yt, yb, y0):   # nothing beyond the output channels (and their 16-byte lane padding) may be touched by either plan
        assert torch.equal(a[..., oc4:], z[..., oc4:]) and torch.equal(b[..., oc4:], z[..., oc4:]), f"{key}: write outside the output channels"
    if sigb is not None:
        out["sig_vs_builtin"] = float((sigt - sigb).abs().max())
    if csb is not None:
        cscale = max(float(t.abs().max()) for t in csb) + 1e-20
        out["colsum_vs_builtin"] = max(float((a - b).abs().max()) for a, b in zip(cst, csb)) / cscale
    # float64 on the CPU: first and last problem of the group, sampled pixels
    pix = _sample_pixels(n, oh, ow, work=c * oc * k[0] * k[1])
    gpix = [t.to(dev) for t in pix]
    geo = (n, h, w, c, oh, ow, oc, k[0], k[1], stride, pad, tr, wlayout)
    e64 = e64b = 0.0
    for g in sorted({0, G - 1}):
        acc = _conv_ref64(xs[g], wps[g], geo, pix)
        if seed % 8000 == 0 and g == 0:   # every 8th entry: the same reference with every product and sum on the CPU
            acc_cpu = _conv_ref64(xs[g], wps[g], geo, pix, on_cpu=True)
            dd = float((acc - acc_cpu).abs().max()) / (float(acc_cpu.abs().max()) + 1e-300)
            assert dd <= 1e-11, f"{key}: float64 references on the device and on the CPU disagree by {dd:.3e}"
            out["f64_device_vs_cpu"] = dd
        o = {"y0": y0[g]}
        for name, lst in opn.items():
            o[name] = lst[g if len(lst) > 1 else 0]
        if alias_pre:
            o["pre"] = o["y0"]
        if self_res:
            o["res"] = o["y0"]
        ref, sref = _epilogue64(acc, f, o, pix, oc)
        rs = float(ref.abs().max()) + 1e-20
        got = yt[g][gpix[0], gpix[1], gpix[2], :oc].cpu().double()
        gotb = yb[g][gpix[0], gpix[1], gpix[2], :oc].cpu().double()
        e64 = max(e64, float((got - ref).abs().max()) / rs)
        e64b = max(e64b, float((gotb - ref).abs().max()) / rs)
        if sref is not None:
            out["sig_vs_f64"] = float((sigt[gpix[0], gpix[1], gpix[2], :oc].cpu().double() - sref).abs().max())
    out["vs_f64"], out["builtin_vs_f64"], out["pixels"] = e64, e64b, int(pix[0].numel())
    out["bf16x3"], out["bf16x6"] = bf == 3, bf == 6
    return out


def _wgrad_ref64(pb, qb, geo, isel, jsel, on_cpu=False):
    """g[i][j][t] = sum P[n,a,b,i] Q[n, a*stride - pad + r, b*stride - pad + s, j] in float64 for i in isel, j in jsel (stock torch float64 where
    the operands live; `on_cpu`: everything on the CPU -- see _conv_ref64)"""
    n, ph, pw, qh, qw, kh, kw, stride, pad = geo
    dev = torch.device("cpu") if on_cpu else pb.device
    P = pb.index_select(-1, torch.tensor(isel, device=pb.device)).to(dev).double().reshape(-1, len(isel))   # [M][I'] (channel gather on the operand's device)
    Q = qb.index_select(-1, torch.tensor(jsel, device=qb.device)).to(dev).double()
    out = torch.zeros((len(isel), len(jsel), kh * kw), dtype=torch.float64, device=dev)
    ext = pad + kh + kw + stride   # zero border wide enough for every tap of every dense pixel
    Qp = torch.nn.functional.pad(Q, (0, 0, ext, ext, ext, ext))
    for r in range(kh):
        for s in range(kw):
            y0, x0 = ext - pad + r, ext - pad + s
            g = Qp[:, y0:y0 + stride * (ph - 1) + 1:stride, x0:x0 + stride * (pw - 1) + 1:stride]   # Q[n, a*stride - pad + r, b*stride - pad + s]
            out[:, :, r * kw + s] = P.t() @ g.reshape(-1, len(jsel))
    return out.cpu()


def _replay_wgrad(rp, key, algo, seed):
    from crdr_amd.hip import ops
    dev = _dev()
    kind = key[0]
    bf = {1: 3, 2: 6}[key[-1]] if len(key) > {"w": 15, "wg": 13, "wm": 16, "ws": 10}[kind] else 0   # (key suffix: ops._mk)
    if kind == "w":
        _, n, ph, pw, pc, ldp, qh, qw, qc, ldq, k, stride, pad, gi, gj = key[:15]
        G = 1
    elif kind == "wg":
        _, G, n, ph, pw, pc, ldp, qc, ldq, k, pad, gi, gj = key[:13]
        qh, qw, stride = ph, pw, 1
    elif kind == "wm":
        _, G, n, ph, pw, pc, ldp, qh, qw, qc, ldq, k, stride, pad, gi, gj = key[:16]
    else:
        _, n, ph, pw, pc, ldp, qc, ldq, k, pad = key[:10]
        G, qh, qw, stride, gi, gj = 1, ph, pw, 1, pc, qc
    T = k[0] * k[1]
    ps = [_rand((n, ph, pw, ldp), seed + 10 * g + 1) for g in range(G)]
    qs = [_rand((n, qh, qw, ldq), seed + 10 * g + 2) for g in range(G)]
    for t in ps:   # lanes between the operand's channels and the next multiple of 4 are layout padding the product keeps at zero
        t[..., gi:pc] = 0   # (RGB: the 4th lane); channels past pc / qc belong to a wider tensor and stay random: they must not be read
    for t in qs:
        t[..., gj:qc] = 0

    def launch(tuned):
        rp.mode(tuned, bf)
        gs = [torch.zeros((gi, gj, k[0], k[1]), device=dev) for _ in range(G)]
        if kind == "w":
            ops.conv2d_wgrad_raw(_nchw(ps[0], gi), _nchw(qs[0], gj), gs[0], k, stride, pad, accumulate=False)
        elif kind == "wg":
            ops.wgrad_group(n, ph, pw, [ops.view(t, 0, pc) for t in ps], [ops.view(t, 0, qc) for t in qs], [(g.data_ptr(), gj) for g in gs],
                            gi, gj, k, pad, device=dev)
        elif kind == "wm":
            ops.wgrad_multi(n, ph, pw, qh, qw, [ops.view(t, 0, pc) for t in ps], [ops.view(t, 0, qc) for t in qs], [g.data_ptr() for g in gs],
                            gi, gj, k, stride, pad, device=dev)
        else:
            ops.wgrad_split(n, ph, pw, ops.view(ps[0], 0, pc), ops.view(qs[0], 0, qc), [(0, pc, gs[0].data_ptr(), qc)], k, pad, device=dev)
        ops.flush_wgrads(("replay", kind, tuned))
        torch.cuda.synchronize()
        return gs

    gb = launch(False)
    gt = launch(True)
    scale = max(float(t.abs().max()) for t in gb) + 1e-20
    out = {"vs_builtin": max(float((a - b).abs().max()) for a, b in zip(gt, gb)) / scale, "bf16x3": bf == 3, "bf16x6": bf == 6}
    isel, jsel = _sel(gi, 32), _sel(gj, 32)
    e = eb = 0.0
    for g in sorted({0, G - 1}):
        ref = _wgrad_ref64(ps[g], qs[g], (n, ph, pw, qh, qw, k[0], k[1], stride, pad), isel, jsel)
        if seed % 8000 == 0 and g == 0:
            ref_cpu = _wgrad_ref64(ps[g], qs[g], (n, ph, pw, qh, qw, k[0], k[1], stride, pad), isel, jsel, on_cpu=True)
            dd = float((ref - ref_cpu).abs().max()) / (float(ref_cpu.abs().max()) + 1e-300)
            assert dd <= 1e-11, f"{key}: float64 references on the device and on the CPU disagree by {dd:.3e}"
        rs = float(ref.abs().max()) + 1e-20
        pick = lambda t: t.cpu().double().reshape(gi, gj, T)[isel][:, jsel]
        e = max(e, float((pick(gt[g]) - ref).abs().max()) / rs)
        eb = max(eb, float((pick(gb[g]) - ref).abs().max()) / rs)
    out["vs_f64"], out["builtin_vs_f64"] = e, eb
    return out


def _depth(key):
    """length of the fp32 accumulation chain behind one output element of the entry"""
    kind = key[0]
    if kind == "c":
        return key[4] * key[8][0] * key[8][1]
    if kind == "g":
        return key[5] * key[7][0] * key[7][1]
    if kind == "m":
        return key[7] * key[9][0] * key[9][1]
    return key[1 if kind in ("w", "ws") else 2] * key[2 if kind in ("w", "ws") else 3] * key[3 if kind in ("w", "ws") else 4]   # N * PH * PW pixels


def _tolerances(key, r):
    """(tuned vs built-in, vs float64).  Two correct fp32 plans differ by summation order only: the bound grows with the square root
    of the accumulation depth (MFMA chains are sequential in K); measured margins: profiles/r4_plan_replay.json."""
    depth = _depth(key)
    wg = key[0].startswith("w")
    plan = max(TOL_PLAN, (1.5e-7 if not wg else 2.5e-7) * depth ** 0.5)
    f64 = TOL_F64_BF16X3 if r["bf16x3"] else (TOL_F64_BF16X6 if r.get("bf16x6") else TOL_F64)
    if r.get("wino4"):   # the F(4x4, 3x3) / F(3x3, 4x4) Winograd kernels (points 0, +-3/4, +-5/4: tests/test_gpu_wino.py measures 0.4e-6 .. 5.2e-6
        plan, f64 = max(plan, 2e-5), 2e-5   # against float64; round 4's points 0, +-1, +-2 needed 6e-5 here)
    if os.environ.get("CRDR_PLAN_REPLAY_MEASURE") == "1":   # first measurement of a new database: gross errors only
        plan, f64 = 1e-3, 3e-3
    return plan, f64


def _run_kind(kind, replay):
    from crdr_amd.hip import lib as L
    entries = _entries(kind)
    assert entries, f"no '{kind}' entries in the shipped perf database"
    bad, rows = [], []
    nwino = nstream = nsplit = 0
    lib = L.load()
    ncfg, nstr = lib.crdr_conv2d_num_configs(), lib.crdr_conv2d_num_stream_configs()
    with _Replay() as rp:
        for i, (key, algo) in enumerate(entries):
            try:
                r = replay(rp, key, algo, 1000 * (i + 1))
            except (AssertionError, L.CrdrHipError) as e:
                bad.append((key, algo, f"{type(e).__name__}: {e}"))
                continue
            if kind in ("c", "g", "m"):
                nstream += ncfg < (algo & 0xFF) <= ncfg + nstr and algo < 256
                nwino += (algo & 0xFF) > ncfg + nstr   # (the F(4x4) kernel's ids may carry K splits in bits 8..11)
                nsplit += algo >= 256
            r["wino4"] = bool(kind in ("c", "g", "m") and lib.crdr_conv2d_num_wino_configs() > 2 and (algo & 0xFF) == ncfg + 1 + nstr + 2)
            r["wino"] = bool(kind in ("c", "g", "m") and (algo & 0xFF) > ncfg + nstr)
            if kind.startswith("w"):   # weight-gradient entries: the Winograd slab kernels are the last configuration and the id behind it
                nwino += (algo & 0xFF) >= lib.crdr_conv2d_wgrad_num_configs()
                r["wino4"] = (algo & 0xFF) == lib.crdr_conv2d_wgrad_num_configs() + 1   # F(3x3, 4x4): the same transform constants
                r["wino"] = (algo & 0xFF) >= lib.crdr_conv2d_wgrad_num_configs()
            tol_plan, tol64 = _tolerances(key, r)
            rows.append({"key": repr(key), "algo": algo, "depth": _depth(key), **{k: v for k, v in r.items()}})
            fails = [n for n, v, t in (("vs_builtin", r["vs_builtin"], tol_plan), ("vs_f64", r["vs_f64"], tol64),
                                       ("builtin_vs_f64", r["builtin_vs_f64"], tol64),
                                       ("colsum_vs_builtin", r.get("colsum_vs_builtin", 0.0), 5 * tol_plan if not r["wino4"] else (2 * tol_plan if r.get("bf16x6") else tol_plan)),   # (bf16x6 entries on an F(4x4) id: the baseline is the split-bf16 direct kernel, another kernel family)
                                       ("sig_vs_builtin", r.get("sig_vs_builtin", 0.0), 1e-5),
                                       ("sig_vs_f64", r.get("sig_vs_f64", 0.0), 1e-4)) if not v <= t]
            if fails:
                bad.append((key, algo, {n: r.get(n) for n in fails}))
            if i % 16 == 15:
                torch.cuda.empty_cache()

    def worst(name, pred=lambda r: True):
        return max([r[name] for r in rows if name in r and pred(r)] + [0.0])
    RESULTS[kind] = {"entries": len(entries), "failed": len(bad), "streaming_1x1_ids": int(nstream), "winograd_ids": int(nwino),
                     "split_ids": int(nsplit),
                     "bf16x6_entries": sum(1 for r in rows if r.get("bf16x6")),
                     "worst": {"vs_builtin": worst("vs_builtin"), "vs_f64": worst("vs_f64", lambda r: not r["bf16x3"] and not r.get("bf16x6")),
                               "vs_f64_bf16x3": worst("vs_f64", lambda r: r["bf16x3"]),
                               "vs_f64_bf16x6_direct": worst("vs_f64", lambda r: r.get("bf16x6") and not r.get("wino4") and not r.get("wino")),
                               "vs_f64_exact_direct": worst("vs_f64", lambda r: not r["bf16x3"] and not r.get("bf16x6") and not r.get("wino4") and not r.get("wino")),
                               "builtin_vs_f64": worst("builtin_vs_f64", lambda r: not r["bf16x3"] and not r.get("bf16x6")),
                               "builtin_vs_f64_bf16x6": worst("builtin_vs_f64", lambda r: r.get("bf16x6")),
                               "colsum_vs_builtin": worst("colsum_vs_builtin"),
                               "vs_builtin_over_sqrt_depth": max([r["vs_builtin"] / r["depth"] ** 0.5 for r in rows] + [0.0])}}
    path = os.environ.get("CRDR_PLAN_REPLAY_DUMP")
    if path:
        prev = json.load(open(path)) if os.path.exists(path) else {"kinds": {}, "rows": {}}
        prev["what"] = ("replay of every entry of crdr_amd/hip/tune_gfx950.json at its own shape: max-abs error over the output scale of the "
                        "tuned plan against the built-in plan and against float64 (tests/test_gpu_tuned_plans.py)")
        prev["database"] = json.load(open(DB))["signature"]
        prev["kinds"][kind] = RESULTS[kind]
        if os.environ.get("CRDR_PLAN_REPLAY_ROWS") == "1":
            prev["rows"][kind] = rows
        with open(path, "w") as fjs:
            json.dump(prev, fjs, indent=1, sort_keys=True)
    assert not bad, f"{len(bad)} of {len(entries)} '{kind}' entries disagree: {bad[:6]}"


@pytest.mark.parametrize("kind", ["c", "g", "m"])
def test_every_tuned_conv_plan_matches_builtin_and_float64(kind):
    _run_kind(kind, _replay_conv)


@pytest.mark.parametrize("kind", ["w", "wg", "wm", "ws"])
def test_every_tuned_wgrad_plan_matches_builtin_and_float64(kind):
    _run_kind(kind, _replay_wgrad)


def test_autotuner_rejects_a_candidate_that_disagrees():
    """ops._autotune compares every candidate's output with the baseline plan's before it may win (a mis-tiled edge must not ship
    because it is fast): a `run` whose candidate 2 writes a wrong value is never chosen, whatever its time."""
    from crdr_amd.hip import ops
    dev = _dev()
    out = torch.zeros(4096, device=dev)
    slow = torch.zeros(1 << 24, device=dev)

    def run(a):
        if a == 0:
            slow.add_(1.0)          # the baseline is the slowest
        out.fill_(1.0)
        if a == 2:
            out[17] = 1.001         # fastest, but wrong
        if a == 3:
            slow[:1 << 20].add_(1.0)
        return a in (0, 2, 3)
    keep = dict(ops._algo_cache)
    try:
        best = ops._autotune(("test-reject",), 3, 0, run, result=lambda: out)
    finally:
        ops._algo_cache.clear()
        ops._algo_cache.update(keep)
    assert best == 3, best
    assert any(k == ("test-reject",) and a == 2 for k, a, _ in ops.TUNE_REJECTED)
