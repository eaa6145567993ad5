"""Fused execution of the channel-autoregressive ("Charm") context model on the HIP conv kernels.

Reference: src/models/subnet/context_model/minnen20_charm_context_model.py:88-141 (forward), :143-240 (codec).  For
slice i the mean / scale / LRP transforms are conv5(->C1)-ReLU-conv5(->C2)-ReLU-conv3(->slice_ch) over
[hyper-prior half, first min(i, ms) decoded slices(, y_hat_i)].  The reference runs 90 small convs in a 10-deep chain
with concatenations in between; here the same arithmetic is re-scheduled:

 * HOIST.  The first conv of every transform is linear in its input channels, so each part of it is computed by the
   launch that has that part of the input.  The hyper-prior part (320 of 320..512 input channels = 72 % of the first-layer
   MACs, 56 % of the whole context model) does not depend on any decoded slice: it is computed for all transforms up front by
   two wide convs (320 -> 19 x 224 from the mean half, 320 -> 9 x 224 from the scale half; slice 0's own transforms have
   no other input and run directly).  The SUPPORT part is scattered the same way (round 3): as soon as slice k is decoded,
   ONE wide conv per half (32 -> 2 (S-1-k) x 224 and 32 -> (S-1-k) x 224, CRDR_EPI_ACCUM) adds its contribution to the
   first-layer pre-activation of EVERY later transform -- 2 x ms efficient launches with N in the thousands instead of a
   small conv per transform and stage whose K grows with the slice index.  A pre-activation that has received all its parts
   is finished by crdr_bias_relu_slots (bias + ReLU in place); the LRP transforms get their last part -- the slice's own
   pre-correction latent -- from a small conv that carries bias + ReLU in its epilogue (CRDR_EPI_PREADD).
 * TAIL.  From slice ms on the support stops growing (`y_hat_slice_list[:5]`, :104-105): the mean / scale transforms of
   ALL remaining slices are independent of each other and run as grouped launches (crdr_conv2d_grouped), then one
   Gaussian-conditional launch over all tail channels, then the LRP transforms as one group.  The sequential depth drops
   from 10 slices to ms + 1.
 * NO CONCATENATION.  Every activation lives in a wide NHWC buffer and is addressed by (channel offset, pixel stride):
   A1 / A2 hold the first / second layer outputs of all 30 transforms ("slots"), MSL holds mu | sigma | lrp, Yh / Ypre
   the decoded latent after / before the LRP correction.
 * BACKWARD is written out by hand in reverse schedule order (no autograd graph inside): ReLU masks ride in the
   input-gradient convs' epilogues (CRDR_EPI_RELUMASK); the gradient of decoded slice k collects the first-layer gradients
   of all its consumers through ONE K-concatenated input-gradient conv per half (CRDR_EPI_ACCUM into dY) and their weight
   gradients through ONE slab launch per half whose rows the batched reduce scatters to the parameters
   (crdr_wgrad_job.gJtot) -- the same two launches per half serve the hoisted hyper-prior parts -- and all bias gradients
   come from three column-sum passes over the wide gradient buffers (crdr_colsum_scatter).

Slot order (A1, A2 and their gradients): [mean_S-1, lrp_S-1, mean_S-2, lrp_S-2, ..., mean_1, lrp_1, lrp_0] (reads the mean
half) [scale_S-1 ... scale_1] (reads the scale half) [mean_0, scale_0]: descending slice index, so the consumers of decoded
slice k -- every transform of a later slice -- are a PREFIX of each half.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import functional as HF
from . import lib as L
from . import ops
from .ops import V


def _chunks(n: int, size: int = L.MAX_GROUP):
    return [(a, min(a + size, n)) for a in range(0, n, size)]


class CharmPlan:
    """Static schedule + persistent weight packs of one Minnen20CharmContextModel instance on one device."""

    def __init__(self, module):
        self.S = module.num_slices
        self.sc = module.slice_ch
        ms = module.max_support_slices
        self.ms = self.S if ms < 0 else ms
        if self.S - self.ms < 2:          # a one-slice tail is just another sequential slice
            self.ms = self.S
        S, ms = self.S, self.ms
        T = S - ms
        self.T = T
        self.tr = {"mean": module.mean_slice_transforms, "scale": module.scale_slice_transforms, "lrp": module.lrp_slice_transforms}
        w1 = self.conv("mean", 0, 0).weight
        self.C1, self.hm = w1.shape[0], w1.shape[1]
        self.C2 = self.conv("mean", 0, 1).weight.shape[0]
        self.k1, self.k2, self.k3 = (tuple(self.conv("mean", 0, l).weight.shape[2:]) for l in range(3))
        assert self.C1 % 32 == 0 and self.C2 % 32 == 0 and self.sc % 32 == 0 and self.hm % 32 == 0, \
            "the fused Charm engine needs channel counts that are multiples of 32"
        self.device = w1.device
        # ---- slots
        order: List[Tuple[str, int]] = []
        for j in range(S - 1, 0, -1):
            order += [("mean", j), ("lrp", j)]
        order += [("lrp", 0)]
        self.n_mu = len(order)
        order += [("scale", j) for j in range(S - 1, 0, -1)]
        self.n_sc = len(order) - self.n_mu
        order += [("mean", 0), ("scale", 0)]
        self.order = order
        self.slot = {t: k for k, t in enumerate(order)}
        self.NT = len(order)
        self.signature = self._signature()
        self._fwd = None
        self._bwd = None
        self._tables = None

    # ---- parameters
    def conv(self, kind: str, i: int, layer: int):
        return getattr(self.tr[kind][i].model, ("0", "2", "4")[layer])

    def params(self):
        for kind in ("mean", "scale", "lrp"):
            for i in range(self.S):
                for l in range(3):
                    c = self.conv(kind, i, l)
                    yield c.weight
                    yield c.bias

    def _signature(self):
        return tuple(p.data_ptr() for p in self.params())

    def sup(self, i: int) -> int:
        """support channels of slice i"""
        return self.sc * min(i, self.ms)

    def ncons(self, k: int) -> int:
        """later slices whose transforms read decoded slice k (0 for k >= ms: not a support slice)"""
        return self.S - 1 - k if k < self.ms else 0

    def consumers(self, k: int, half: str):
        """(first slot, [(kind, j)...]) of the transforms that read slice k through the `half` ("mu" / "sc") of A1: a prefix"""
        nc = self.ncons(k)
        lo, cnt = (0, 2 * nc) if half == "mu" else (self.n_mu, nc)
        return lo, self.order[lo:lo + cnt]

    # ---- packs
    def _alloc(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def fwd_packs(self):
        """Forward operands: packs[(kind, i, what)] = address; entries to keep fresh."""
        if self._fwd is not None:
            return self._fwd
        S, ms, C1, C2, hm, sc = self.S, self.ms, self.C1, self.C2, self.hm, self.sc
        t1, t2, t3 = self.k1[0] * self.k1[1], self.k2[0] * self.k2[1], self.k3[0] * self.k3[1]
        ents, addr, keep = [], {}, []
        # hoisted hyper parts: [t1][n * C1][hm]
        for name, lo, n in (("hyp_mu", 0, self.n_mu), ("hyp_sc", self.n_mu, self.n_sc)):
            buf = self._alloc(t1, n * C1, hm)
            keep.append(buf)
            addr[name] = buf.data_ptr()
            for k in range(n):
                kind, i = self.order[lo + k]
                ents.append(HF.sub_pack(self.conv(kind, i, 0).weight, 0, hm, buf, k * C1 * hm, C1, hm, False, dld=hm, tstride=n * C1 * hm))
        # support parts: decoded slice k -> every later transform, N-concatenated in slot order: [t1][n * C1][sc]
        for k in range(ms):
            for half in ("mu", "sc"):
                lo, cons = self.consumers(k, half)
                if not cons:
                    continue
                n = len(cons)
                buf = self._alloc(t1, n * C1, sc)
                keep.append(buf)
                addr[("sup_" + half, k)] = buf.data_ptr()
                for idx, (kind, j) in enumerate(cons):
                    ents.append(HF.sub_pack(self.conv(kind, j, 0).weight, hm + k * sc, hm + (k + 1) * sc, buf, idx * C1 * sc, C1, sc, False,
                                            dld=sc, tstride=n * C1 * sc))
        for kind in ("mean", "scale", "lrp"):
            for i in range(S):
                w = self.conv(kind, i, 0).weight
                if i == 0 and kind != "lrp":      # slice 0: the whole first conv directly
                    b = self._alloc(t1, C1, hm)
                    ents.append(HF.sub_pack(w, 0, hm, b, 0, C1, hm, False))
                    addr[(kind, 0, "l1")] = b.data_ptr()
                    keep.append(b)
                if kind == "lrp":                 # the slice's own pre-correction latent: the last part of its first conv
                    s = self.sup(i)
                    b = self._alloc(t1, C1, sc)
                    ents.append(HF.sub_pack(w, hm + s, hm + s + sc, b, 0, C1, sc, False))
                    addr[(kind, i, "own")] = b.data_ptr()
                    keep.append(b)
                w2, w3 = self.conv(kind, i, 1).weight, self.conv(kind, i, 2).weight
                b2, b3 = self._alloc(t2, C2, C1), self._alloc(t3, sc, C2)
                ents.append(HF.sub_pack(w2, 0, C1, b2, 0, C2, C1, False))
                ents.append(HF.sub_pack(w3, 0, C2, b3, 0, sc, C2, False))
                addr[(kind, i, "l2")], addr[(kind, i, "l3")] = b2.data_ptr(), b3.data_ptr()
                keep += [b2, b3]
        self._fwd = (addr, ents, keep)
        return self._fwd

    def bwd_packs(self):
        """Input-gradient operands ([tap][in][out])."""
        if self._bwd is not None:
            return self._bwd
        S, ms, C1, C2, hm, sc, T = self.S, self.ms, self.C1, self.C2, self.hm, self.sc, self.T
        t1, t2, t3 = self.k1[0] * self.k1[1], self.k2[0] * self.k2[1], self.k3[0] * self.k3[1]
        ents, addr, keep = [], {}, []
        for name, lo, n in (("hyp_mu", 0, self.n_mu), ("hyp_sc", self.n_mu, self.n_sc)):
            buf = self._alloc(t1, hm, n * C1)
            keep.append(buf)
            addr[name] = buf.data_ptr()
            for k in range(n):
                kind, i = self.order[lo + k]
                ents.append(HF.sub_pack(self.conv(kind, i, 0).weight, 0, hm, buf, k * C1, hm, C1, True, dld=n * C1, tstride=hm * n * C1))
        # support parts, K-concatenated over the consumers of slice k: [t1][sc][n * C1]
        for k in range(ms):
            for half in ("mu", "sc"):
                lo, cons = self.consumers(k, half)
                if not cons:
                    continue
                n = len(cons)
                buf = self._alloc(t1, sc, n * C1)
                keep.append(buf)
                addr[("sup_" + half, k)] = buf.data_ptr()
                for idx, (kind, j) in enumerate(cons):
                    ents.append(HF.sub_pack(self.conv(kind, j, 0).weight, hm + k * sc, hm + (k + 1) * sc, buf, idx * C1, sc, C1, True,
                                            dld=n * C1, tstride=sc * n * C1))
        for kind in ("mean", "scale", "lrp"):
            for i in range(S):
                w = self.conv(kind, i, 0).weight
                if i == 0 and kind != "lrp":
                    b = self._alloc(t1, hm, C1)
                    ents.append(HF.sub_pack(w, 0, hm, b, 0, hm, C1, True))
                    addr[(kind, 0, "l1")] = b.data_ptr()
                    keep.append(b)
                if kind == "lrp":
                    s = self.sup(i)
                    b = self._alloc(t1, sc, C1)
                    ents.append(HF.sub_pack(w, hm + s, hm + s + sc, b, 0, sc, C1, True))
                    addr[(kind, i, "own")] = b.data_ptr()
                    keep.append(b)
                w2, w3 = self.conv(kind, i, 1).weight, self.conv(kind, i, 2).weight
                b2, b3 = self._alloc(t2, C1, C2), self._alloc(t3, C2, sc)
                ents.append(HF.sub_pack(w2, 0, C1, b2, 0, C1, C2, True))
                ents.append(HF.sub_pack(w3, 0, C2, b3, 0, C2, sc, True))
                addr[(kind, i, "l2")], addr[(kind, i, "l3")] = b2.data_ptr(), b3.data_ptr()
                keep += [b2, b3]
        self._bwd = (addr, ents, keep)
        return self._bwd

    def bias_tables(self):
        """Device pointer tables for crdr_colsum_scatter: L1 / L2 biases in slot order, L3 biases in MSL order."""
        g = HF._grad_slot
        sig = (tuple(g(self.conv(k, i, 0).bias).data_ptr() for (k, i) in self.order),
               tuple(g(self.conv(k, i, 1).bias).data_ptr() for (k, i) in self.order),
               tuple(g(self.conv(k, i, 2).bias).data_ptr() for k in ("mean", "scale", "lrp") for i in range(self.S)))
        if self._tables is None or self._tables[3] != sig:  # (the flat gradient buffers never move: built once in training)
            if torch.cuda.is_current_stream_capturing():
                raise L.CrdrHipError("Charm: bias gradient slots moved during graph capture")
            mk = lambda t: torch.tensor(t, dtype=torch.int64, device=self.device)
            self._tables = (mk(sig[0]), mk(sig[1]), mk(sig[2]), sig)
        return self._tables[:3]


def plan_for(module) -> CharmPlan:
    p = getattr(module, "_charm_plan", None)
    if p is None or p.signature != p._signature():
        if p is not None and torch.cuda.is_current_stream_capturing():
            raise L.CrdrHipError("Charm: the parameters moved during graph capture")
        p = CharmPlan(module)
        object.__setattr__(module, "_charm_plan", p)
    return p


class CharmRun:
    """One pass over a batch: owns the wide buffers; `forward()` runs the whole schedule, the stage methods let the
    decoder interleave the host entropy coder."""

    def __init__(self, plan: CharmPlan, hyper_mu: torch.Tensor, hyper_sc: Optional[torch.Tensor]):
        """hyper_mu / hyper_sc: the two halves of the hyper-decoder output (channel slices of one tensor are fine);
        hyper_sc = None: reconstruction only (no scale transforms, no likelihoods)."""
        self.p = plan
        hyper_mu, ldm = ops.nhwc(hyper_mu)
        n, c, h, w = hyper_mu.shape
        assert c == plan.hm
        self.n, self.h, self.w = n, h, w
        self.M = n * h * w
        self.dev = hyper_mu.device
        self.h_mu = V(hyper_mu.data_ptr(), ldm, plan.hm)
        self.with_scale = hyper_sc is not None
        self.h_sc = None
        if hyper_sc is not None:
            hyper_sc, lds = ops.nhwc(hyper_sc)
            assert lds == ldm, "both hyper-prior halves must share one pixel stride"
            self.h_sc = V(hyper_sc.data_ptr(), lds, plan.hm)
        self._hyper = (hyper_mu, hyper_sc)
        P = plan
        self.Cy = P.S * P.sc
        mk = lambda c: torch.empty((self.M, c), dtype=torch.float32, device=self.dev)
        self.A1, self.A2 = mk(P.NT * P.C1), mk(P.NT * P.C2)
        self.MSL = mk(3 * self.Cy)
        self.Yh, self.Ypre = mk(self.Cy), mk(self.Cy)
        self.addr, ents, _ = P.fwd_packs()
        HF.ensure_fresh(ents)
        self._bias = {}

    # ---- views
    def a1(self, slot, n=1):
        return V(self.A1.data_ptr() + 4 * slot * self.p.C1, self.A1.shape[1], n * self.p.C1)

    def a2(self, slot, n=1):
        return V(self.A2.data_ptr() + 4 * slot * self.p.C2, self.A2.shape[1], n * self.p.C2)

    def msl(self, which: int, i: int, n=1):
        return V(self.MSL.data_ptr() + 4 * (which * self.Cy + i * self.p.sc), self.MSL.shape[1], n * self.p.sc)

    def yh(self, c0, c):
        return V(self.Yh.data_ptr() + 4 * c0, self.Cy, c)

    def yp(self, c0, c):
        return V(self.Ypre.data_ptr() + 4 * c0, self.Cy, c)

    def bias(self, kind, i, layer):
        return self.p.conv(kind, i, layer).bias.data_ptr()

    def _conv(self, xs, ws, ys, oc, k, *, wrows, wcols, biases=None, pres=None, relu=False, label="", ms=False, accum=False):
        """ms: a launch over mean (+ scale) transforms.  The mean-only pass (with_scale False) runs it with the plan the full
        pass uses for twice the problems, so that mu -- hence y_hat -- comes out bit-identical in both."""
        mult = 2 if (ms and not self.with_scale) else 1
        for a, b in _chunks(len(xs), L.MAX_GROUP // mult):
            ops.conv_group(self.n, self.h, self.w, xs[a:b], ws[a:b], ys[a:b], oc, k, k[0] // 2, False, wrows=wrows, wcols=wcols,
                           biases=None if biases is None else biases[a:b], pres=None if pres is None else pres[a:b],
                           flags=(L.EPI_RELU if relu else 0) | (L.EPI_ACCUM if accum else 0), device=self.dev, label=label,
                           plan_as=mult * (b - a) if mult > 1 else None)

    # ---- stages
    def stages(self):
        P = self.p
        st = [(i,) for i in range(P.ms)]
        if P.T:
            st.append(tuple(range(P.ms, P.S)))
        return st

    def hoist(self):
        P = self.p
        self._conv([self.h_mu], [self.addr["hyp_mu"]], [self.a1(0, P.n_mu)], P.n_mu * P.C1, P.k1, wrows=P.n_mu * P.C1, wcols=P.hm,
                   label="charm.hoist")
        if self.with_scale:
            self._conv([self.h_sc], [self.addr["hyp_sc"]], [self.a1(P.n_mu, P.n_sc)], P.n_sc * P.C1, P.k1, wrows=P.n_sc * P.C1,
                       wcols=P.hm, label="charm.hoist")

    def _l23(self, trs, outs, label, ms=False):
        """second and third layer of the transforms `trs` -> outs (V into MSL)"""
        P = self.p
        sl = [P.slot[t] for t in trs]
        self._conv([self.a1(s) for s in sl], [self.addr[(k, i, "l2")] for k, i in trs], [self.a2(s) for s in sl], P.C2, P.k2,
                   wrows=P.C2, wcols=P.C1, biases=[self.bias(k, i, 1) for k, i in trs], relu=True, label=label + ".l2", ms=ms)
        self._conv([self.a2(s) for s in sl], [self.addr[(k, i, "l3")] for k, i in trs], outs, P.sc, P.k3,
                   wrows=P.sc, wcols=P.C2, biases=[self.bias(k, i, 2) for k, i in trs], label=label + ".l3", ms=ms)

    def finish(self, trs):
        """bias + ReLU in place on the first-layer pre-activations of `trs`, which have received all their parts"""
        P = self.p
        lib = L.load()
        for a, b in _chunks(len(trs)):
            c0 = (C.c_int32 * (b - a))(*[P.slot[t] * P.C1 for t in trs[a:b]])
            bs = (C.c_void_p * (b - a))(*[self.bias(k, i, 0) for k, i in trs[a:b]])
            L.check(lib.crdr_bias_relu_slots(self.A1.data_ptr(), self.A1.shape[1], self.M, P.C1, b - a, c0, bs, ops._stream()), "bias_relu_slots")

    def mean_scale(self, st):
        """mu (and sigma) of the slices of stage `st` -> MSL"""
        P = self.p
        kinds = ("mean", "scale") if self.with_scale else ("mean",)
        trs = [(k, i) for k in kinds for i in st]
        sl = [P.slot[t] for t in trs]
        if st[0] == 0:
            xs = [self.h_mu, self.h_sc][: len(kinds)]
            self._conv(xs, [self.addr[(k, 0, "l1")] for k in kinds], [self.a1(s) for s in sl], P.C1, P.k1, wrows=P.C1, wcols=P.hm,
                       biases=[self.bias(k, 0, 0) for k in kinds], relu=True, label="charm.ms.l1", ms=True)
        else:       # hoisted hyper-prior part + one part per support slice are in: finish
            self.finish(trs)
        outs = [self.msl(0 if k == "mean" else 1, i) for k, i in trs]
        self._l23(trs, outs, "charm.ms", ms=True)

    def push_support(self, k: int):
        """decoded slice k (final: after its LRP correction) -> the first-layer pre-activations of every later transform"""
        P = self.p
        if not P.ncons(k):
            return
        x = self.yh(k * P.sc, P.sc)
        for half in (("mu", "sc") if self.with_scale else ("mu",)):
            lo, cons = P.consumers(k, half)
            n = len(cons)
            self._conv([x], [self.addr[("sup_" + half, k)]], [self.a1(lo, n)], n * P.C1, P.k1, wrows=n * P.C1, wcols=P.sc, accum=True,
                       label="charm.sup")

    def quantize(self, st, y: Optional[V], noise: Optional[V], philox, lik_n, lik_q, bits_n, bits_q):
        """Gaussian conditional over the channels of stage `st`: Yh = Ypre = round(y - mu) + mu, likelihoods, bit sums."""
        P = self.p
        c0, c = st[0] * P.sc, len(st) * P.sc
        lib = L.load()
        d = L.GcDesc2(N=self.n, HW=self.h * self.w, C=c, ldy=y.ld, ldmu=self.MSL.shape[1], ldsigma=self.MSL.shape[1], ldyhat=self.Cy,
                      ldyhat2=self.Cy, ldnoise=noise.ld if noise is not None else 0, ldlik=self.Cy, ldgrad=0, lddyhat=0,
                      Ctot=self.Cy, c0=c0, scale_bound=self.scale_bound, likelihood_bound=self.lik_bound)
        io = L.GcIO(y=y.ptr + 4 * c0, mu=self.msl(0, st[0]).ptr, sigma=self.msl(1 if self.with_scale else 0, st[0]).ptr,
                    noise=None if noise is None else noise.ptr + 4 * c0, philox=philox,
                    yhat=self.yh(c0, c).ptr, yhat2=self.yp(c0, c).ptr,
                    lik_noisy=None if lik_n is None else lik_n.data_ptr() + 4 * c0,
                    lik_quant=None if lik_q is None else lik_q.data_ptr() + 4 * c0, bits_noisy=bits_n, bits_quant=bits_q)
        HF.gauss_cond_fwd2(d, io, self.dev)

    def lrp(self, st):
        """LRP transforms of stage `st` on the pre-correction latent in Yh, then Yh = Ypre + 0.5 tanh(.); a sequential stage then
        scatters its finished slice to the later transforms"""
        P = self.p
        lt = [("lrp", i) for i in st]
        sl = [self.a1(P.slot[t]) for t in lt]
        self._conv([self.yh(i * P.sc, P.sc) for i in st], [self.addr[("lrp", i, "own")] for i in st], sl, P.C1, P.k1,
                   wrows=P.C1, wcols=P.sc, biases=[self.bias("lrp", i, 0) for i in st], pres=sl, relu=True, label="charm.lrp.l1")
        self._l23(lt, [self.msl(2, i) for i in st], "charm.lrp")
        c0, c = st[0] * P.sc, len(st) * P.sc
        L.check(L.load().crdr_lrp(self.yp(c0, c).ptr, self.Cy, self.msl(2, st[0]).ptr, self.MSL.shape[1], self.yh(c0, c).ptr, self.Cy,
                                  self.M, c, ops._stream()), "lrp")
        if len(st) == 1:
            self.push_support(st[0])

    # ---- whole pass
    def forward(self, y: torch.Tensor, noise: Optional[torch.Tensor], philox: Optional[int], scale_bound: float, lik_bound: float,
                want_lik: bool, want_bits: bool):
        y, ldy = ops.nhwc(y)
        self.y = y
        self.yv = yv = V(y.data_ptr(), ldy, self.Cy)
        self.nv = nv = None
        if noise is not None:
            noise, ldn = ops.nhwc(noise)
            self.noise = noise
            self.nv = nv = V(noise.data_ptr(), ldn, self.Cy)
        self.scale_bound, self.lik_bound = scale_bound, lik_bound
        noisy = noise is not None or philox is not None
        mk = lambda: torch.empty((self.M, self.Cy), dtype=torch.float32, device=self.dev)
        self.lik_n = mk() if (want_lik and noisy) else None
        self.lik_q = mk() if want_lik else None
        self.bits_n = torch.zeros(self.n, dtype=torch.float32, device=self.dev) if (want_bits and noisy) else None
        self.bits_q = torch.zeros(self.n, dtype=torch.float32, device=self.dev) if want_bits else None
        self.hoist()
        for st in self.stages():
            self.mean_scale(st)
            self.quantize(st, yv, nv, philox, self.lik_n, self.lik_q, ops._p(self.bits_n), ops._p(self.bits_q))
            self.lrp(st)

    def as_nchw(self, buf: torch.Tensor, c0: int = 0, c: Optional[int] = None) -> torch.Tensor:
        c = buf.shape[1] - c0 if c is None else c
        return buf.view(self.n, self.h, self.w, buf.shape[1]).permute(0, 3, 1, 2)[:, c0:c0 + c]


# ---------------------------------------------------------------------------------------------------------
# backward
# ---------------------------------------------------------------------------------------------------------
_local_defer: Dict = {}


def charm_backward(run: CharmRun, dyhat: Optional[torch.Tensor], gbits: Optional[torch.Tensor], philox: Optional[int]):
    """-> (dy, dhyper) as NCHW views; parameter gradients are accumulated straight into the (flat) .grad slots."""
    P = run.p
    lib = L.load()
    dev, n, h, w, M, Cy = run.dev, run.n, run.h, run.w, run.M, run.Cy
    S, ms, C1, C2, sc, hm, T = P.S, P.ms, P.C1, P.C2, P.sc, P.hm, P.T
    baddr, bents, _ = P.bwd_packs()
    HF.ensure_fresh(bents)
    mk = lambda c, zero=False: (torch.zeros if zero else torch.empty)((M, c), dtype=torch.float32, device=dev)
    dA1, dA2 = mk(P.NT * C1), mk(P.NT * C2)
    G4 = mk(4 * Cy)            # dMU | dSG | dLR | dy (one pixel stride for the Gaussian-conditional backward)
    dY = mk(Cy)
    dH = mk(2 * hm)
    if dyhat is None:
        dY.zero_()
    else:
        dY.view(n, h, w, Cy).copy_(dyhat.permute(0, 2, 3, 1))
    if gbits is None:
        gbits = torch.zeros(n, dtype=torch.float32, device=dev)
    gbits = gbits.contiguous()
    da1 = lambda slot, k=1: V(dA1.data_ptr() + 4 * slot * C1, dA1.shape[1], k * C1)
    da2 = lambda slot, k=1: V(dA2.data_ptr() + 4 * slot * C2, dA2.shape[1], k * C2)
    g4 = lambda which, i, k=1: V(G4.data_ptr() + 4 * (which * Cy + i * sc), 4 * Cy, k * sc)
    dyv = lambda c0, c: V(dY.data_ptr() + 4 * c0, Cy, c)
    pad1, pad2, pad3 = P.k1[0] // 2, P.k2[0] // 2, P.k3[0] // 2
    t1 = P.k1[0] * P.k1[1]

    def wg(kind, i, layer):
        return HF._grad_slot(P.conv(kind, i, layer).weight)

    def dgrad(xs, ws, ys, oc, k, wrows, wcols, masks=None, accum=False, label=""):
        for a, b in _chunks(len(xs)):
            ops.conv_group(n, h, w, xs[a:b], ws[a:b], ys[a:b], oc, k, k[0] // 2, True, wrows=wrows, wcols=wcols,
                           masks=None if masks is None else masks[a:b], flags=L.EPI_ACCUM if accum else 0, device=dev, label=label)

    def wgrad(ps, qs, gs, gi, gj, k, label=""):
        for a, b in _chunks(len(ps)):
            ops.wgrad_group(n, h, w, ps[a:b], qs[a:b], gs[a:b], gi, gj, k, k[0] // 2, device=dev, label=label)

    prev_defer = ops.WGRAD_DEFER
    local = prev_defer is None or prev_defer.device != dev
    if local:
        d = _local_defer.get(dev)
        if d is None:
            d = _local_defer[dev] = ops.DeferredWgrad(dev, arena_bytes=256 << 20)
        ops.WGRAD_DEFER = d
    try:
        def l32_bwd(trs, douts, label):
            """third + second layer backward of transforms `trs` given the gradients of their outputs (V into G4):
            leaves dz1 (masked) in dA1 slots, queues the weight gradients"""
            sl = [P.slot[t] for t in trs]
            dgrad(douts, [baddr[(k, i, "l3")] for k, i in trs], [da2(s) for s in sl], C2, P.k3, C2, sc,
                  masks=[run.a2(s) for s in sl], label=label + ".l3")
            wgrad(douts, [run.a2(s) for s in sl], [(wg(k, i, 2).data_ptr(), 0) for k, i in trs], sc, C2, P.k3, label=label + ".l3")
            dgrad([da2(s) for s in sl], [baddr[(k, i, "l2")] for k, i in trs], [da1(s) for s in sl], C1, P.k2, C1, C2,
                  masks=[run.a1(s) for s in sl], label=label + ".l2")
            wgrad([da2(s) for s in sl], [run.a1(s) for s in sl], [(wg(k, i, 1).data_ptr(), 0) for k, i in trs], C2, C1, P.k2,
                  label=label + ".l2")

        def gc_bwd(st):
            c0, c = st[0] * sc, len(st) * sc
            d = L.GcDesc2(N=n, HW=h * w, C=c, ldy=run.yv.ld, ldmu=3 * Cy, ldsigma=3 * Cy,
                          ldyhat=0, ldyhat2=0, ldnoise=0 if run.nv is None else run.nv.ld, ldlik=0, ldgrad=4 * Cy, lddyhat=Cy,
                          Ctot=Cy, c0=c0, scale_bound=run.scale_bound, likelihood_bound=run.lik_bound)
            io = L.GcIO(y=run.yv.ptr + 4 * c0, mu=run.msl(0, st[0]).ptr, sigma=run.msl(1, st[0]).ptr,
                        noise=None if run.nv is None else run.nv.ptr + 4 * c0, philox=philox, gbits=gbits.data_ptr(),
                        dyhat=dyv(c0, c).ptr, dy=g4(3, st[0]).ptr, dmu=g4(0, st[0]).ptr, dsigma=g4(1, st[0]).ptr)
            L.check(lib.crdr_gauss_cond_bwd2(C.byref(d), C.byref(io), ops._stream()), "gauss_cond_bwd2")

        for st in reversed(run.stages()):
            c0, c = st[0] * sc, len(st) * sc
            lt = [("lrp", i) for i in st]
            if len(st) == 1 and P.ncons(st[0]):
                # decoded slice k fed the first layer of every later transform: their (masked) first-layer gradients are final by
                # now -- ONE K-concatenated input-gradient conv per half adds them to d Yh_k, ONE slab launch per half yields the
                # weight gradients of those parts (rows scattered to the parameters' support columns by the batched reduce)
                k = st[0]
                for half in ("mu", "sc"):
                    lo, cons = P.consumers(k, half)
                    nc = len(cons)
                    dgrad([da1(lo, nc)], [baddr[("sup_" + half, k)]], [dyv(k * sc, sc)], sc, P.k1, sc, nc * C1, accum=True, label="charm.sup")
                    parts = []
                    for idx, (kind, j) in enumerate(cons):
                        gw = wg(kind, j, 0)
                        parts.append((idx * C1, C1, gw.data_ptr() + 4 * (hm + k * sc) * t1, gw.shape[1]))
                    ops.wgrad_split(n, h, w, da1(lo, nc), run.yh(k * sc, sc), parts, P.k1, pad1, device=dev, label="charm.sup")
            # Yh = Ypre + 0.5 tanh(lrp): d lrp
            L.check(lib.crdr_lrp_bwd(dyv(c0, c).ptr, Cy, run.msl(2, st[0]).ptr, 3 * Cy, g4(2, st[0]).ptr, 4 * Cy, M, c, ops._stream()),
                    "lrp_bwd")
            l32_bwd(lt, [g4(2, i) for i in st], "charm.lrp")
            # the slice's own pre-correction latent: d Ypre_i on top of d Yh_i (Yh = Ypre + ...: the identity path shares dY)
            dgrad([da1(P.slot[t]) for t in lt], [baddr[("lrp", i, "own")] for i in st], [dyv(i * sc, sc) for i in st], sc, P.k1, sc, C1,
                  accum=True, label="charm.lrp.l1own")
            wgrad([da1(P.slot[t]) for t in lt], [run.yp(i * sc, sc) for i in st],
                  [(wg("lrp", i, 0).data_ptr() + 4 * (hm + P.sup(i)) * t1, wg("lrp", i, 0).shape[1]) for i in st], C1, sc, P.k1,
                  label="charm.lrp.l1own")
            gc_bwd(st)
            trs = [(k, i) for k in ("mean", "scale") for i in st]
            l32_bwd(trs, [g4(0 if k == "mean" else 1, i) for k, i in trs], "charm.ms")
            if st[0] == 0:
                dgrad([da1(P.slot[("mean", 0)]), da1(P.slot[("scale", 0)])], [baddr[("mean", 0, "l1")], baddr[("scale", 0, "l1")]],
                      [V(dH.data_ptr(), 2 * hm, hm), V(dH.data_ptr() + 4 * hm, 2 * hm, hm)], hm, P.k1, hm, C1, label="charm.ms.l1")
                wgrad([da1(P.slot[("mean", 0)]), da1(P.slot[("scale", 0)])], [run.h_mu, run.h_sc],
                      [(wg("mean", 0, 0).data_ptr(), 0), (wg("scale", 0, 0).data_ptr(), 0)], C1, hm, P.k1, label="charm.ms.l1")
        # hoisted hyper-prior parts
        for name, lo, cnt, hv, c0 in (("hyp_mu", 0, P.n_mu, run.h_mu, 0), ("hyp_sc", P.n_mu, P.n_sc, run.h_sc, hm)):
            dgrad([da1(lo, cnt)], [baddr[name]], [V(dH.data_ptr() + 4 * c0, 2 * hm, hm)], hm, P.k1, hm, cnt * C1, accum=True,
                  label="charm.hoist")
            parts = []
            for k in range(cnt):
                kind, i = P.order[lo + k]
                gw = wg(kind, i, 0)
                parts.append((k * C1, C1, gw.data_ptr(), gw.shape[1]))
            ops.wgrad_split(n, h, w, da1(lo, cnt), hv, parts, P.k1, pad1, device=dev, label="charm.hoist")
        # bias gradients: three column-sum passes
        tb1, tb2, tb3 = P.bias_tables()
        ops.colsum_scatter(V(dA1.data_ptr(), dA1.shape[1], dA1.shape[1]), M, C1, tb1, dev)
        ops.colsum_scatter(V(dA2.data_ptr(), dA2.shape[1], dA2.shape[1]), M, C2, tb2, dev)
        ops.colsum_scatter(V(G4.data_ptr(), 4 * Cy, 3 * Cy), M, sc, tb3, dev)
        if local:
            ops.WGRAD_DEFER.flush(("charm-local", dev.index))
    finally:
        ops.WGRAD_DEFER = prev_defer
    run._keep_bwd = (dA1, dA2, G4, dY, dH)
    dy = G4.view(n, h, w, 4 * Cy).permute(0, 3, 1, 2)[:, 3 * Cy:]
    dhyper = dH.view(n, h, w, 2 * hm).permute(0, 3, 1, 2)
    return dy, dhyper


class _CharmFn(torch.autograd.Function):
    """(y, hyper_out[, noise]) -> (y_hat, bits_noisy[N], bits_quant[N], lik_noisy?, lik_quant?, mu, sigma)"""

    @staticmethod
    def forward(ctx, y, hyper_out, noise, module, philox_state, scale_bound, lik_bound, want_lik, is_train):
        plan = plan_for(module)
        h_mu, h_sc = torch.chunk(hyper_out, 2, dim=1)
        run = CharmRun(plan, h_mu, h_sc)
        ph = None
        if is_train and noise is None:
            ph_call = torch.empty(2, dtype=torch.int64, device=y.device)
            inc = (y.numel() + 3) // 4 + 1
            L.check(L.load().crdr_philox_fork(philox_state.data_ptr(), ph_call.data_ptr(), inc, ops._stream()), "philox_fork")
            ph = ph_call.data_ptr()
            ctx.ph_call = ph_call
        run.forward(y, noise if is_train else None, ph, scale_bound, lik_bound, want_lik, True)
        ctx.run, ctx.ph = run, ph
        if ops.RELU_MASK_SINK is not None and any(ctx.needs_input_grad):   # (test hook: the activations charm_backward takes its ReLU masks from)
            for (kind, i), slot in plan.slot.items():
                ops.RELU_MASK_SINK(plan.conv(kind, i, 0).weight, run.as_nchw(run.A1, slot * plan.C1, plan.C1), None)
                ops.RELU_MASK_SINK(plan.conv(kind, i, 1).weight, run.as_nchw(run.A2, slot * plan.C2, plan.C2), None)
        yhat = run.as_nchw(run.Yh)
        mu, sigma = run.as_nchw(run.MSL, 0, run.Cy), run.as_nchw(run.MSL, run.Cy, run.Cy)
        lik_n = run.as_nchw(run.lik_n) if run.lik_n is not None else None
        lik_q = run.as_nchw(run.lik_q) if run.lik_q is not None else None
        ctx.mark_non_differentiable(run.bits_q, mu, sigma)
        for t in (lik_n, lik_q):
            if t is not None:
                ctx.mark_non_differentiable(t)
        return yhat, run.bits_n, run.bits_q, lik_n, lik_q, mu, sigma

    @staticmethod
    def backward(ctx, dyhat, dbits_n, *_):
        run = ctx.run
        if run.bits_n is None:
            raise L.CrdrHipError("Charm: backward needs the noisy (training) forward")
        dy, dh = charm_backward(run, dyhat, dbits_n, ctx.ph)
        ctx.run = None
        return dy, dh, None, None, None, None, None, None, None


def charm_forward(module, y, hyper_out, noise, philox_state, scale_bound, lik_bound, want_lik, is_train):
    return _CharmFn.apply(y, hyper_out, noise, module, philox_state, float(scale_bound), float(lik_bound), bool(want_lik),
                          bool(is_train))


@torch.no_grad()
def charm_reconstruct(module, y, hyper_mu, scale_bound, lik_bound) -> CharmRun:
    """y_hat only (the no-grad high-rate pass of stage 3): no scale transforms, no likelihoods.  Returns the run (y_hat =
    run.as_nchw(run.Yh))."""
    run = CharmRun(plan_for(module), hyper_mu, None)
    run.forward(y, None, None, float(scale_bound), float(lik_bound), False, False)
    return run
