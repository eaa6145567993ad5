"""Hand-scheduled forward / backward of conv chains (residual bottleneck stacks, NLAM branches, discriminator trunks).

The reference builds these from nn.Sequential pieces (src/models/layer/elic_layers.py:23-53, cheng_nlam.py:31-46,
elic_interpca_beta_cond_autoencoder.py:42-84, clic21_gvae_discriminator.py:12-50) and lets autograd walk them op by op.
Here a *chain* -- a list of units, each a list of conv layers with an optional residual from the unit's input to its
output -- is ONE autograd node whose backward is written out:

 * the ReLU / LeakyReLU backward of layer l rides in the epilogue of the input-gradient conv of layer l+1
   (CRDR_EPI_RELUMASK / LRELUMASK on the saved activation, CRDR_EPI_MASKOFF when a beta vector was added after the ReLU),
   so no elementwise pass re-reads the activation and its gradient;
 * bias and beta-vector gradients are column sums produced by the same epilogue (CRDR_EPI_COLSUM) and finished by one
   batched launch per chain (ops.ColsumQueue);
 * the residual gradient is the RES operand of the unit's first input-gradient conv: no separate accumulation kernel;
 * G parallel chains of one geometry (the trunk and attention branches of an NLAM) run as grouped launches.
The forward issues exactly the launches the per-layer path issued (same kernels, same packs), so values are unchanged."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import functional as HF
from . import lib as L
from . import ops
from .ops import V


class Layer:
    """One conv of a chain: `conv` is a HipConv2d / HipConvTranspose2d (weight, bias, spec); act in (None, 'relu', 'lrelu');
    vec_index: index into the chain's beta vectors added AFTER the activation (or None)."""
    __slots__ = ("conv", "act", "vec_index")

    def __init__(self, conv, act=None, vec_index=None):
        self.conv, self.act, self.vec_index = conv, act, vec_index


class Unit:
    __slots__ = ("layers", "residual")

    def __init__(self, layers: Sequence[Layer], residual: bool):
        self.layers, self.residual = list(layers), residual


class ChainSpec:
    """Static description: `groups` parallel chains (lists of units) of identical geometry; `affine` = the last conv of
    group 0's last unit carries a per-channel scale + shift epilogue (InterpChAtt); nvec = number of beta vectors."""

    def __init__(self, groups: Sequence[Sequence[Unit]], affine: bool = False, nvec: int = 0, name: str = "chain"):
        self.groups = [list(g) for g in groups]
        self.G = len(self.groups)
        self.U = len(self.groups[0])
        assert all(len(g) == self.U for g in self.groups)
        self.affine, self.nvec, self.name = affine, nvec, name
        assert not (affine and self.G > 1)
        self._dv = None

    def unit(self, g, u) -> Unit:
        return self.groups[g][u]

    def parameters(self):
        out = []
        for grp in self.groups:
            for un in grp:
                for lay in un.layers:
                    out.append(lay.conv.weight)
                    if lay.conv.bias is not None:
                        out.append(lay.conv.bias)
        return out

    def dv_buffers(self, sizes, device):
        """persistent gradient buffers of the beta vectors (stable addresses for the column-sum job tables)"""
        if self._dv is None or [t.numel() for t in self._dv[1]] != list(sizes) or self._dv[0].device != device:
            flat = torch.zeros(sum((s + 3) // 4 * 4 for s in sizes), dtype=torch.float32, device=device)
            vs, off = [], 0
            for s in sizes:
                vs.append(flat[off:off + s])
                off += (s + 3) // 4 * 4
            self._dv = (flat, vs)
        return self._dv


def _geom(conv, h, w):
    s = conv.spec
    oh, ow = s.out_hw(h, w)
    return oh, ow


def _flags(act):
    return L.EPI_RELU if act == "relu" else L.EPI_LRELU if act == "lrelu" else 0


def _bufs(m, c, g, device):
    """g dense [M][c'] buffers (c' = c rounded up to 4, padding lanes zeroed): one per parallel chain, same pixel stride"""
    cp = (c + 3) // 4 * 4
    mk = torch.zeros if cp != c else torch.empty
    return [mk((m, cp), dtype=torch.float32, device=device) for _ in range(g)], cp


def _nchw(buf, n, h, w, c):
    return buf.view(n, h, w, buf.shape[1]).permute(0, 3, 1, 2)[:, :c]


# A forward pass kept for a SECOND backward (round 6).  The stage-3 step evaluates the discriminator on the reconstruction twice with unchanged
# weights: in the generator phase (D frozen, the adversarial term's gradient flows to the image) and in the discriminator phase (image detached,
# the gradient flows to D's weights) -- the reference even does it three times (multirate_hr_rgan_beta_cond_rate_distortion_trainer.py:52, 92,
# 98).  Same input, same weights, same launches: the activations of the first pass ARE those of the second.  With KEEP_HANDLES a list, every
# chain forward appends a handle (spec, saved activations, geometry, input, outputs); run_chain_reuse(handle) is then an autograd node whose
# forward launches nothing and whose backward is _ChainFn.backward on the kept activations.
KEEP_HANDLES = None


class ChainHandle:
    __slots__ = ("spec", "saved", "geom", "x", "oc")


class _KeptCtx:
    """what _ChainFn.backward reads of its ctx, for a backward on kept activations: the chain input counts as detached
    (needs_input_grad[0]), there are no beta vectors ([5 + i])"""
    needs_input_grad = (False,) * 8

    def __init__(self, ctx):
        self.spec, self.saved_acts, self.geom, self.nrest, self.saved_tensors = ctx.spec, ctx.saved_acts, ctx.geom, ctx.nrest, ctx.saved_tensors


class _ChainFn(torch.autograd.Function):
    """(x, spec, scale, shift, nvec, *vecs, *params) -> tuple of G outputs (NCHW views).  The parameters are passed only so
    that autograd sees the node as differentiable when x is detached (discriminator phase); their gradients are
    accumulated straight into the (flat) .grad slots by the kernels, the node returns None for them."""

    @staticmethod
    def forward(ctx, x, spec: ChainSpec, scale, shift, nvec, *rest):
        vecs = rest[:nvec]
        ctx.nrest = len(rest)
        x, ldx = ops.nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        G = spec.G
        xin = [V(x.data_ptr(), ldx, c)] * G
        saved = []   # per unit: list over layers of (buffer tensor, width, (n, h, w) of the OUTPUT)
        cur, cur_hw = xin, (h, w)
        unit_inputs = []
        for u in range(spec.U):
            unit_inputs.append((cur, cur_hw))
            acts = []
            nl = len(spec.unit(0, u).layers)
            for li in range(nl):
                lay0 = spec.unit(0, u).layers[li]
                convs = [spec.unit(g, u).layers[li].conv for g in range(G)]
                sp = convs[0].spec
                ih, iw = cur_hw
                oh, ow = sp.out_hw(ih, iw)
                bufs, cp = _bufs(n * oh * ow, sp.out_ch, G, dev)
                ys = [V(b.data_ptr(), cp, sp.out_ch) for b in bufs]
                last = li == nl - 1
                ress = unit_inputs[u][0] if (last and spec.unit(0, u).residual) else None
                aff = spec.affine and last and u == spec.U - 1
                packs = [cv.spec.pack(cv.weight, False) for cv in convs]
                ops.conv_multi(n, ih, iw, oh, ow, cur, [pk.data_ptr() for pk in packs], ys, sp.out_ch, sp.k, sp.stride, sp.pad,
                               sp.transposed, wrows=packs[0].shape[1], wcols=packs[0].shape[2],
                               biases=[cv.bias.data_ptr() for cv in convs] if convs[0].bias is not None else None,
                               ress=ress, vec2=None if lay0.vec_index is None else vecs[lay0.vec_index].data_ptr(),
                               scale=scale.data_ptr() if aff else None, shift=shift.data_ptr() if aff else None,
                               flags=_flags(lay0.act) | (L.EPI_VEC2 if lay0.vec_index is not None else 0),
                               wlayout=1 if sp.smallc else 0, device=dev, label=spec.name)
                acts.append((bufs, cp, (n, oh, ow)))
                if ops.RELU_MASK_SINK is not None and lay0.act == "relu" and any(ctx.needs_input_grad):   # (a no-grad pass -- the high-rate reconstruction -- has no backward to take masks for)
                    for g in range(G):   # (test hook: the activations this node's backward takes its ReLU masks from)
                        ops.RELU_MASK_SINK(convs[g].weight, _nchw(bufs[g], n, oh, ow, sp.out_ch), None if lay0.vec_index is None else vecs[lay0.vec_index])
                cur, cur_hw = ys, (oh, ow)
            saved.append(acts)
        ctx.spec, ctx.saved_acts, ctx.geom = spec, saved, (n, c, h, w, ldx)
        ctx.save_for_backward(x, scale, shift, *vecs)
        bufs, cp, (n_, oh, ow) = saved[-1][-1]
        oc = spec.unit(0, spec.U - 1).layers[-1].conv.spec.out_ch
        if KEEP_HANDLES is not None and not vecs and scale is None:
            hd = ChainHandle()
            hd.spec, hd.saved, hd.geom, hd.x, hd.oc = spec, saved, (n, c, h, w, ldx), x, oc
            KEEP_HANDLES.append(hd)
        return tuple(_nchw(b, n_, oh, ow, oc) for b in bufs)

    @staticmethod
    def backward(ctx, *douts):
        spec: ChainSpec = ctx.spec
        x, scale, shift, *vecs = ctx.saved_tensors
        n, c, h, w, ldx = ctx.geom
        dev = x.device
        G = spec.G
        saved = ctx.saved_acts
        q = ops.colsum_queue(dev)
        dvs = []
        if vecs:
            flat, dvs = spec.dv_buffers([v.numel() for v in vecs], dev)
            flat.zero_()
        prev_defer = ops.WGRAD_DEFER
        local = prev_defer is None or prev_defer.device != dev
        if local:
            from .charm import _local_defer
            d = _local_defer.get(dev)
            if d is None:
                d = _local_defer[dev] = ops.DeferredWgrad(dev, arena_bytes=256 << 20)
            ops.WGRAD_DEFER = d

        def views(bufs, cp, cc, nn, hh, ww):
            """per-problem (V, NCHW tensor view) of the G dense buffers"""
            return [V(b.data_ptr(), cp, cc) for b in bufs], [_nchw(b, nn, hh, ww, cc) for b in bufs]

        def bias_slot(cv):
            return HF._grad_slot(cv.bias) if (cv.bias is not None and cv.bias.requires_grad) else None
        try:
            # ---- gradient at the chain output(s).  An affine epilogue on the last conv (InterpChAtt) is undone by the
            # fused elementwise pass, which also yields d scale / d shift and the column sums of the pre-affine gradient.
            obufs, cp, (n_, oh, ow) = saved[-1][-1]
            last_layers = [spec.unit(g, spec.U - 1).layers[-1] for g in range(G)]
            oc = last_layers[0].conv.spec.out_ch
            dscale = dshift = None
            if spec.affine:
                out = _nchw(obufs[0], n_, oh, ow, oc)
                lay = last_layers[0]
                _, gres, _, cs = ops.epilogue_bwd(douts[0], out, L.EPI_AFFINE | L.EPI_RES, scale=scale, shift=shift, need_dz=False,
                                                 dbias_accum=bias_slot(lay.conv))
                gres, ldg = ops.nhwc(gres)
                dz_v, dz_t = [V(gres.data_ptr(), ldg, oc)], [gres]
                dscale, dshift = cs[2], cs[3]
                if lay.vec_index is not None:
                    dvs[lay.vec_index].add_(cs[0])
            else:
                ts = [douts[g] if douts[g] is not None else torch.zeros_like(_nchw(obufs[g], n_, oh, ow, oc)) for g in range(G)]
                pairs = [ops.nhwc(t) for t in ts]
                if len({ld for _, ld in pairs}) > 1:   # a grouped launch needs one pixel stride: repack densely
                    wbs, cpp = _bufs(n_ * oh * ow, oc, G, dev)
                    for g in range(G):
                        _nchw(wbs[g], n_, oh, ow, oc).copy_(ts[g])
                    dz_v, dz_t = views(wbs, cpp, oc, n_, oh, ow)
                else:
                    dz_t = [t for t, _ in pairs]
                    dz_v = [V(t.data_ptr(), ld, oc) for t, ld in pairs]
                for g in range(G):  # bias / beta-vector gradient of the chain's last layer: a plain column sum
                    tgt = bias_slot(last_layers[g].conv)
                    if tgt is not None:
                        ops.colsum(dz_t[g], tgt, accumulate=True)
                    if last_layers[g].vec_index is not None:
                        ops.colsum(dz_t[g], dvs[last_layers[g].vec_index], accumulate=True)
            dz_hw = (oh, ow)
            dx_out = None
            # ---- units in reverse
            for u in reversed(range(spec.U)):
                layers0 = spec.unit(0, u).layers
                nl = len(layers0)
                if u == 0:
                    uin, (uh, uw), uin_c = [V(x.data_ptr(), ldx, c)] * G, (h, w), c
                else:
                    pbs, pcp, (_, ph_, pw_) = saved[u - 1][-1]
                    uin_c = spec.unit(0, u - 1).layers[-1].conv.spec.out_ch
                    uin, (uh, uw) = [V(b.data_ptr(), pcp, uin_c) for b in pbs], (ph_, pw_)
                g_unit, g_unit_keep = dz_v, dz_t   # gradient wrt the unit output (kept alive: the residual operand of the
                # unit's first input-gradient conv reads it after the per-layer gradients have been replaced)
                assert layers0[-1].act is None, "the last layer of a unit carries no activation"
                for li in reversed(range(nl)):
                    convs = [spec.unit(g, u).layers[li].conv for g in range(G)]
                    sp = convs[0].spec
                    if li == 0:
                        lin, (ih, iw), lin_c = uin, (uh, uw), uin_c
                    else:
                        abs_, acp, (_, ah, aw) = saved[u][li - 1]
                        lin_c = layers0[li - 1].conv.spec.out_ch
                        lin, (ih, iw) = [V(b.data_ptr(), acp, lin_c) for b in abs_], (ah, aw)
                    oh_, ow_ = dz_hw
                    wts = [cv.weight for cv in convs]
                    if wts[0].requires_grad:   # weight gradient: (dz, layer input)
                        gp = [HF._grad_slot(wt).data_ptr() for wt in wts]
                        if sp.transposed:
                            ops.wgrad_multi(n, ih, iw, oh_, ow_, lin, dz_v, gp, wts[0].shape[0], wts[0].shape[1], sp.k, sp.stride, sp.pad,
                                            device=dev, label=spec.name)
                        else:
                            ops.wgrad_multi(n, oh_, ow_, ih, iw, dz_v, lin, gp, wts[0].shape[0], wts[0].shape[1], sp.k, sp.stride, sp.pad,
                                            device=dev, label=spec.name)
                    first = li == 0
                    if first and u == 0 and not ctx.needs_input_grad[0]:
                        break
                    if first and u == 0 and (not sp.transposed) and sp.in_ch <= 4 and sp.k[0] * sp.k[1] > 1:
                        assert G == 1   # RGB image gradient: GEMM + gather path (nothing to fuse at the chain input)
                        dx_out = [HF.scatter_conv(dz_t[0], convs[0].weight, None, sp, (ih, iw))]
                        break
                    # input gradient, with the backward of the layer below (and of the residual) fused in
                    below = [spec.unit(g, u).layers[li - 1] for g in range(G)] if not first else None
                    gbufs, ocp = _bufs(n * ih * iw, lin_c, G, dev)
                    ys, yts = views(gbufs, ocp, lin_c, n, ih, iw)
                    flags, masks, vec2 = 0, None, None
                    targets = [(None, None)] * G    # (address for the pre-mask sum, address for the post-mask sum)
                    if below is not None:
                        b0 = below[0]
                        if b0.act is not None:
                            flags |= L.EPI_RELUMASK if b0.act == "relu" else L.EPI_LRELUMASK
                            masks = lin
                            if b0.vec_index is not None:
                                flags |= L.EPI_MASKOFF
                                vec2 = vecs[b0.vec_index].data_ptr()
                        targets = [(dvs[b.vec_index].data_ptr() if b.vec_index is not None else None,
                                    None if bias_slot(b.conv) is None else bias_slot(b.conv).data_ptr()) for b in below]
                    elif u > 0:   # this launch produces the gradient of unit u-1's output = of its last layer
                        prev = [spec.unit(g, u - 1).layers[-1] for g in range(G)]
                        targets = [(dvs[b.vec_index].data_ptr() if b.vec_index is not None else None,
                                    None if bias_slot(b.conv) is None else bias_slot(b.conv).data_ptr()) for b in prev]
                    ress = g_unit if (first and spec.unit(0, u).residual) else None
                    want_cs = any(a is not None or b is not None for a, b in targets)
                    packs = [cv.spec.pack(cv.weight, True) for cv in convs]
                    cs = ops.conv_multi(n, oh_, ow_, ih, iw, dz_v, [pk.data_ptr() for pk in packs], ys, lin_c, sp.k, sp.stride, sp.pad,
                                        not sp.transposed, wrows=packs[0].shape[1], wcols=packs[0].shape[2], masks=masks, ress=ress,
                                        vec2=vec2, flags=flags, colsum=want_cs, wlayout=1 if sp.smallc_dgrad else 0, device=dev,
                                        label=spec.name + ".bwd")
                    if want_cs:
                        for g in range(G):
                            ptr, rows, ld = cs[g]
                            q.add(ptr, rows, ld, lin_c, targets[g][0], targets[g][1], True)
                    dz_v, dz_t, dz_hw = ys, yts, (ih, iw)
                if u == 0 and dx_out is None and ctx.needs_input_grad[0]:
                    dx_out = dz_t
            q.flush(("chain", id(spec)))
            if local:
                ops.WGRAD_DEFER.flush(("chain-local", dev.index))
        finally:
            ops.WGRAD_DEFER = prev_defer
        dx = None
        if dx_out is not None:
            dx = dx_out[0]
            for g in range(1, len(dx_out)):
                dx = dx + dx_out[g]
        # the beta-vector gradients live in the chain's persistent buffer, which the next backward of this spec zeroes.  Under
        # graph capture the fixed address is the point (the projections' backward consumes them inside the same captured
        # pass); eagerly autograd gets copies, so a second backward of the same chain (retain_graph, the chain used twice in
        # one graph) cannot clobber gradients the engine still holds
        keep = torch.cuda.is_current_stream_capturing()
        return (dx, None, dscale, dshift, None) + tuple((dvs[i] if keep else dvs[i].clone()) if ctx.needs_input_grad[5 + i] else None
                                                        for i in range(len(vecs))) \
            + (None,) * (ctx.nrest - len(vecs))


def run_chain(x, spec: ChainSpec, scale=None, shift=None, vecs=()):
    return _ChainFn.apply(x, spec, scale, shift, len(vecs), *vecs, *spec.parameters())


class _ChainReuseFn(torch.autograd.Function):
    """(handle, *params) -> the kept forward's outputs; backward = _ChainFn.backward on the kept activations (weight and bias gradients into the
    flat .grad slots; no gradient for the chain input: it was produced by another graph and is treated as detached)"""

    @staticmethod
    def forward(ctx, handle: ChainHandle, *params):
        ctx.spec, ctx.saved_acts, ctx.geom = handle.spec, handle.saved, handle.geom
        ctx.nrest = len(params)
        ctx.save_for_backward(handle.x, None, None)
        bufs, cp, (n_, oh, ow) = handle.saved[-1][-1]
        return tuple(_nchw(b, n_, oh, ow, handle.oc) for b in bufs)

    @staticmethod
    def backward(ctx, *douts):
        _ChainFn.backward(_KeptCtx(ctx), *douts)
        return (None,) * (1 + ctx.nrest)


def run_chain_reuse(handle: ChainHandle):
    """The outputs of a chain forward kept by KEEP_HANDLES, as a differentiable function of the chain's parameters (see above)."""
    return _ChainReuseFn.apply(handle, *handle.spec.parameters())
