"""torch.autograd plumbing around the HIP kernels.

Each Function's forward/backward is one or a few C-ABI launches; torch only owns the graph, the memory and the
stream.  Parameter gradients of conv layers are accumulated by the kernels straight into `param.grad` (so the
trainer can keep all gradients in one flat buffer for a single RCCL all-reduce).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as L
from . import ops

_weights_epoch = 0


def bump_weights_epoch() -> None:
    """Invalidate every weight pack (parameters were modified behind torch's back by something other than an optimiser
    that owns a PackTable -- the fused Adam refills its packs itself, see PackTable)."""
    global _weights_epoch
    _weights_epoch += 1


def _grad_slot(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class _PackEntry:
    """One persistent weight pack: destination buffer + how to refill it from its parameter.  A *sub-block* entry
    (`src_off` / `srcJ` / `dst_off` / `dld` / `tstride`, see crdr_pack_item) packs an input-channel range of the parameter
    into a row / column range of a wider pack shared with other parameters (the Charm's hoisted first-layer convs)."""
    __slots__ = ("weight", "dst", "I", "J", "T", "rows", "cols", "mode", "key", "src_off", "srcJ", "dst_off", "dld", "tstride")

    def __init__(self):
        self.src_off = self.srcJ = self.dst_off = self.dld = self.tstride = 0

    def item(self) -> "L.PackItem":
        return L.PackItem(src=self.weight.data_ptr() + self.src_off, dst=self.dst.data_ptr() + self.dst_off, I=self.I, J=self.J,
                          T=self.T, rows=self.rows, cols=self.cols, mode=self.mode, srcJ=self.srcJ, dld=self.dld,
                          tstride=self.tstride)

    def fill(self) -> None:
        lib = L.load()
        ops.register_persistent_pack(self.dst)
        ops.bump_pack_version(self.dst.data_ptr())
        if self.mode in (0, 1):
            it = self.item()
            L.check(lib.crdr_pack_weight_item(C.byref(it), ops._stream()), "pack_weight_item")
        else:
            L.check(lib.crdr_pack_weight(self.weight.data_ptr(), self.dst.data_ptr(), self.I, self.J, self.T, self.rows, self.cols,
                                         self.mode, ops._stream()), "pack_weight")


def sub_pack(weight: torch.Tensor, j0: int, j1: int, dst: torch.Tensor, dst_off: int, rows: int, cols: int, transposed: bool,
             dld: int = 0, tstride: int = 0) -> _PackEntry:
    """Register a sub-block pack: input channels [j0, j1) of `weight` [I][J][kh][kw] -> the [T][rows][cols] block that
    starts `dst_off` floats into `dst` (row stride `dld`, tap stride `tstride`; 0 = dense).  transposed=False: pack row =
    output channel i, column = input channel j (forward operand); True: row = j, column = i (input-gradient operand)."""
    global _pack_serial
    ops._require_gpu(weight)
    assert weight.is_contiguous() and weight.dim() == 4
    e = _PackEntry()
    e.key = None
    e.weight = weight.detach()
    e.I, e.J, e.T = weight.shape[0], j1 - j0, weight.shape[2] * weight.shape[3]
    e.src_off, e.srcJ = 4 * j0 * e.T, weight.shape[1]
    e.mode = 1 if transposed else 0
    e.rows, e.cols = rows, cols
    assert rows % 8 == 0 and cols % 32 == 0 and rows >= (e.J if transposed else e.I) and cols >= (e.I if transposed else e.J)
    e.dst, e.dst_off, e.dld, e.tstride = dst, 4 * dst_off, dld, tstride
    ops.register_persistent_pack(dst)
    _pack_entries.append(e)
    _pack_serial += 1
    return e


def ensure_fresh(entries) -> None:
    """Refill the entries whose parameter changed since their last fill (first use, load_state_dict, a foreign optimiser);
    the fused Adam keeps them fresh through its PackTable."""
    for e in entries:
        k = _current_key(e.weight)
        if e.key != k:
            e.fill()
            e.key = k


PACK_MISS_LOG = {} if __import__("os").environ.get("CRDR_DEBUG_PACK") == "1" else None  # {(I, J, T, mode, why): count}
_pack_entries = []   # every pack ever made, in creation order (PackTable selects the ones of one optimiser)
_pack_serial = 0     # bumped when _pack_entries grows


def _current_key(w: torch.Tensor):
    return (w.data_ptr(), w._version, _weights_epoch)


class ConvSpec:
    """Static description of one conv layer + its two persistent weight packs (forward / input-gradient operand)."""

    def __init__(self, in_ch, out_ch, k, stride, pad, transposed=False, out_pad=0):
        self.in_ch, self.out_ch, self.k = in_ch, out_ch, (k, k) if isinstance(k, int) else tuple(k)
        self.stride, self.pad, self.transposed, self.out_pad = stride, pad, transposed, out_pad
        self._packs = {}
        self.smallc = (not transposed) and in_ch <= 4  # RGB-input convs use the tap-major forward pack
        # ... and so does the input-gradient of an RGB-output ConvTranspose2d (a regular conv that reduces over <= 4 channels)
        self.smallc_dgrad = transposed and out_ch <= 4

    def out_hw(self, h, w):
        return (ops.conv_out_size(h, self.k[0], self.stride, self.pad, self.transposed, self.out_pad),
                ops.conv_out_size(w, self.k[1], self.stride, self.pad, self.transposed, self.out_pad))

    def pack(self, weight: torch.Tensor, for_dgrad: bool) -> torch.Tensor:
        # forward pack has rows = out channels: Conv2d weight [O][I] -> no transpose; ConvT weight [I][O] -> transpose
        transpose = (self.transposed != for_dgrad)
        tapmajor = (self.smallc and not for_dgrad) or (self.smallc_dgrad and for_dgrad)
        return self._pack(weight, transpose, 2 if tapmajor else int(transpose))

    def mark_stale(self) -> None:
        """The weight buffer was rewritten in place by a kernel (spectral norm): refill the packs on next use."""
        for ent in self._packs.values():
            ent.key = None

    def pack_scatter(self, weight: torch.Tensor) -> torch.Tensor:
        """[4 T][first weight dim] pack of an RGB-output transposed op done as GEMM + col2im (crdr_col2im_rgb)."""
        return self._pack(weight, "scatter", 3)

    def _pack(self, weight: torch.Tensor, slot, mode: int) -> torch.Tensor:
        global _pack_serial
        ent = self._packs.get(slot)
        key = _current_key(weight)
        if ent is not None and ent.key == key:
            return ent.dst
        if ent is None or ent.weight.data_ptr() != weight.data_ptr() or ent.weight.shape != weight.shape:
            ops._require_gpu(weight)
            assert weight.is_contiguous()
            ent = _PackEntry()
            ent.key = None
            ent.weight = weight.detach()
            ent.I, ent.J = weight.shape[0], weight.shape[1]
            ent.T = weight.shape[2] * weight.shape[3] if weight.dim() == 4 else 1
            ent.mode = mode
            if mode == 2:
                shape = (1, ops.round32(ent.I), ops.round32(4 * ent.T))
            elif mode == 3:
                shape = (1, ops.round32(4 * ent.T), ops.round32(ent.I))
            else:
                shape = (ent.T, ops.round32(ent.J), ops.round32(ent.I)) if mode else (ent.T, ops.round32(ent.I), ops.round32(ent.J))
            ent.rows, ent.cols = shape[1], shape[2]
            ent.dst = torch.empty(shape, dtype=torch.float32, device=weight.device)
            ops.register_persistent_pack(ent.dst)
            self._packs[slot] = ent
            _pack_entries.append(ent)
            _pack_serial += 1
        if PACK_MISS_LOG is not None:
            why = "new" if ent.key is None else "ptr" if ent.key[0] != key[0] else "version" if ent.key[1] != key[1] else "epoch"
            k = (ent.I, ent.J, ent.T, ent.mode, why)
            PACK_MISS_LOG[k] = PACK_MISS_LOG.get(k, 0) + 1
        ent.fill()
        ent.key = key
        return ent.dst


def scatter_conv(u: torch.Tensor, weight: torch.Tensor, bias, spec: "ConvSpec", out_hw) -> torch.Tensor:
    """RGB-output transposed op (ConvTranspose2d C -> <=4, or the input gradient of a Conv2d <=4 -> C) as ONE 1x1 GEMM
    with 4 T output columns + a gather, instead of per-tap / per-phase GEMMs padded from 3 to 32 output columns."""
    lib = L.load()
    n, _, h, w = u.shape
    T = spec.k[0] * spec.k[1]
    pk = spec.pack_scatter(weight)
    cols = ops.conv2d_raw(u, pk, 4 * T, (1, 1), 1, 0, False, (h, w))
    c = weight.shape[1]
    out = ops.empty_nhwc(n, c, out_hw[0], out_hw[1], u.device)
    L.check(lib.crdr_col2im_rgb(cols.data_ptr(), 4 * T, n, h, w, spec.k[0], spec.k[1], spec.stride, spec.pad,
                                None if bias is None else bias.data_ptr(), out.data_ptr(), ops.ld_for(c), out_hw[0], out_hw[1], c,
                                ops._stream()), "col2im_rgb")
    return out


class PackTable:
    """The packs whose parameter lives in one address range (an optimiser's flat buffer, or a partition of it), refilled
    by ONE launch right after that optimiser's update -- instead of one small launch per layer on next use.

    The device-side table has a fixed capacity and is rewritten in place when new packs appear, so a HIP graph that
    captured the launch keeps covering everything."""
    CAP = 4096
    ITEM = 56  # sizeof(crdr_pack_item)

    def __init__(self, flat: torch.Tensor, lo: int = 0, hi: Optional[int] = None):
        hi = flat.numel() if hi is None else hi
        self.lo, self.hi = flat.data_ptr() + 4 * lo, flat.data_ptr() + 4 * hi
        self.device = flat.device
        self.items = torch.zeros(self.CAP * self.ITEM, dtype=torch.uint8, device=flat.device)
        self.prefix = torch.zeros(self.CAP + 1, dtype=torch.int64, device=flat.device)
        self.meta = torch.zeros(2, dtype=torch.int64, device=flat.device)
        self.entries, self.singles = [], []
        self.filters = None
        self._seen = -1

    def _refresh(self) -> None:
        import numpy as np
        if self._seen == _pack_serial:
            return
        mine = [e for e in _pack_entries if self.lo <= e.weight.data_ptr() < self.hi and e.dst.device == self.device]
        ents = [e for e in mine if e.mode in (0, 1) and e.T <= 32]   # what the batched kernel takes
        self.singles = [e for e in mine if not (e.mode in (0, 1) and e.T <= 32)]
        if [id(e) for e in ents] != [id(e) for e in self.entries]:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PackTable: new weight packs appeared during graph capture (run eager warm-up iterations first)")
            assert len(ents) <= self.CAP
            rec = np.zeros(len(ents), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("I", "<i4"), ("J", "<i4"), ("T", "<i4"),
                                                        ("rows", "<i4"), ("cols", "<i4"), ("mode", "<i4"), ("srcJ", "<i4"),
                                                        ("dld", "<i4"), ("tstride", "<i8")]))
            pre = np.zeros(len(ents) + 1, dtype=np.int64)
            for k, e in enumerate(ents):
                rec[k] = (e.weight.data_ptr() + e.src_off, e.dst.data_ptr() + e.dst_off, e.I, e.J, e.T, e.rows, e.cols, e.mode,
                          e.srcJ, e.dld, e.tstride)
                pre[k + 1] = pre[k] + (e.rows // 8) * (e.cols // 32)
            assert rec.dtype.itemsize == self.ITEM
            if len(ents):
                self.items[:len(ents) * self.ITEM].copy_(torch.from_numpy(rec.view(np.uint8).copy()))
            self.prefix[:len(ents) + 1].copy_(torch.from_numpy(pre))
            self.meta.copy_(torch.tensor([len(ents), int(pre[-1])], dtype=torch.int64))
            self.entries = ents
        self._seen = _pack_serial

    def refill(self) -> None:
        """Refill every pack of the range from the current parameter values and mark them fresh."""
        self._refresh()
        for ptr in {e.dst.data_ptr() for e in self.entries}:
            ops.bump_pack_version(ptr)
        if self.entries:
            lib = L.load()
            L.check(lib.crdr_pack_weights_batched(self.items.data_ptr(), self.prefix.data_ptr(), self.meta.data_ptr(),
                                                  ops._stream()), "pack_weights_batched")
        for e in self.singles:
            e.fill()
        for e in self.entries + self.singles:
            e.key = _current_key(e.weight)
        # ... and, behind the packs, every transformed-filter cache of the F(4x4) kernel derived from them: one launch (ops.FilterTable)
        if self.filters is None:
            self.filters = ops.FilterTable(self.device)
        self.filters.refill({e.dst.data_ptr() for e in self.entries + self.singles})
        if torch.cuda.is_current_stream_capturing():
            ops.on_replay(self._replayed)   # (trainer/graphs.py: a replay runs these launches without this method)

    def _replayed(self) -> None:
        """The graph that captured refill() has just been replayed: refill()'s host-side bookkeeping without its launches.  The packs and the
        filter caches in the device tables AS THE REPLAYED LAUNCHES SAW THEM are fresh; packs / caches that appeared since (an eager
        validation pass at another image size, say) join the tables now and are covered from the next replay on -- until then their own
        staleness checks (pack keys, version stamps) make their launches refill / re-transform."""
        rewritten = {e.dst.data_ptr() for e in self.entries + self.singles}
        for ptr in rewritten:
            ops.bump_pack_version(ptr)
        if self.filters is not None:
            self.filters.replayed(rewritten)
        self._refresh()


def _flags(bias, act, vec2, res, gate, affine) -> int:
    f = 0
    if bias is not None:
        f |= L.EPI_BIAS
    if act == "relu":
        f |= L.EPI_RELU
    elif act == "lrelu":
        f |= L.EPI_LRELU
    elif act is not None:
        raise ValueError(act)
    if vec2 is not None:
        f |= L.EPI_VEC2
    if res is not None:
        f |= L.EPI_RES
    if gate:
        f |= L.EPI_GATE
    if affine:
        f |= L.EPI_AFFINE
    return f


class _FusedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, vec2, res, scale, shift, gx, gt, spec: ConvSpec, act, return_wgrad=False):
        ctx.return_wgrad = return_wgrad
        flags = _flags(bias, act, vec2, res, gx is not None, scale is not None)
        n, _, h, w = x.shape
        oh, ow = spec.out_hw(h, w)
        sig = ops.empty_nhwc(n, spec.out_ch, oh, ow, x.device) if gx is not None else None
        if spec.transposed and spec.out_ch <= 4 and not (flags & ~L.EPI_BIAS) and spec.k[0] * spec.k[1] > 1:
            out = scatter_conv(x, weight, bias, spec, (oh, ow))
        else:
            out = ops.conv2d_raw(x, spec.pack(weight, False), spec.out_ch, spec.k, spec.stride, spec.pad, spec.transposed,
                                 (oh, ow), bias=bias, flags=flags, vec2=vec2, res=res, scale=scale, shift=shift,
                                 gate_x=gx, gate_t=gt, sig_out=sig, wlayout=1 if spec.smallc else 0)
        ctx.spec, ctx.flags, ctx.in_hw = spec, flags, (h, w)
        if ops.RELU_MASK_SINK is not None and act == "relu" and vec2 is None and res is None and scale is None and gx is None and any(ctx.needs_input_grad):   # (a no-grad pass -- the high-rate reconstruction -- has no backward to take masks for)
            ops.RELU_MASK_SINK(weight, out, None)
        ctx.has = (bias is not None, vec2 is not None, res is not None, scale is not None, gx is not None)
        need_out = flags & (L.EPI_RELU | L.EPI_LRELU | L.EPI_AFFINE)
        ctx.save_for_backward(x, weight, bias, vec2, scale, shift, gt, sig, out if need_out else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias, vec2, scale, shift, gt, sig, out = ctx.saved_tensors
        spec, flags = ctx.spec, ctx.flags
        has_bias, has_vec2, has_res, has_aff, has_gate = ctx.has
        needs = ctx.needs_input_grad
        dout, _ = ops.nhwc(dout)
        heavy = flags & (L.EPI_RELU | L.EPI_LRELU | L.EPI_AFFINE | L.EPI_GATE)
        gres = dgt = dscale = dshift = dvec2 = None
        if heavy or has_vec2:
            dz, gres, dgt, cs = ops.epilogue_bwd(dout, out, flags, vec2=vec2, scale=scale, shift=shift, gate_t=gt, sig=sig,
                                                 need_dz=bool(heavy),
                                                 dbias_accum=_grad_slot(bias) if (has_bias and needs[2]) else None)
            if dz is None:
                dz = dout
            if has_vec2:
                dvec2 = cs[1]
            if has_aff:
                dscale, dshift = cs[2], cs[3]
        else:
            dz = dout
            if has_bias and needs[2]:
                ops.colsum(dz, _grad_slot(bias), accumulate=True)
        if gres is None and (has_res or has_gate):
            gres = dout  # no affine in front: the residual branch sees dout itself
        dx = None
        if needs[0]:
            if (not spec.transposed) and spec.in_ch <= 4 and spec.k[0] * spec.k[1] > 1:
                dx = scatter_conv(dz, weight, None, spec, ctx.in_hw)  # RGB image gradient: one GEMM + gather
            else:
                dx = ops.conv2d_raw(dz, spec.pack(weight, True), x.shape[1], spec.k, spec.stride, spec.pad,
                                    not spec.transposed, ctx.in_hw, wlayout=1 if spec.smallc_dgrad else 0)
        dw = None
        if needs[1]:
            # a parameter's gradient is accumulated straight into its (flat) slot; a derived weight (spectral norm) gets
            # its gradient returned through autograd instead, reduced immediately
            g = torch.empty_like(weight) if ctx.return_wgrad else _grad_slot(weight)
            g4 = g if g.dim() == 4 else g.view(g.shape[0], g.shape[1], 1, 1)
            kw = dict(accumulate=not ctx.return_wgrad, defer=not ctx.return_wgrad)
            if spec.transposed:
                ops.conv2d_wgrad_raw(x, dz, g4, spec.k, spec.stride, spec.pad, **kw)
            else:
                ops.conv2d_wgrad_raw(dz, x, g4, spec.k, spec.stride, spec.pad, **kw)
            dw = g if ctx.return_wgrad else None
        return (dx, dw, None, dvec2, gres if has_res else None, dscale, dshift, gres if has_gate else None,
                dgt, None, None, None)


class _SmallLinear(torch.autograd.Function):
    """1x1 conv over <= 16 single-pixel rows (conditioning MLP / projections): GEMV kernels on the raw parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        lib = L.load()
        ops._require_gpu(x)
        m, i, o = x.shape[0], x.shape[1], weight.shape[0]
        x2 = x.reshape(m, i).contiguous()
        y = torch.empty((m, o), dtype=torch.float32, device=x.device)
        L.check(lib.crdr_linear_fwd(x2.data_ptr(), m, i, i, weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                    y.data_ptr(), o, o, int(relu), ops._stream()), "linear_fwd")
        ctx.relu = relu
        ctx.save_for_backward(x2, weight, bias, y if relu else None)
        return y.view(m, o, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, y = ctx.saved_tensors
        lib = L.load()
        m, i, o = x2.shape[0], x2.shape[1], weight.shape[0]
        dy2 = dy.reshape(m, o).contiguous()
        needs = ctx.needs_input_grad
        dx = torch.empty((m, i), dtype=torch.float32, device=dy.device) if needs[0] else None
        dw = _grad_slot(weight) if needs[1] else None
        db = _grad_slot(bias) if (bias is not None and needs[2]) else None
        L.check(lib.crdr_linear_bwd(x2.data_ptr(), m, i, i, weight.data_ptr(), dy2.data_ptr(), o, None if y is None else y.data_ptr(),
                                    o, o, None if dx is None else dx.data_ptr(), i, None if dw is None else dw.data_ptr(),
                                    None if db is None else db.data_ptr(), ops._stream()), "linear_bwd")
        return (None if dx is None else dx.view(m, i, 1, 1)), None, None, None


class _LinearGroup(torch.autograd.Function):
    """Several linear layers of ONE input (the nine beta projections of a bottleneck stack read the same [1, 512] vector):
    one launch forward, two backward; the input gradient is summed over the layers inside the kernel (fixed order), so
    autograd sees a single consumer of the input.  apply(x, n, w_0 .. w_{n-1}, b_0 .. b_{n-1}) -> n outputs [M, O_g]."""

    @staticmethod
    def forward(ctx, x, n: int, *wb):
        lib = L.load()
        ops._require_gpu(x)
        ws, bs = wb[:n], wb[n:]
        m, i = x.shape[0], x.shape[1]
        x2 = x.reshape(m, i).contiguous()
        g = L.LinearGroup()
        ys = []
        for k in range(n):
            o = ws[k].shape[0]
            assert ws[k].is_contiguous() and ws[k].numel() == o * i
            y = torch.empty((m, o), dtype=torch.float32, device=x.device)
            g.w[k], g.y[k], g.O[k] = ws[k].data_ptr(), y.data_ptr(), o
            g.b[k] = None if bs[k] is None else bs[k].data_ptr()
            ys.append(y)
        L.check(lib.crdr_linear_group_fwd(x2.data_ptr(), m, i, i, C.byref(g), n, ops._stream()), "linear_group_fwd")
        ctx.n = n
        ctx.x_shape = tuple(x.shape)
        ctx.has_b = [b is not None for b in bs]
        ctx.save_for_backward(x2, *ws, *[b for b in bs if b is not None])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        x2, ws, rest = saved[0], saved[1:1 + n], list(saved[1 + n:])
        lib = L.load()
        m, i = x2.shape
        needs = ctx.needs_input_grad
        g = L.LinearGroup()
        keep = []
        for k in range(n):
            o = ws[k].shape[0]
            dy = dys[k]
            dy = torch.zeros((m, o), dtype=torch.float32, device=x2.device) if dy is None else dy.reshape(m, o).contiguous()
            keep.append(dy)
            b = rest.pop(0) if ctx.has_b[k] else None
            g.w[k], g.dy[k], g.O[k] = ws[k].data_ptr(), dy.data_ptr(), o
            g.dw[k] = _grad_slot(ws[k]).data_ptr() if needs[2 + k] else None
            g.db[k] = _grad_slot(b).data_ptr() if (b is not None and needs[2 + n + k]) else None
        dx = torch.empty((m, i), dtype=torch.float32, device=x2.device) if needs[0] else None
        L.check(lib.crdr_linear_group_bwd(x2.data_ptr(), m, i, i, C.byref(g), n, None if dx is None else dx.data_ptr(), i,
                                          ops._stream()), "linear_group_bwd")
        return (None if dx is None else dx.view(ctx.x_shape), None) + (None,) * (2 * n)


def linear_group(x, layers):
    """y_g = layer_g(x) for 1x1 conv / linear layers (weight [O, I(, 1, 1)], optional bias) sharing the input x [M <= 16, I(, 1, 1)]:
    one grouped launch per direction.  Returns a list of [M, O_g] tensors."""
    outs = []
    for c0 in range(0, len(layers), L.MAX_GROUP):
        chunk = layers[c0:c0 + L.MAX_GROUP]
        ws = [ly.weight for ly in chunk]  # the parameters themselves ([O, I] or [O, I, 1, 1]): their .grad slots are written
        bs = [ly.bias for ly in chunk]
        outs += list(_LinearGroup.apply(x, len(chunk), *ws, *bs))
    return outs


def fused_conv(x, weight, bias, spec: ConvSpec, *, act: Optional[str] = None, vec2=None, res=None,
               affine: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               gate: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, return_wgrad: bool = False):
    if (not return_wgrad and spec.k == (1, 1) and spec.stride == 1 and spec.pad == 0 and not spec.transposed and x.dim() == 4 and x.shape[2] == 1
            and x.shape[3] == 1 and x.shape[0] <= 16 and vec2 is None and res is None and affine is None and gate is None
            and act in (None, "relu") and weight.is_contiguous()):
        return _SmallLinear.apply(x, weight, bias, act == "relu")
    scale, shift = affine if affine is not None else (None, None)
    gx, gt = gate if gate is not None else (None, None)
    return _FusedConv.apply(x, weight, bias, vec2, res, scale, shift, gx, gt, spec, act, return_wgrad)


class _SpectralNorm(torch.autograd.Function):
    """weight_orig -> weight_orig / sigma with torch.nn.utils.spectral_norm's semantics (one power iteration per
    training-mode call, u / v updated in place and treated as constants by the backward)."""

    @staticmethod
    def forward(ctx, w_orig, u, v, training: bool, out_buf: torch.Tensor, eps: float):
        lib = L.load()
        ops._require_gpu(w_orig)
        o, k = w_orig.shape[0], w_orig.numel() // w_orig.shape[0]
        sigma = torch.empty(1, dtype=torch.float32, device=w_orig.device)
        ws, wsn = ops.workspace((o + k + 2) * 4, w_orig.device)
        L.check(lib.crdr_spectral_norm_fwd(w_orig.data_ptr(), o, k, u.data_ptr(), v.data_ptr(), int(training), float(eps),
                                           out_buf.data_ptr(), sigma.data_ptr(), ws, wsn, ops._stream()), "spectral_norm_fwd")
        ctx.save_for_backward(w_orig, u.clone() if training else u, v.clone() if training else v, sigma)
        return out_buf.detach()  # a fresh alias of the persistent buffer (stable address for the weight packs)

    @staticmethod
    def backward(ctx, dw_sn):
        w_orig, u, v, sigma = ctx.saved_tensors
        lib = L.load()
        o, k = w_orig.shape[0], w_orig.numel() // w_orig.shape[0]
        g = _grad_slot(w_orig)
        nb = lib.crdr_reduce_workspace(o * k) + 16
        ws, wsn = ops.workspace(nb, w_orig.device)
        L.check(lib.crdr_spectral_norm_bwd(dw_sn.contiguous().data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(),
                                           sigma.data_ptr(), o, k, g.data_ptr(), ws, wsn, ops._stream()), "spectral_norm_bwd")
        return None, None, None, None, None, None


def spectral_norm_weight(w_orig, u, v, training: bool, out_buf: torch.Tensor, eps: float = 1e-12):
    return _SpectralNorm.apply(w_orig, u, v, training, out_buf, eps)


class _InterpCaVectors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, B, q: float):
        lib = L.load()
        Lv, Cc = W.shape[0], W.shape[2]
        scale = torch.empty(Cc, dtype=torch.float32, device=W.device)
        shift = torch.empty(Cc, dtype=torch.float32, device=W.device)
        L.check(lib.crdr_interp_ca_params(W.data_ptr(), None if B is None else B.data_ptr(), Lv, Cc, float(q),
                                          scale.data_ptr(), shift.data_ptr(), ops._stream()), "interp_ca_params")
        ctx.q, ctx.has_b = float(q), B is not None
        ctx.B = B  # (a parameter: only its .grad slot is touched in backward)
        ctx.save_for_backward(W)
        return scale, shift

    @staticmethod
    def backward(ctx, dscale, dshift):
        (W,) = ctx.saved_tensors
        lib = L.load()
        Lv, Cc = W.shape[0], W.shape[2]
        # the kernel accumulates: leaf parameters get their rows added straight into .grad (the flat gradient buffer), like
        # the bias gradients of the conv layers -- no zero-filled temporary, no autograd add per module
        direct = W.is_leaf and (ctx.B is None or ctx.B.is_leaf)
        if direct:
            dW = _grad_slot(W) if ctx.needs_input_grad[0] else None
            dB = _grad_slot(ctx.B) if (ctx.has_b and ctx.needs_input_grad[1]) else None
        else:
            dW = torch.zeros_like(W)
            dB = torch.zeros_like(W) if ctx.has_b else None
        if dW is not None or dB is not None:
            scratch = None
            if dW is None:  # (scale frozen, bias trained: the kernel still wants a destination)
                scratch = dW = torch.zeros_like(W)
            L.check(lib.crdr_interp_ca_params_bwd(W.data_ptr(), Lv, Cc, ctx.q, dscale.contiguous().data_ptr(),
                                                  dshift.contiguous().data_ptr(), dW.data_ptr(),
                                                  None if dB is None else dB.data_ptr(), ops._stream()), "interp_ca_params_bwd")
            if scratch is not None:
                dW = None
        if direct:
            return None, None, None
        return dW, dB, None


def interp_ca_vectors(W, B, q: float):
    return _InterpCaVectors.apply(W, B, float(q))


class _Affine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        lib = L.load()
        x, ld = ops.nhwc(x)
        n, c, h, w = x.shape
        y = ops.empty_nhwc(n, c, h, w, x.device)
        L.check(lib.crdr_affine(x.data_ptr(), ld, scale.data_ptr(), shift.data_ptr(), y.data_ptr(), ops.ld_for(c), n * h * w, c,
                                ops._stream()), "affine")
        ctx.save_for_backward(y, scale, shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale, shift = ctx.saved_tensors
        dz, _, _, cs = ops.epilogue_bwd(dy, y, L.EPI_AFFINE, scale=scale, shift=shift)
        return dz, cs[2], cs[3]


def affine(x, scale, shift):
    return _Affine.apply(x, scale, shift)


class _Lrp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, z):
        lib = L.load()
        a, lda = ops.nhwc(a)
        z, ldz = ops.nhwc(z)
        n, c, h, w = a.shape
        y = ops.empty_nhwc(n, c, h, w, a.device)
        L.check(lib.crdr_lrp(a.data_ptr(), lda, z.data_ptr(), ldz, y.data_ptr(), c, n * h * w, c, ops._stream()), "lrp")
        ctx.save_for_backward(z)
        return y

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        lib = L.load()
        dy, lddy = ops.nhwc(dy)
        z, ldz = ops.nhwc(z)
        n, c, h, w = z.shape
        dz = ops.empty_nhwc(n, c, h, w, z.device)
        L.check(lib.crdr_lrp_bwd(dy.data_ptr(), lddy, z.data_ptr(), ldz, dz.data_ptr(), c, n * h * w, c, ops._stream()), "lrp_bwd")
        return dy, dz


def lrp(a, z):
    """a + 0.5 * tanh(z)"""
    return _Lrp.apply(a, z)


def _gc_desc(y, ldy, ldmu, ldsg, ldyh, scale_bound, lik_bound):
    n, c, h, w = y.shape
    return L.GcDesc(N=n, HW=h * w, C=c, ldy=ldy, ldmu=ldmu, ldsigma=ldsg, ldyhat=ldyh, scale_bound=scale_bound,
                    likelihood_bound=lik_bound)


def gauss_cond_fwd2(d, io, device) -> None:
    """crdr_gauss_cond_fwd2 with the scratch for its per-block partial bit sums attached (fixed-order finishing pass)."""
    lib = L.load()
    io.ws, io.ws_bytes = ops.workspace(lib.crdr_gauss_cond_fwd_workspace(C.byref(d)), device)
    L.check(lib.crdr_gauss_cond_fwd2(C.byref(d), C.byref(io), ops._stream()), "gauss_cond_fwd2")


class _GaussCond(torch.autograd.Function):
    """(y, mu, sigma, noise) -> (y_hat, bits_noisy[N], bits_quant[N], lik_noisy?, lik_quant?)"""

    @staticmethod
    def forward(ctx, y, mu, sigma, noise, scale_bound, lik_bound, want_lik):
        lib = L.load()
        y, ldy = ops.nhwc(y)
        mu, ldmu = ops.nhwc(mu)
        sigma, ldsg = ops.nhwc(sigma)
        n, c, h, w = y.shape
        dev = y.device
        yhat = ops.empty_nhwc(n, c, h, w, dev)
        bits_n = torch.zeros(n, dtype=torch.float32, device=dev)
        bits_q = torch.zeros(n, dtype=torch.float32, device=dev)
        lik_n = ops.empty_nhwc(n, c, h, w, dev) if (want_lik and noise is not None) else None
        lik_q = ops.empty_nhwc(n, c, h, w, dev) if want_lik else None
        if noise is not None:
            noise, ldn = ops.nhwc(noise)
            if ldn != c:
                noise = noise.contiguous(memory_format=torch.channels_last)
        d = L.GcDesc2(N=n, HW=h * w, C=c, ldy=ldy, ldmu=ldmu, ldsigma=ldsg, ldyhat=c, scale_bound=scale_bound, likelihood_bound=lik_bound)
        io = L.GcIO(y=y.data_ptr(), mu=mu.data_ptr(), sigma=sigma.data_ptr(), noise=ops._p(noise), yhat=yhat.data_ptr(),
                    lik_noisy=ops._p(lik_n), lik_quant=ops._p(lik_q), bits_noisy=bits_n.data_ptr(), bits_quant=bits_q.data_ptr())
        gauss_cond_fwd2(d, io, dev)
        ctx.bounds = (scale_bound, lik_bound)
        ctx.save_for_backward(y, mu, sigma, noise)
        ctx.mark_non_differentiable(bits_q)
        if lik_n is not None:
            ctx.mark_non_differentiable(lik_n)
        if lik_q is not None:
            ctx.mark_non_differentiable(lik_q)
        return yhat, bits_n, bits_q, lik_n, lik_q

    @staticmethod
    def backward(ctx, dyhat, dbits_n, _dq, _dln, _dlq):
        y, mu, sigma, noise = ctx.saved_tensors
        if noise is None:
            raise L.CrdrHipError("gauss_cond: backward needs the noisy (training) forward")
        lib = L.load()
        y, ldy = ops.nhwc(y)
        mu, ldmu = ops.nhwc(mu)
        sigma, ldsg = ops.nhwc(sigma)
        n, c, h, w = y.shape
        dev = y.device
        if dbits_n is None:
            dbits_n = torch.zeros(n, dtype=torch.float32, device=dev)
        lddyh = 0
        if dyhat is not None:
            dyhat, lddyh = ops.nhwc(dyhat)
        dy, dmu, dsg = (ops.empty_nhwc(n, c, h, w, dev) for _ in range(3))
        d = _gc_desc(y, ldy, ldmu, ldsg, c, *ctx.bounds)
        L.check(lib.crdr_gauss_cond_bwd(C.byref(d), y.data_ptr(), mu.data_ptr(), sigma.data_ptr(), noise.data_ptr(),
                                        dbits_n.contiguous().data_ptr(), ops._p(dyhat), lddyh, dy.data_ptr(),
                                        dmu.data_ptr(), dsg.data_ptr(), ops._stream()), "gauss_cond_bwd")
        return dy, dmu, dsg, None, None, None, None


def gauss_cond(y, mu, sigma, noise, scale_bound=0.11, lik_bound=1e-9, want_lik=False):
    return _GaussCond.apply(y, mu, sigma, noise, float(scale_bound), float(lik_bound), bool(want_lik))


class _EntropyBottleneck(torch.autograd.Function):
    """(z, params[C,58], medians[C], noise) -> (z_hat, lik, bits[N])"""

    @staticmethod
    def forward(ctx, z, params, medians, noise, lik_bound):
        lib = L.load()
        z, ldz = ops.nhwc(z)
        n, c, h, w = z.shape
        if ldz != c:
            z = z.contiguous(memory_format=torch.channels_last)
        if noise is not None:
            noise, ldn = ops.nhwc(noise)
            if ldn != c:
                noise = noise.contiguous(memory_format=torch.channels_last)
        dev = z.device
        zhat, lik = ops.empty_nhwc(n, c, h, w, dev), ops.empty_nhwc(n, c, h, w, dev)
        bits = torch.zeros(n, dtype=torch.float32, device=dev)
        params = params.contiguous()
        medians = medians.contiguous()
        L.check(lib.crdr_entropy_bottleneck_fwd(z.data_ptr(), ops._p(noise), params.data_ptr(), medians.data_ptr(), n, h * w,
                                                c, lik_bound, zhat.data_ptr(), lik.data_ptr(), bits.data_ptr(), ops._stream()),
                "entropy_bottleneck_fwd")
        ctx.lik_bound = lik_bound
        ctx.save_for_backward(z, params, noise)
        ctx.mark_non_differentiable(lik)
        return zhat, lik, bits

    @staticmethod
    def backward(ctx, dzhat, _dlik, dbits):
        z, params, noise = ctx.saved_tensors
        if noise is None:
            raise L.CrdrHipError("entropy_bottleneck: backward needs the noisy (training) forward")
        lib = L.load()
        n, c, h, w = z.shape
        dev = z.device
        if dbits is None:
            dbits = torch.zeros(n, dtype=torch.float32, device=dev)
        if dzhat is not None:
            dzhat, ldd = ops.nhwc(dzhat)
            if ldd != c:
                dzhat = dzhat.contiguous(memory_format=torch.channels_last)
        dz = ops.empty_nhwc(n, c, h, w, dev)
        dparams = torch.empty_like(params)
        L.check(lib.crdr_entropy_bottleneck_bwd(z.data_ptr(), noise.data_ptr(), params.data_ptr(), n, h * w, c, ctx.lik_bound,
                                                dbits.contiguous().data_ptr(), ops._p(dzhat), dz.data_ptr(),
                                                dparams.data_ptr(), ops._stream()), "entropy_bottleneck_bwd")
        # z_hat = ste_round(z - median) + median: no gradient reaches the medians through it
        return dz, dparams, None, None, None


def entropy_bottleneck(z, params, medians, noise, lik_bound=1e-9):
    return _EntropyBottleneck.apply(z, params, medians, noise, float(lik_bound))


# ---------------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------------
def _flat(t: torch.Tensor) -> torch.Tensor:
    """The memory behind a (possibly channel-padded NHWC) tensor as a flat contiguous buffer.  Padding lanes
    must hold equal values in both operands of a difference (they are zero by construction)."""
    ops._require_gpu(t)
    if t.is_contiguous():
        return t.reshape(-1)
    if t.dim() == 4:
        t2, ld = ops.nhwc(t)
        n, c, h, w = t2.shape
        return torch.as_strided(t2, (n * h * w * ld,), (1,), t2.storage_offset())
    return t.contiguous().reshape(-1)


def _same_layout(a, b):
    fa, fb = _flat(a), _flat(b)
    if fa.numel() != fb.numel():
        a = a.contiguous(memory_format=torch.channels_last)
        b = b.contiguous(memory_format=torch.channels_last)
        fa, fb = _flat(a), _flat(b)
    return fa, fb


class _SqDiffSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = L.load()
        fa, fb = _same_layout(a, b)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        nb = lib.crdr_reduce_workspace(fa.numel())
        ws, wsn = ops.workspace(nb, a.device)
        L.check(lib.crdr_sqdiff_sum(fa.data_ptr(), fb.data_ptr(), fa.numel(), out.data_ptr(), ws, wsn, ops._stream()), "sqdiff_sum")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        lib = L.load()
        fa, fb = _same_layout(a, b)
        da = torch.empty_like(fa) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(fb) if ctx.needs_input_grad[1] else None
        L.check(lib.crdr_sqdiff_bwd(fa.data_ptr(), fb.data_ptr(), fa.numel(), g.contiguous().data_ptr(), 1.0, ops._p(da),
                                    ops._p(db), ops._stream()), "sqdiff_bwd")

        def back(flat, like):
            if flat is None:
                return None
            like2, ld = ops.nhwc(like) if like.dim() == 4 and not like.is_contiguous() else (like, None)
            if ld is None:
                return flat.view(like.shape)
            n, c, h, w = like2.shape
            return flat.view(n, h, w, ld).permute(0, 3, 1, 2)[:, :c]
        return back(da, a), back(db, b)


def sqdiff_sum(a, b):
    return _SqDiffSum.apply(a, b)


class _BceDiffSum(torch.autograd.Function):
    """sum BCEWithLogits(p - q, target)"""

    @staticmethod
    def forward(ctx, p, q, target):
        lib = L.load()
        fp, fq = p.contiguous().reshape(-1), q.contiguous().reshape(-1)
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        nb = lib.crdr_reduce_workspace(fp.numel())
        ws, wsn = ops.workspace(nb, p.device)
        L.check(lib.crdr_bce_diff_sum(fp.data_ptr(), fq.data_ptr(), fp.numel(), float(target), out.data_ptr(), ws, wsn,
                                      ops._stream()), "bce_diff_sum")
        ctx.target = float(target)
        ctx.save_for_backward(fp, fq)
        ctx.shape = p.shape
        return out

    @staticmethod
    def backward(ctx, g):
        fp, fq = ctx.saved_tensors
        lib = L.load()
        dp = torch.empty_like(fp) if ctx.needs_input_grad[0] else None
        dq = torch.empty_like(fq) if ctx.needs_input_grad[1] else None
        L.check(lib.crdr_bce_diff_bwd(fp.data_ptr(), fq.data_ptr(), fp.numel(), ctx.target, g.contiguous().data_ptr(), 1.0,
                                      ops._p(dp), ops._p(dq), ops._stream()), "bce_diff_bwd")
        return (None if dp is None else dp.view(ctx.shape)), (None if dq is None else dq.view(ctx.shape)), None


def bce_diff_sum(p, q, target: float):
    return _BceDiffSum.apply(p, q, float(target))


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = L.load()
        x, ld = ops.nhwc(x)
        n, c, h, w = x.shape
        if ld != c:
            x = x.contiguous(memory_format=torch.channels_last)
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        y = ops.empty_nhwc(n, c, oh, ow, x.device)
        L.check(lib.crdr_maxpool3s2_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, ops._stream()), "maxpool_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lib = L.load()
        n, c, h, w = x.shape
        dy, ld = ops.nhwc(dy)
        if ld != c:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = ops.empty_nhwc(n, c, h, w, x.device)
        L.check(lib.crdr_maxpool3s2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, h, w, c, ops._stream()), "maxpool_bwd")
        return dx


def maxpool3s2(x):
    return _MaxPool3s2.apply(x)


class _LpipsLayer(torch.autograd.Function):
    """(f_real, f_fake, lin[C]) -> per-image distance [N]; gradient flows to f_fake only."""

    @staticmethod
    def forward(ctx, f0, f1, lin):
        lib = L.load()
        f0, l0 = ops.nhwc(f0)
        f1, l1 = ops.nhwc(f1)
        n, c, h, w = f0.shape
        if l0 != c:
            f0 = f0.contiguous(memory_format=torch.channels_last)
        if l1 != c:
            f1 = f1.contiguous(memory_format=torch.channels_last)
        out = torch.zeros(n, dtype=torch.float32, device=f0.device)
        ws, wsn = ops.workspace(n * 64 * 4, f0.device)
        L.check(lib.crdr_lpips_layer_fwd(f0.data_ptr(), f1.data_ptr(), lin.data_ptr(), n, h * w, c, out.data_ptr(), ws, wsn,
                                         ops._stream()), "lpips_layer_fwd")
        ctx.save_for_backward(f0, f1, lin)
        return out

    @staticmethod
    def backward(ctx, g):
        f0, f1, lin = ctx.saved_tensors
        lib = L.load()
        n, c, h, w = f0.shape
        df1 = ops.empty_nhwc(n, c, h, w, f0.device)
        L.check(lib.crdr_lpips_layer_bwd(f0.data_ptr(), f1.data_ptr(), lin.data_ptr(), n, h * w, c, g.contiguous().data_ptr(),
                                         df1.data_ptr(), ops._stream()), "lpips_layer_bwd")
        return None, df1, None


def lpips_layer(f_real, f_fake, lin):
    return _LpipsLayer.apply(f_real, f_fake, lin)
