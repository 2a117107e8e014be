"""Block-level autograd Functions: each Conformer / Transformer sub-block is one Function whose
forward and backward are explicit kernel sequences over the C ABI (espnet_amd.ops).

torch.autograd is only the tape between blocks.  Weight gradients are accumulated straight into
the flat gradient arena (``param._eamd_grad`` installed by espnet_amd.train.FlatParams) when
present, otherwise they are returned to autograd like any other Function would.
"""
import math
import os

import torch

from . import ops
from .ops import (ACT_NONE, ACT_RELU, ACT_SWISH, EPI_DACT_FACTOR, EPI_MUL_AUX, EPI_MUL_DSWISH, EPI_MUL_RELU_MASK, EPI_NONE,
                  EPI_RELU)


class GradSink:
    """Where backward kernels accumulate parameter gradients."""

    on_done = None   # optional callback(params): one use of these parameters has written its gradients (DDP buckets)
    on_use = None    # optional callback(params): a forward pass recorded one more use of these parameters

    @staticmethod
    def use(params):
        """every block forward passes its parameters through here: a parameter that several blocks (or several
        time steps of one recurrent cell) use is final only when EACH use has reported from backward"""
        if GradSink.on_use is not None and torch.is_grad_enabled():
            GradSink.on_use(params)
        return params

    def __init__(self, params):
        self.params = list(params)
        self.ret = [None] * len(self.params)

    def buf(self, i):
        p = self.params[i]
        if p is None:        # bare modules have no LayerNorm in front
            return None
        g = getattr(p, "_eamd_grad", None)
        if g is not None:
            return g
        if self.ret[i] is None:
            self.ret[i] = torch.zeros_like(p)
        return self.ret[i]

    def results(self):
        if GradSink.on_done is not None:
            ops.wgrad_join()   # the all-reduce of these gradients must see the side-stream GEMMs
            GradSink.on_done(self.params)
        return tuple(self.ret)


# ---- incoming-gradient dropout fused into the producer -------------------------------------------------------------
# In backward every block first applies its output dropout to the incoming gradient and casts it for the GEMMs:
# one more pass over a [rows, D] tensor.  That gradient is produced by the LayerNorm backward of the NEXT block, so the
# forward tags each block output with its (p, salt) and the next block's LayerNorm backward writes the dropped bf16
# copy alongside dx (eamd_layernorm_bwd_drop); the copy travels as an attribute of the gradient tensor.  Anything that
# breaks the chain (gradient accumulation from several consumers, another dtype / width) falls back to the plain pass.
FUSE_GRAD_DROP = True


def _tag_out(out, p_out, s_out):
    if FUSE_GRAD_DROP and p_out > 0.0:
        out._eamd_out_drop = (float(p_out), int(s_out))
    return out


def _prev_drop(x):
    return getattr(x, "_eamd_out_drop", None) if FUSE_GRAD_DROP else None


def _grad_in(dout, do, p_out, s_out):
    """bf16 GEMM operand of the incoming gradient: the fused copy if the producer made one for this (p, salt)"""
    pre = getattr(dout, "_eamd_dropped", None)
    if pre is not None and pre[1] == (float(p_out), int(s_out)) and pre[0].numel() == do.numel():
        return pre[0].view(do.shape)
    return ops.dropout(do, p_out, s_out, out_dtype=ops.act_dtype()) if p_out > 0.0 else ops.to_act(do)


def _grad_operand(dout, do, p_out, s_out):
    """-> (operand, a_drop) for the GEMMs that consume a block's incoming gradient: in fp32 mode the block-output dropout
    of the gradient is applied while those GEMMs stage the operand (no dropped copy is written)"""
    if p_out > 0.0 and ops.f32_operand_drop():
        return do, (float(p_out), int(s_out))
    return _grad_in(dout, do, p_out, s_out), None


def _ln_fwd_in(x2, ln_w, ln_b, eps, adt):
    """pre-norm of a block; eps None = bare module (the reference's standalone MultiHeadedAttention /
    PositionwiseFeedForward / ConvolutionModule forward: no LayerNorm in front, no residual behind)"""
    if eps is None:
        return ops.to_act(x2), None, None
    return ops.layernorm_fwd(x2, ln_w, ln_b, eps, adt)


def _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, gbuf, bbuf, prev, shp):
    """LayerNorm backward of a block + residual; with `prev` = the previous block's (p, salt) also its dropped bf16 copy"""
    if mean is None:       # bare module: the gradient of the branch input is the block's input gradient
        return dxn.float().view(shp) if dxn.dtype != torch.float32 else dxn.view(shp)
    if prev is not None and x2.shape[1] in (256, 512):
        dx, dx16 = ops.layernorm_bwd(dxn, x2, ln_w, mean, rstd, do, gbuf, bbuf, drop=prev)
        out = dx.view(shp)
        out._eamd_dropped = (dx16, prev)
        return out
    return ops.layernorm_bwd(dxn, x2, ln_w, mean, rstd, do, gbuf, bbuf).view(shp)


def _rp_ln_bwd(dy, img, x2, ln_w, mean, rstd, do, gbuf, bbuf, prev, shp):
    """dx = LayerNorm'(dy W) + do by ONE eamd_rowproj launch (the input-gradient product with the LayerNorm backward as its
    epilogue); the gamma / beta partials join the pass's batched second stage; with `prev` also the dropped copy of dx"""
    M, D = x2.shape
    ws = ops.rowproj_lnb_ws(M, x2.device)
    dcopy = torch.empty(M, D, device=x2.device, dtype=torch.float32) if prev is not None else None
    dx = ops.rowproj(dy, img, D, lnb=(x2, ln_w, mean, rstd, do, ws, dcopy, prev))
    ops.ln_partials_reduce(ws, gbuf, bbuf, (M + 31) // 32, D)
    out = dx.view(shp)
    if prev is not None:
        out._eamd_dropped = (dcopy, prev)
    return out


def _act_epi(act):
    return {ACT_RELU: EPI_MUL_RELU_MASK, ACT_SWISH: EPI_MUL_DSWISH}[act]


# =================================================================================================
# Dropout (mask = hash(device step counter, salt, index); backward re-applies the same mask)
# =================================================================================================
class _NoGradCtx:
    """stands in for the autograd context when a Function's forward is called directly (inference)"""
    needs_input_grad = (False,) * 64

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def run(fn, *args):
    """fn.apply(*args) - or, with autograd off, fn.forward on a stand-in context: Function.apply costs ~5 us of host time per
    call, and a beam step of the decoder is ~45 such calls on a launch-bound path"""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    with ops.inference():
        return fn.forward(_NoGradCtx(), *args)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, salt):
        ctx.cfg = (p, salt)
        return ops.dropout(x.contiguous(), p, salt)

    @staticmethod
    def backward(ctx, dy):
        p, salt = ctx.cfg
        return ops.dropout(dy.contiguous(), p, salt), None, None


def dropout(x, p, salt, training):
    return DropoutFn.apply(x, p, salt) if (training and p > 0.0) else x


class MaskRowsFn(torch.autograd.Function):
    """x.masked_fill(~keep, 0) over the frames of a padded batch (rnn/encoders.py:323-325); keep: bool [B,T,1]"""

    @staticmethod
    def forward(ctx, x, keep):
        k8 = keep.reshape(-1).to(torch.uint8).contiguous()
        ctx.save_for_backward(k8)
        return ops.mask_rows(x.reshape(-1, x.shape[-1]).contiguous(), k8).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (k8,) = ctx.saved_tensors
        return ops.mask_rows(dy.reshape(-1, dy.shape[-1]).contiguous(), k8).view(dy.shape), None


# =================================================================================================
# LayerNorm  (reference: transformer/layer_norm.py:12-38)
# =================================================================================================
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, weight, bias, eps)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.pr = GradSink.use((weight, bias))
        ctx.shp = shp
        ctx.prev = _prev_drop(x)       # the block in front of a layer's final norm: its gradient dropout rides along
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        w, b = ctx.pr
        sink = GradSink([w, b])
        dy2 = dy.reshape(x2.shape).contiguous()
        dx = _ln_bwd_out(dy2, x2, w, mean, rstd, None, sink.buf(0), sink.buf(1), ctx.prev, ctx.shp)
        return (dx,) + sink.results() + (None,)


# =================================================================================================
# nn.Linear  (reference: torch.nn.Linear call sites; decoder.py:247 output_layer, ctc.py:26 ctc_lo)
# =================================================================================================
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        shp = x.shape
        x2 = ops.to_act(x.reshape(-1, shp[-1]).contiguous())
        y = ops.linear_fwd(x2, ops.wshadow(weight), bias)
        ctx.save_for_backward(x2)
        ctx.pr = GradSink.use((weight, bias))
        ctx.shp = shp
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        w, b = ctx.pr
        sink = GradSink([w, b] if b is not None else [w])
        dy2 = ops.to_act(dy.reshape(x2.shape[0], w.shape[0]).contiguous())
        dx = ops.linear_bwd_x(dy2, ops.wshadow(w)) if ctx.needs_input_grad[0] else None
        ops.linear_bwd_w(dy2, x2, sink.buf(0), db=sink.buf(1) if b is not None else None)
        res = sink.results()
        return (dx.view(ctx.shp) if dx is not None else None, res[0], res[1] if b is not None else None)


# =================================================================================================
# Position-wise feed-forward block with pre-LayerNorm and scaled residual
#   out = x + scale * (W2 act(W1 LN(x) + b1) + b2)
# reference: positionwise_feed_forward.py:12-32, conformer/encoder_layer.py:97-103,141-146,
#            transformer/encoder_layer.py / decoder_layer.py:123-128 (scale = 1, ReLU)
# =================================================================================================
import os as _os
FFN_FACTOR = _os.environ.get("EAMD_FFN_FACTOR", "1") != "0"   # tests flip this to keep the pre-activation and re-derive mask / activation derivative in backward


class FFNBlockFn(torch.autograd.Function):
    """drop = (p_inner, salt_inner, p_out, salt_out): dropout after the activation
    (positionwise_feed_forward.py:27) and on the block output before the residual add
    (conformer/encoder_layer.py:101,144); p = 0 keeps the fully fused path."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, scale, act, eps, drop=(0.0, 0, 0.0, 0)):
        shp = x.shape
        D = shp[-1]
        adt = ops.act_dtype()
        p_in, s_in, p_out, s_out = drop
        x2 = x.reshape(-1, D).contiguous()
        assert eps is not None or p_out <= 0.0
        fused = ops.fast()      # bf16-operand GEMMs carry the dropout masks in their epilogues
        h, zf = None, False
        one = eps is not None and (ops.fast() or (ops.f32_epilogue_drop() and not ops.f32_operand_drop()))
        ln_in = None
        if one and ops.FFN_LN_FUSED and x2.dtype == torch.float32 and D == 256:
            # the one-launch kernel below normalises its rows itself (eamd_ffn_t.ln_x): no LayerNorm launch, and the
            # normalised rows are written once (for the weight gradient) instead of written and read back
            xn = torch.empty(x2.shape, device=x2.device, dtype=adt)
            if ops.ffn_fused_ok(xn, ops.wshadow(w1), ops.wshadow(w2), act):
                mean = torch.empty(x2.shape[0], device=x2.device, dtype=torch.float32)
                rstd = torch.empty(x2.shape[0], device=x2.device, dtype=torch.float32)
                ln_in = (x2, ln_w, ln_b, eps, mean, rstd)
        if ln_in is None:
            xn, mean, rstd = _ln_fwd_in(x2, ln_w, ln_b, eps, adt)               # GEMM operand (bf16 in fast mode)
        if one and (ln_in is not None or ops.ffn_fused_ok(xn, ops.wshadow(w1), ops.wshadow(w2), act)):
            # fp32 mode, d = 256: BOTH products in one launch (csrc/ffn_f32.hip) - 32 rows per workgroup, the hidden units
            # never make a round trip for the second product; kept for backward: h and the factor f (no grad: neither)
            need = any(ctx.needs_input_grad[:7])          # (grad mode is off inside forward: needs_input_grad tells no-grad calls apart)
            packs = ops.ffn_pack(w1, w2)       # the four packed weight images of this layer (forward pair, backward pair)
            out, z, h = ops.ffn_fwd(xn, ops.wshadow(w1), b1, ops.wshadow(w2), b2, act=act, alpha=scale, R=x2,
                                    drop=(p_in, s_in, p_out, s_out), save=need, packed=packs[:2], ln=ln_in)
            ctx.packs = packs[2:]
            ctx.save_for_backward(x2, mean, rstd, xn, z, h)
            ctx.pr = GradSink.use((ln_w, ln_b, w1, b1, w2, b2))
            ctx.cfg = (scale, act, shp, drop)
            ctx.in_opd, ctx.zf, ctx.one_launch = False, True, True
            ctx.prev = _prev_drop(x)
            return _tag_out(out.view(shp), p_out, s_out)
        ctx.one_launch = False
        if act not in (ACT_RELU, ACT_SWISH):
            # hardtanh / tanh / selu (nets_utils.py:485-498): with inner dropout the dual-output epilogue below is generic
            # over eamd_act ids; without it the activation is its own pass (eamd_act_fwd / _bwd) - no staging-side variant
            if not (p_in > 0.0 and (fused or ops.f32_epilogue_drop())):
                z = ops.linear_fwd(xn, ops.wshadow(w1), b1)
                h = ops.act_fwd_any(z, act)
                if p_in > 0.0:
                    h = ops.dropout(h, p_in, s_in)
                hop = ops.to_act(h)
                if p_out > 0.0 and (fused or ops.f32_epilogue_drop()):
                    out = ops.linear_fwd(hop, ops.wshadow(w2), b2, R=x2 if eps is not None else None, alpha=scale, drop=(p_out, s_out))
                elif p_out > 0.0:
                    out = ops.axpby(x2, ops.dropout(ops.linear_fwd(hop, ops.wshadow(w2), b2), p_out, s_out), 1.0, scale)
                else:
                    out = ops.linear_fwd(hop, ops.wshadow(w2), b2, R=x2 if eps is not None else None, alpha=scale)
                ctx.save_for_backward(x2, mean, rstd, xn, z, hop)
                ctx.pr = GradSink.use((ln_w, ln_b, w1, b1, w2, b2))
                ctx.cfg = (scale, act, shp, drop)
                ctx.in_opd, ctx.zf, ctx.generic_act = False, False, True
                ctx.prev = _prev_drop(x)
                return _tag_out(out.view(shp), p_out, s_out)
        ctx.generic_act = False
        if p_in > 0.0 and (fused or (ops.f32_epilogue_drop() and not ops.f32_operand_drop())):
            # z and h = dropout(act(z)) from one launch (second output of the epilogue), in the operand dtype
            # ... and what is kept for backward is not z but the ready factor f = mask / (1 - p) * act'(z): the input-gradient
            # GEMM multiplies by it (EPI_MUL_AUX) instead of re-deriving the mask and the activation derivative per element
            h = torch.empty(x2.shape[0], w1.shape[0], device=x2.device, dtype=adt)
            z = ops.linear_fwd(xn, ops.wshadow(w1), b1, out_dtype=adt, drop=(p_in, s_in), Hb=h, h_act=act,
                               act=EPI_DACT_FACTOR if FFN_FACTOR else EPI_NONE)
            zf = FFN_FACTOR
        else:
            z = ops.linear_fwd(xn, ops.wshadow(w1), b1, out_dtype=adt)          # [M, F] pre-activation
            if p_in > 0.0 and not ops.f32_operand_drop():
                h = ops.dropout(z, p_in, s_in, act=act)                         # drop(act(z)), materialised
        src, a_act = (h, ACT_NONE) if h is not None else (z, act)
        # fp32 mode: h = drop(act(z)) is never written - the W2 product applies activation + mask while staging z
        a_drop = (p_in, s_in) if (h is None and p_in > 0.0) else None
        if p_out > 0.0 and (fused or ops.f32_epilogue_drop()):
            out = ops.linear_fwd(src, ops.wshadow(w2), b2, R=x2, alpha=scale, a_act=a_act, drop=(p_out, s_out), a_drop=a_drop)
        elif p_out > 0.0:
            br = ops.linear_fwd(src, ops.wshadow(w2), b2, a_act=a_act, a_drop=a_drop)
            out = ops.axpby(x2, ops.dropout(br, p_out, s_out), 1.0, scale)
        else:
            out = ops.linear_fwd(src, ops.wshadow(w2), b2, R=x2 if eps is not None else None, alpha=scale, a_act=a_act,
                                 a_drop=a_drop)
        ctx.save_for_backward(x2, mean, rstd, xn, z, h)
        ctx.pr = GradSink.use((ln_w, ln_b, w1, b1, w2, b2))
        ctx.cfg = (scale, act, shp, drop)
        ctx.in_opd = a_drop is not None
        ctx.zf = zf
        ctx.prev = _prev_drop(x)
        return _tag_out(out.view(shp), p_out, s_out)

    @staticmethod
    def backward(ctx, dout):
        x2, mean, rstd, xn, z, h = ctx.saved_tensors
        ln_w, ln_b, w1, b1, w2, b2 = ctx.pr
        scale, act, shp, (p_in, s_in, p_out, s_out) = ctx.cfg
        adt = ops.act_dtype()
        sink = GradSink(ctx.pr)
        do = dout.reshape(x2.shape).contiguous()
        # gradient of the branch output (dropout mask re-derived: as a bf16 copy, or - fp32 mode - inside the consuming GEMMs)
        dob, g_drop = _grad_operand(dout, do, p_out, s_out)
        inner = h is not None or ctx.in_opd          # the forward applied the inner dropout
        if h is not None:
            ops.linear_bwd_w(dob, h, sink.buf(4), alpha=scale, db=sink.buf(5), a_drop=g_drop)
        else:                                         # dW2 += s * drop(do)^T drop(act(z))
            ops.linear_bwd_w(dob, z, sink.buf(4), alpha=scale, b_act=act, db=sink.buf(5), a_drop=g_drop,
                             b_drop=(p_in, s_in) if ctx.in_opd else None)
        if getattr(ctx, "generic_act", False):
            dh = ops.linear_bwd_x(dob, ops.wshadow(w2), alpha=scale, a_drop=g_drop)
            if p_in > 0.0:
                dh = ops.dropout(dh, p_in, s_in)
            dz = ops.to_act(ops.act_bwd_any(dh, z, act))
            ops.linear_bwd_w(dz, xn, sink.buf(2), db=sink.buf(3))
            dxn = ops.linear_bwd_x(dz, ops.wshadow(w1))
            dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, shp)
            return (dx,) + sink.results() + (None, None, None, None)
        if ctx.one_launch and g_drop is None and dob.dtype == z.dtype:
            # dz = s (dob W2) (.) f and dxn = dz W1: one launch (bf16 operands: on the transposed weight copies)
            M, D = x2.shape
            if mean is not None and D == 256 and ops.ROWPROJ and ops.ffn_bwd_lnb_ok(M, w1.shape[0], dob.dtype):
                # ... and the LayerNorm backward of the block as the launch's epilogue (csrc/ln_bwd_rows.h)
                ws = ops.rowproj_lnb_ws(M, x2.device)
                prev = ctx.prev
                dcopy = torch.empty(M, D, device=x2.device, dtype=torch.float32) if prev is not None else None
                dz, dx = ops.ffn_bwd(dob, w1, w2, z, alpha=scale, packed=ctx.packs,
                                     lnb=(x2, ln_w, mean, rstd, do, ws, dcopy, prev))
                ops.ln_partials_reduce(ws, sink.buf(0), sink.buf(1), (M + 31) // 32, D)
                ops.linear_bwd_w(dz, xn, sink.buf(2), db=sink.buf(3))
                dx = dx.view(shp)
                if prev is not None:
                    dx._eamd_dropped = (dcopy, prev)
                return (dx,) + sink.results() + (None, None, None, None)
            dz, dxn = ops.ffn_bwd(dob, w1, w2, z, alpha=scale, packed=ctx.packs)
            ops.linear_bwd_w(dz, xn, sink.buf(2), db=sink.buf(3))
            dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, shp)
            return (dx,) + sink.results() + (None, None, None, None)
        if ctx.zf:       # z holds mask / (1 - p) * act'(z) already
            dz = ops.linear_bwd_x(dob, ops.wshadow(w2), epilogue=EPI_MUL_AUX, aux=z, alpha=scale, out_dtype=adt, a_drop=g_drop)
        elif inner and (ops.fast() or ops.f32_epilogue_drop()):
            dz = ops.linear_bwd_x(dob, ops.wshadow(w2), epilogue=_act_epi(act), aux=z, alpha=scale, out_dtype=adt,
                                  drop=(p_in, s_in), a_drop=g_drop)
        else:
            dz = ops.linear_bwd_x(dob, ops.wshadow(w2), epilogue=_act_epi(act), aux=z, alpha=scale, out_dtype=adt, a_drop=g_drop)
            if inner:
                dz = ops.dropout(dz, p_in, s_in)
        ops.linear_bwd_w(dz, xn, sink.buf(2), db=sink.buf(3))
        dxn = ops.linear_bwd_x(dz, ops.wshadow(w1))
        dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, shp)
        return (dx,) + sink.results() + (None, None, None, None)


class Conv1dFFNBlockFn(torch.autograd.Function):
    """x + scale * drop(W2 (*) drop(relu(W1 (*) LN(x)))) with (*) a conv1d along time, the positionwise variants
    MultiLayeredConv1d (both layers conv1d, k2 = k) and Conv1dLinear (second layer linear, k2 = 1).
    reference: transformer/multi_layer_conv.py:13-105 (always ReLU), encoder_layer / conformer block wiring as
    FFNBlockFn.  Each conv1d is im2col along time (eamd_unfold1d) + one GEMM against the tap-major [N, k*C] weight;
    drop = (p_inner, salt_inner, p_out, salt_out)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, scale, eps, drop=(0.0, 0, 0.0, 0)):
        B, T, D = x.shape
        H, k1 = w1.shape[0], w1.shape[2]
        k2 = w2.shape[2] if w2.dim() == 3 else 1
        adt = ops.act_dtype()
        p_in, s_in, p_out, s_out = drop
        x2 = x.reshape(-1, D).contiguous()
        xn, mean, rstd = ops.layernorm_fwd(x2, ln_w, ln_b, eps, adt)
        w1k = torch.empty(H, k1 * D, device=x.device, dtype=torch.float32)
        ops.permute4(w1.contiguous(), w1k, (H, D, k1, 1), (k1 * D, 1, D, 0))                 # [H, C, k] -> [H, k, C]
        w1k = ops.to_act(w1k)
        col1 = ops.unfold1d(xn, B, T, D, k1) if k1 > 1 else xn
        z = ops.linear_fwd(col1, w1k, b1)                                                     # fp32 pre-activation
        h = ops.dropout(z, p_in, s_in, act=ACT_RELU, out_dtype=adt)                           # p = 0: relu + cast
        if k2 > 1:
            w2k = torch.empty(D, k2 * H, device=x.device, dtype=torch.float32)
            ops.permute4(w2.contiguous(), w2k, (D, H, k2, 1), (k2 * H, 1, H, 0))
            w2k = ops.to_act(w2k)
            col2 = ops.unfold1d(h, B, T, H, k2)
        else:
            w2k, col2 = ops.wshadow(w2), h
        if p_out > 0.0:
            br = ops.linear_fwd(col2, w2k, b2)
            out = ops.axpby(x2, ops.dropout(br, p_out, s_out), 1.0, scale)
        else:
            out = ops.linear_fwd(col2, w2k, b2, R=x2, alpha=scale)
        ctx.save_for_backward(x2, mean, rstd, col1, z, col2, w1k, w2k)
        ctx.pr = GradSink.use((ln_w, ln_b, w1, b1, w2, b2))
        ctx.cfg = (scale, (B, T, D, H, k1, k2), drop)
        return out.view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        x2, mean, rstd, col1, z, col2, w1k, w2k = ctx.saved_tensors
        ln_w, ln_b, w1, b1, w2, b2 = ctx.pr
        scale, (B, T, D, H, k1, k2), (p_in, s_in, p_out, s_out) = ctx.cfg
        adt = ops.act_dtype()
        sink = GradSink(ctx.pr)
        do = dout.reshape(x2.shape).contiguous()
        dob = ops.dropout(do, p_out, s_out, out_dtype=adt) if p_out > 0.0 else ops.to_act(do)
        if k2 > 1:      # tap-major weight gradient, then back to the [D, H, k] parameter layout
            dw2k = torch.zeros(D, k2 * H, device=do.device, dtype=torch.float32)
            ops.linear_bwd_w(dob, col2, dw2k, alpha=scale, db=sink.buf(5))
            ops.permute4(dw2k, sink.buf(4), (D, k2, H, 1), (H * k2, 1, k2, 0), accumulate=True)
            dh = ops.fold1d(ops.linear_bwd_x(dob, w2k, alpha=scale), B, T, H, k2)
        else:
            ops.linear_bwd_w(dob, col2, sink.buf(4), alpha=scale, db=sink.buf(5))
            dh = ops.linear_bwd_x(dob, w2k, alpha=scale)
        if p_in > 0.0:
            dh = ops.dropout(dh, p_in, s_in)
        dz = ops.to_act(ops.act_bwd_any(dh, z, ACT_RELU))
        dw1k = torch.zeros(H, k1 * D, device=do.device, dtype=torch.float32)
        ops.linear_bwd_w(dz, col1, dw1k, db=sink.buf(3))
        ops.permute4(dw1k, sink.buf(2), (H, k1, D, 1), (D * k1, 1, k1, 0), accumulate=True)
        dcol = ops.linear_bwd_x(dz, w1k)
        dxn = ops.fold1d(dcol, B, T, D, k1) if k1 > 1 else dcol
        dx = ops.layernorm_bwd(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1))
        return (dx.view(B, T, D),) + sink.results() + (None, None, None)


# =================================================================================================
# Attention core on projected q/k/v laid out [B, T, H, dk] (i.e. the Linear outputs, untransposed)
# scores live in [H, B, T1, ldp] so that for one head the (b, i) rows are uniformly strided.
# =================================================================================================
def _ldp(T2):
    return (T2 + 7) // 8 * 8


class _MV:
    """[rows, D] matrix stored as a column block of a wider row-major tensor: element (r, c) at off + r*ld + c"""

    __slots__ = ("t", "off", "ld")

    def __init__(self, t, off=0, ld=None):
        self.t, self.off, self.ld = t, off, (t.shape[-1] if ld is None else ld)


def _mv(x):
    return x if isinstance(x, _MV) else _MV(x)


def attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk):
    D = H * dk
    ldp = _ldp(T2)
    qu, k = _mv(qu), _mv(k)
    ac = torch.empty(H * B * T1 * ldp, device=qu.t.device, dtype=torch.float32)
    ops.gemm(qu.t, k.t, ac, T1, T2, dk, qu.ld, k.ld, ldp, batch=(B, H), sA=(T1 * qu.ld, dk), sB=(T2 * k.ld, dk),
             sC=(T1 * ldp, B * T1 * ldp), a_off=qu.off, b_off=k.off)
    bd = None
    if p is not None:
        pm = _mv(p)
        bd = torch.empty_like(ac)
        ops.gemm(qv, pm.t, bd, T1, T2, dk, D, pm.ld, ldp, batch=(B, H), sA=(T1 * D, dk), sB=(0, dk),
                 sC=(T1 * ldp, B * T1 * ldp), b_off=pm.off)
    if ops.fast():
        P = torch.empty(H * B * T1 * ldp, device=ac.device, dtype=torch.bfloat16)
    else:
        P = ac   # in place
    ops.softmax_fwd(ac, bd, mask, P, H * B, B, T1, T2, ldp, 1.0 / math.sqrt(dk))
    return P


FUSE_ATTN = True   # tests flip this to compare against the GEMM / softmax / GEMM path
# attention blocks whose probabilities (P + dropped copy) exceed this many MB do not keep them for backward but run the fused
# forward again there (MHABlockFn): memory kept per block O(T') instead of O(T'^2).  < 0: never; 0: always.
ATTN_RECOMPUTE_MB = float(os.environ.get("EAMD_ATTN_RECOMPUTE_MB", "256"))
ATTN_TAP = None    # a list while E2E.calculate_all_attentions runs: every attention block appends its probabilities


def attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk, drop=None, shift_len=None):
    """(P, Pd, ctx) from eamd_attn_fwd, or None when the library declines the operands; drop = (p, salt): attention
    dropout inside the kernel (Pd = dropout(P) is what the context is built from; Pd is P without dropout)"""
    qu, k, v = _mv(qu), _mv(k), _mv(v)
    qv3 = p3 = None
    if p is not None:
        qvm, pm = _mv(qv), _mv(p)
        qv3, p3 = (qvm.t, qvm.off, qvm.ld), (pm.t, pm.off, pm.ld)
    return ops.attn_fwd((qu.t, qu.off, qu.ld), qv3, (k.t, k.off, k.ld), (v.t, v.off, v.ld), p3, mask, B, T1, T2, H, dk,
                        _ldp(T2), 1.0 / math.sqrt(dk), drop=drop, shift_len=shift_len)


def attn_context_fwd(P, v, B, T1, T2, H, dk):
    D = H * dk
    ldp = _ldp(T2)
    v = _mv(v)
    ctxv = torch.empty(B * T1, D, device=P.device, dtype=ops.act_dtype())
    ops.gemm(P, v.t, ctxv, T1, dk, T2, ldp, v.ld, D, transB=1, batch=(B, H), sA=(T1 * ldp, B * T1 * ldp),
             sB=(T2 * v.ld, dk), sC=(T1 * D, dk), b_off=v.off)
    return ctxv


def attn_core_bwd(dctx, P, qu, qv, k, v, p, B, T1, T2, H, dk, Pd=None, attn_drop=(0.0, 0), dqkv=None, dkv_out=None, shift_len=None,
                  dp_out=None):
    """returns dqu (fp32), dqv (fp32 or None), dk, dv (GEMM-operand dtype), dp (fp32 or None)
    Pd = dropped-out probabilities actually used for the context (None when attention dropout is off).
    dqkv (fused QKV projection, self-attention): a [B*T, 3D] operand-dtype buffer; dk / dv are written into its
    column blocks 1 / 2, and without relative positions dq goes straight into block 0 (dqu is then None).
    dkv_out (shared cross-attention projection): _MV of a [B*T2, *] operand-dtype buffer; dk / dv are written into
    its columns off .. off + D and off + D .. off + 2D.  dp_out (shared positional projection): _MV of a zeroed fp32
    [T2, *] buffer that receives dp (dp is then returned as None)."""
    D = H * dk
    ldp = _ldp(T2)
    dev = dctx.device
    adt = ops.act_dtype()
    qu, k, v = _mv(qu), _mv(k), _mv(v)
    sP = (T1 * ldp, B * T1 * ldp)
    if dkv_out is not None:
        assert dqkv is None
        ldo = dkv_out.ld
        dkk, dk_off, dv, dv_off = dkv_out.t, dkv_out.off, dkv_out.t, dkv_out.off + D
    elif dqkv is None:
        dv, dv_off, ldo = torch.empty(B * T2, D, device=dev, dtype=adt), 0, D
        dkk, dk_off = torch.empty(B * T2, D, device=dev, dtype=adt), 0
    else:
        ldo = 3 * D
        dv, dv_off, dkk, dk_off = dqkv, 2 * D, dqkv, D
    # the key-side products (dv, dk) are ONE launch behind the query side (eamd_attn_bwd_kv_f32 / eamd_attn_bwd_kv)
    kv_try = kv_fused = FUSE_ATTN and ops.FUSED_ATTN_KV and P.dtype == adt and dk == 64 and dctx.dtype == adt \
        and qu.t.dtype == adt
    if not kv_fused:
        ops.gemm(Pd if Pd is not None else P, dctx, dv, T2, dk, T1, ldp, D, ldo, transA=1, transB=1, batch=(B, H), sA=sP,
                 sB=(T1 * D, dk), sC=(T2 * ldo, dk), c_off=dv_off)                                        # P^T dctx
    dq_in_qkv = dqkv is not None and p is None     # no relative positions: dq is needed only as a GEMM operand
    dS = dbd = dqu = None
    if FUSE_ATTN and P.dtype == adt and ops.attn_fwd_supported(T1, T2, dk, p is not None):
        # dP, softmax backward (+ inverse rel-shift scatter) and dq in one launch
        dS = torch.empty(H * B * T1 * ldp, device=dev, dtype=adt)
        dbd = torch.empty(H * B * T1 * ldp, device=dev, dtype=adt) if p is not None else None
        dqu = None if dq_in_qkv else torch.empty(B * T1, D, device=dev, dtype=torch.float32)
        dq_view = (dqkv, 0, ldo) if dq_in_qkv else (dqu, 0, D)
        if not ops.attn_bwd_q((dctx, 0, D), (k.t, k.off, k.ld), (v.t, v.off, v.ld), P, dS, dbd, dq_view, B, T1, T2, H, dk,
                              ldp, 1.0 / math.sqrt(dk), drop=attn_drop if Pd is not None else None, shift_len=shift_len):
            dS = None
    if dS is None and shift_len is not None:
        raise ops._lib.EamdError("relative-position attention backward on a shape-bucketed batch needs the fused kernel")
    if dS is None:
        dP = torch.empty(H * B * T1 * ldp, device=dev, dtype=torch.float32)
        ops.gemm(dctx, v.t, dP, T1, T2, dk, D, v.ld, ldp, batch=(B, H), sA=(T1 * D, dk), sB=(T2 * v.ld, dk), sC=sP,
                 b_off=v.off)                                                                             # dctx v^T
        if Pd is not None:
            dP = ops.dropout(dP, attn_drop[0], attn_drop[1])      # same mask as the forward probabilities
        dbd = torch.empty(H * B * T1 * ldp, device=dev, dtype=adt) if p is not None else None   # fully written by softmax_bwd
        if ops.fast():
            dS = torch.empty(H * B * T1 * ldp, device=dev, dtype=torch.bfloat16)
            ops.softmax_bwd(P, dP, dbd, H * B, T1, T2, ldp, 1.0 / math.sqrt(dk), dS16=dS)
        else:
            ops.softmax_bwd(P, dP, dbd, H * B, T1, T2, ldp, 1.0 / math.sqrt(dk))   # dP <- d(ac)
            dS = dP
        if dq_in_qkv:
            dqu = None
            ops.gemm(dS, k.t, dqkv, T1, dk, T2, ldp, k.ld, ldo, transB=1, batch=(B, H), sA=sP, sB=(T2 * k.ld, dk),
                     sC=(T1 * ldo, dk), b_off=k.off)
        else:
            dqu = torch.empty(B * T1, D, device=dev, dtype=torch.float32)
            ops.gemm(dS, k.t, dqu, T1, dk, T2, ldp, k.ld, D, transB=1, batch=(B, H), sA=sP, sB=(T2 * k.ld, dk),
                     sC=(T1 * D, dk), b_off=k.off)
    dqv = dp = dpt = None
    if p is not None:
        pm = _mv(p)
        dqv = torch.empty(B * T1, D, device=dev, dtype=torch.float32)
        ops.gemm(dbd, pm.t, dqv, T1, dk, T2, ldp, pm.ld, D, transB=1, batch=(B, H), sA=sP, sB=(0, dk), sC=(T1 * D, dk),
                 b_off=pm.off)
        if dp_out is None:
            dp = torch.zeros(T2, D, device=dev, dtype=torch.float32)
            dpt, dp_off, ldd = dp, 0, D
        else:
            dpt, dp_off, ldd = dp_out.t, dp_out.off, dp_out.ld
    if kv_fused:
        # dv and dk in one launch.  (The kernel can also accumulate dpos += dbd^T qv - eamd_attn_bwd_kv_f32 with dbd / qv /
        # dpos set - but that product reduces over the BATCH too: 32-way contended atomics made it slower than the
        # split-K GEMM below, measured 133.6 vs 133.2 us for the whole backward; it is left to the GEMM.)
        kv_fused = ops.attn_bwd_kv(Pd if Pd is not None else P, dS, None, (dctx, 0, D), (qu.t, qu.off, qu.ld), None,
                                   (dv, dv_off, ldo), (dkk, dk_off, ldo), None, B, T1, T2, H, dk, ldp)
    if not kv_fused:
        if kv_try:       # the library declined the operands: dv was skipped above
            ops.gemm(Pd if Pd is not None else P, dctx, dv, T2, dk, T1, ldp, D, ldo, transA=1, transB=1, batch=(B, H), sA=sP,
                     sB=(T1 * D, dk), sC=(T2 * ldo, dk), c_off=dv_off)
        ops.gemm(dS, qu.t, dkk, T2, dk, T1, ldp, qu.ld, ldo, transA=1, transB=1, batch=(B, H), sA=sP, sB=(T1 * qu.ld, dk),
                 sC=(T2 * ldo, dk), b_off=qu.off, c_off=dk_off)
    if p is not None:
        # dp[j, h, :] = sum_{b,i} dbd[h, b, i, j] * qv[b, i, h, :]   (reduction over B*T1 rows, split-K)
        ops.gemm(dbd, qv, dpt, T2, dk, B * T1, ldp, D, ldd, transA=1, transB=1, batch=(1, H),
                 sA=(0, B * T1 * ldp), sB=(0, dk), sC=(0, dk), c_off=dp_off,
                 splitk=int(os.environ.get("EAMD_DPOS_SK", "0")) or max(2, ops.auto_splitk(T2, dk, B * T1) // 2))
        # (measured at config 2, 4 tiles x 4 heads: split-K 9 -> 26.7 us, 16 -> 20.7 us, 24 -> 22.4, 32 -> 23.9)
    return dqu, dqv, dkk, dv, dp


FUSE_QKV = True   # tests flip this to compare against the three-GEMM path


def _adjacent3(a, b, c):
    n = a.numel() * a.element_size()
    return (a.is_contiguous() and b.is_contiguous() and c.is_contiguous() and a.numel() == b.numel() == c.numel()
            and b.data_ptr() == a.data_ptr() + n and c.data_ptr() == b.data_ptr() + n)


def _qkv_adjacent(wq, wk, wv, bq, bk, bv):
    """True when the q/k/v projection weights (fp32 masters, bf16 shadows, gradient buffers) and biases lie back
    to back in their arenas (espnet_amd.train.FlatParams arranges that), so that the three Linear layers can run
    as one GEMM over a [3D, D] view.  The state_dict keeps the reference's separate linear_q / linear_k / linear_v."""
    if not FUSE_QKV or bq is None or not (_adjacent3(wq, wk, wv) and _adjacent3(bq, bk, bv)):
        return False
    gw = [getattr(t, "_eamd_grad", None) for t in (wq, wk, wv, bq, bk, bv)]
    if any(g is None for g in gw) or not (_adjacent3(*gw[:3]) and _adjacent3(*gw[3:])):
        return False
    return _adjacent3(ops.wshadow(wq), ops.wshadow(wk), ops.wshadow(wv))


def _span3(first, shape):
    """view over three adjacent equally sized tensors, starting at `first`"""
    stride = (shape[1], 1) if len(shape) == 2 else (1,)
    return first.detach().as_strided(shape, stride)


SHARE_PROJ = True   # tests flip this to reach the per-layer projections


def _adjacent_run(ts):
    """tensors lie back to back (equal sizes not required)"""
    return all(a.is_contiguous() and b.is_contiguous() and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()
               for a, b in zip(ts, ts[1:]))


def shared_proj_ok(weights, biases=None):
    """True when `weights` (and `biases`) - fp32 masters, operand shadows and gradient buffers - each form one run in
    their arenas (espnet_amd.train.FlatParams arranges that for the cross-attention k / v projections of a decoder
    stack and the linear_pos projections of an encoder stack): the projections then run as ONE GEMM (SharedProjFn)"""
    if not SHARE_PROJ or len(weights) < 2:
        return False
    for grp in (weights, biases):
        if grp is None:
            continue
        gr = [getattr(t, "_eamd_grad", None) for t in grp]
        if any(g is None for g in gr) or not (_adjacent_run(list(grp)) and _adjacent_run(gr)):
            return False
        if not _adjacent_run([ops.wshadow(t) for t in grp]):
            return False
    return True


class SharedProj:
    """Output of ONE projection GEMM whose column blocks feed several attention blocks, and the buffer that collects
    the blocks' gradients.  Block i = columns i * width .. (i + 1) * width."""

    def __init__(self, out, nblk, zero_grad, grad_dtype):
        self.out, self.n = out, nblk
        self.ld = out.shape[1]
        self.width = self.ld // nblk
        self.dout, self.pending, self.done = None, [], 0
        self.zero_grad, self.grad_dtype = zero_grad, grad_dtype

    def block(self, i, sub=0):
        return _MV(self.out, i * self.width + sub, self.ld)

    def grad_block(self, i):
        """column block i of the gradient buffer (allocated by the first block that reports in a backward pass)"""
        if self.dout is None:
            mk = torch.zeros if self.zero_grad else torch.empty
            self.dout = mk(self.out.shape, device=self.out.device, dtype=self.grad_dtype)
        self.pending.append(i)
        return _MV(self.dout, i * self.width, self.ld)


class SharedProjFn(torch.autograd.Function):
    """token = SharedProjFn(inp, box, zero_grad, nblk, n_w, *weights [, *biases]): out = inp W_all^T + b_all for the
    row-stacked weights of `nblk` consumers in one GEMM.  The result travels in a SharedProj appended to `box`, not
    through autograd: consumers read their column block, write their block of the gradient buffer in backward and
    take `token` as an input they return no gradient for - autograd then runs this backward once every consumer of
    the backward pass under way has reported (one pass, or one per phase of a phased data-parallel backward: only the
    blocks that reported are processed).  Weight / bias gradients go straight to the arena.
    reference: transformer/attention.py:94-114 (linear_k / linear_v), :164-206 (linear_pos) - per layer there."""

    @staticmethod
    def forward(ctx, inp, box, zero_grad, nblk, n_w, *params):
        ctx.set_materialize_grads(False)
        ws, bs = params[:n_w], params[n_w:]
        Din = ws[0].shape[1]
        rows_w = sum(w.shape[0] for w in ws)
        adt = ops.act_dtype()
        inp2 = ops.to_act_shared(inp).reshape(-1, Din)
        W = ops.wshadow(ws[0]).detach().as_strided((rows_w, Din), (Din, 1))
        b = bs[0].detach().as_strided((rows_w,), (1,)) if bs else None
        out = ops.linear_fwd(inp2, W, b, out_dtype=adt)
        sp = SharedProj(out, nblk, zero_grad, torch.float32 if zero_grad else adt)
        box.append(sp)
        ctx.sp, ctx.inp2, ctx.W, ctx.shape = sp, inp2, W, inp.shape      # plain attributes: survive a phased backward
        ctx.pr = GradSink.use(params)
        ctx.n_w = n_w
        return torch.zeros((), device=inp.device)

    @staticmethod
    def backward(ctx, _dtoken):
        sp, inp2, W = ctx.sp, ctx.inp2, ctx.W
        params = ctx.pr
        ws, bs = params[:ctx.n_w], params[ctx.n_w:]
        blocks = sorted(sp.pending)
        sp.pending = []
        dinp = None
        if blocks:
            assert blocks == list(range(blocks[0], blocks[-1] + 1)), "consumers of one backward pass are consecutive blocks"
            lo, hi = blocks[0] * sp.width, (blocks[-1] + 1) * sp.width
            M, Din, n = inp2.shape[0], inp2.shape[1], hi - lo
            dy = sp.dout if sp.dout.dtype == inp2.dtype else ops.to_act(sp.dout)
            rows_w = W.shape[0]
            dW = ws[0]._eamd_grad.as_strided((rows_w, Din), (Din, 1))[lo:hi]
            db = bs[0]._eamd_grad.as_strided((rows_w,), (1,))[lo:hi] if bs else None
            sk = ops.auto_splitk(n, Din, M)
            ops.gemm(dy, inp2, dW, n, Din, M, sp.ld, Din, Din, transA=1, transB=1, splitk=sk,
                     beta=1.0 if sk == 1 else 0.0, colsum=db, a_off=lo)            # dW += dY^T X, db += column sums
            if ctx.needs_input_grad[0]:
                dinp = torch.empty(M, Din, device=inp2.device, dtype=torch.float32)
                ops.gemm(dy, W, dinp, M, Din, n, sp.ld, Din, Din, transB=1, a_off=lo, b_off=lo * Din)   # dX = dY W
                dinp = dinp.view(ctx.shape)
            sp.done += len(blocks)
        if sp.done >= sp.n and GradSink.on_done is not None:
            ops.wgrad_join()
            GradSink.on_done(params)
        return (dinp, None, None, None, None) + (None,) * len(params)


class MHABlockFn(torch.autograd.Function):
    """out = x + Wo . Attention(LN(x) [, memory]) with optional legacy relative positions.

    reference: transformer/attention.py:16-114 (MultiHeadedAttention), :117-206
    (RelPositionMultiHeadedAttention), conformer/encoder_layer.py:106-129,
    transformer/decoder_layer.py:82-121 (self-attention and source attention with pre-norm).
    params: ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo [, wpos, pos_u, pos_v]
    """

    @staticmethod
    def forward(ctx, x, memory, pos_emb, mask, H, eps, last_query_only, drop, pre, token, *params):
        # drop = (p_attn, salt_attn, p_out, salt_out): attention.py:91 (dropout on the probabilities) and the
        # block-output dropout before the residual add (encoder_layer.py:126, decoder_layer.py:104,115)
        # pre = (kind, SharedProj, block index) with `token` from SharedProjFn: kind "kv" - k / v of the memory come
        # from the decoder stack's shared projection; kind "pos" - the projected positions from the encoder stack's
        p_att, s_att, p_out, s_out = drop
        pre_kv = pre if (pre is not None and pre[0] == "kv") else None
        pre_pos = pre if (pre is not None and pre[0] == "pos") else None
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = params[:10]
        rel = len(params) > 10
        B, T1f, D = x.shape
        dk = D // H
        adt = ops.act_dtype()
        x2 = x.reshape(-1, D).contiguous()
        fused = memory is None and not last_query_only and _qkv_adjacent(wq, wk, wv, bq, bk, bv)
        # row-block projections (csrc/rowproj_f32.hip; fp32 mode, D = 256, enough rows): LayerNorm + q/k/v as ONE launch, the
        # output projection with its dropout + residual as one, and in backward dctx = dy Wo and dxn = dqkv W3 + LayerNorm backward
        rp = None
        if fused and eps is not None and D == 256 and x2.dtype == torch.float32 and adt == torch.float32 \
                and ops.rowproj_ok(x2.shape[0], D, 3 * D) and (p_out <= 0.0 or ops.f32_epilogue_drop()):
            w3 = _span3(ops.wshadow(wq), (3 * D, D))
            rp = ops.rowproj_images(wq, [(w3, False), (w3, True), (wo.detach(), False), (wo.detach(), True)])
        if rp is not None:
            xn = torch.empty_like(x2)
            mean = torch.empty(x2.shape[0], device=x2.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
        else:
            xn, mean, rstd = _ln_fwd_in(x2, ln_w, ln_b, eps, adt)
        assert eps is not None or (p_out <= 0.0 and not last_query_only)
        if memory is None:
            kv_in, T2 = xn, T1f
        else:
            kv_in, T2 = ops.to_act_shared(memory).reshape(-1, D), memory.shape[1]
        if last_query_only:   # cached decoding: only the newest position queries (decoder_layer.py:88-101)
            xq = xn.view(B, T1f, D)[:, -1, :]            # row-strided views: ops.linear_fwd copies only if its kernel needs dense rows
            res = x2.view(B, T1f, D)[:, -1, :]
            if torch.is_grad_enabled():
                xq, res = xq.contiguous(), res.contiguous()
            T1 = 1
        else:
            xq, res, T1 = xn, (x2 if eps is not None else None), T1f
        if rp is not None:
            qkv = ops.rowproj(xn, rp[0], 3 * D, bias=_span3(bq, (3 * D,)), ln=(x2, ln_w, ln_b, eps, mean, rstd))
            q, k, v = _MV(qkv, 0, 3 * D), _MV(qkv, D, 3 * D), _MV(qkv, 2 * D, 3 * D)
        elif fused:
            # q, k, v weights sit back to back in the arenas (FlatParams): one [M, D] x [3D, D]^T projection
            w3, b3 = _span3(ops.wshadow(wq), (3 * D, D)), _span3(bq, (3 * D,))
            qkv = ops.linear_fwd(xq, w3, b3, out_dtype=adt)
            q, k, v = _MV(qkv, 0, 3 * D), _MV(qkv, D, 3 * D), _MV(qkv, 2 * D, 3 * D)
        else:
            qkv = None
            q = ops.linear_fwd(xq, ops.wshadow(wq), bq, out_dtype=adt)
            if pre_kv is not None:
                assert memory is not None
                k, v = pre_kv[1].block(pre_kv[2], 0), pre_kv[1].block(pre_kv[2], D)
            else:
                k = ops.linear_fwd(kv_in, ops.wshadow(wk), bk, out_dtype=adt)
                v = ops.linear_fwd(kv_in, ops.wshadow(wv), bv, out_dtype=adt)
        pos2 = None
        if rel:
            wpos, pu, pv = params[10:13]
            if pre_pos is not None:
                p = pre_pos[1].block(pre_pos[2])
            else:
                pos2 = ops.to_act_shared(pos_emb).reshape(-1, D)
                p = ops.linear_fwd(pos2, ops.wshadow(wpos), None, out_dtype=adt)
            if fused:
                qu, qv = ops.add_bias2(qkv, pu.reshape(-1), pv.reshape(-1), rows=B * T1, D=D, ldq=3 * D, q_off=0)
            else:
                qu, qv = ops.add_bias2(q, pu.reshape(-1), pv.reshape(-1))
        else:
            p, qu, qv = None, q, None
        fwd = None
        tb = ops.time_bound() if (rel and T1 == T2) else None      # padded batch: rel_shift over the batch's own length
        ctx.shift_len = tb
        if FUSE_ATTN and ops.attn_fwd_supported(T1, T2, dk, rel):
            # scores, softmax, attention dropout and context in one launch
            fwd = attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk, drop=(p_att, s_att) if p_att > 0.0 else None,
                                 shift_len=tb)
        if fwd is not None:
            P, Pd, cx = fwd
        else:
            if tb is not None:
                # a padded batch needs the rel_shift over its own length; only the fused kernels take it (shift_len)
                raise ops._lib.EamdError("relative-position attention on a shape-bucketed batch needs the fused attention kernels "
                                         "(d_k = 64, T' <= 4096: ops.attn_fwd_supported; train.BucketedGraphStep checks this per bucket up front); "
                                         "this shape runs the GEMM + softmax path")
            P = attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
            Pd = ops.dropout(P, p_att, s_att) if p_att > 0.0 else P
            cx = attn_context_fwd(Pd, v, B, T1, T2, H, dk)
        if ATTN_TAP is not None:      # calculate_all_attentions: the probabilities in the reference's (B, H, T1, T2) layout
            ATTN_TAP.append(Pd.view(H, B, T1, _ldp(T2))[..., :T2].permute(1, 0, 2, 3).float())
        if rp is not None:
            out = ops.rowproj(cx, rp[2], D, bias=bo, R=res, drop=(p_out, s_out) if p_out > 0.0 else None)
        elif p_out > 0.0 and (ops.fast() or ops.f32_epilogue_drop()):
            out = ops.linear_fwd(cx, ops.wshadow(wo), bo, R=res, drop=(p_out, s_out))
        elif p_out > 0.0:
            br = ops.linear_fwd(cx, ops.wshadow(wo), bo)
            out = ops.axpby(res, ops.dropout(br, p_out, s_out), 1.0, 1.0)
        else:
            out = ops.linear_fwd(cx, ops.wshadow(wo), bo, R=res)
        ctx.rp = rp
        # O(T') memory kept for backward: beyond ATTN_RECOMPUTE_MB of probabilities (P and its dropped copy: B x H x T1 x T2 each) they
        # are NOT kept - backward runs the fused forward kernel again (same operands, same dropout counter: the same bits) into
        # transient buffers.  One more attention-forward launch per block in backward (2 - 3 % of a step at T' = 249, where it is
        # off by default; at T' = 1000 P and Pd are 0.98 GB of the 2.57 GB a block keeps)
        ctx.recompute = None
        if fwd is not None and ATTN_RECOMPUTE_MB >= 0:
            held = P.numel() * P.element_size() * (2 if p_att > 0.0 else 1)
            if held >= ATTN_RECOMPUTE_MB * 2 ** 20:
                ctx.recompute = mask
                P = Pd = None
        # tensors that live in a SharedProj stay out of save_for_backward (they are no outputs of this Function)
        p_sv = None if pre_pos is not None else p
        if fused:   # k, v (and q without relative positions) are column blocks of qkv
            ctx.save_for_backward(x2, mean, rstd, xn, None, qu if rel else None, qv, qkv, None, p_sv, P, cx, pos2,
                                  Pd if (p_att > 0.0 and P is not None) else None)
        elif pre_kv is not None:
            ctx.save_for_backward(x2, mean, rstd, xn, None, qu, qv, None, None, p_sv, P, cx, pos2,
                                  Pd if (p_att > 0.0 and P is not None) else None)
        else:
            ctx.save_for_backward(x2, mean, rstd, xn, kv_in if memory is not None else None, qu, qv, k, v, p_sv, P, cx,
                                  pos2, Pd if (p_att > 0.0 and P is not None) else None)
        ctx.pre = pre
        if pre is not None:      # the shared projection owns these parameters' gradients
            skip = (4, 5, 6, 7) if pre_kv is not None else (10,)
            params = tuple(None if i in skip else q_ for i, q_ in enumerate(params))
        ctx.pr = GradSink.use(params)
        ctx.fused = fused
        ctx.cfg = (B, T1, T2, H, dk, D, rel, memory is not None, last_query_only, drop)
        ctx.prev = _prev_drop(x)
        return _tag_out(out.view(B, T1, D), p_out, s_out)

    @staticmethod
    def backward(ctx, dout):
        x2, mean, rstd, xn, mem2, qu, qv, k, v, p, P, cx, pos2, Pd = ctx.saved_tensors
        B, T1, T2, H, dk, D, rel, cross, last, (p_att, s_att, p_out, s_out) = ctx.cfg
        assert not last, "cached decoding path is inference-only"
        params = ctx.pr
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = params[:10]
        adt = ops.act_dtype()
        sink = GradSink(params)
        pre = ctx.pre
        pre_kv = pre if (pre is not None and pre[0] == "kv") else None
        dp_out = None
        if pre is not None and pre[0] == "pos":
            p, dp_out = pre[1].block(pre[2]), pre[1].grad_block(pre[2])
        do = dout.reshape(-1, D).contiguous()
        dob, g_drop = _grad_operand(dout, do, p_out, s_out)
        rp = ctx.rp if g_drop is None else None
        ops.linear_bwd_w(dob, cx, sink.buf(8), db=sink.buf(9), a_drop=g_drop)
        if rp is not None:
            dctx = ops.rowproj(dob, rp[3], D)
        else:
            dctx = ops.linear_bwd_x(dob, ops.wshadow(wo), out_dtype=adt, a_drop=g_drop)
        def again(qu_, qv_, k_, v_):
            """the probabilities this block did not keep: the fused forward once more (MHABlockFn.forward, ATTN_RECOMPUTE_MB)"""
            f = attn_fwd_fused(qu_, qv_, k_, v_, p, ctx.recompute, B, T1, T2, H, dk, drop=(p_att, s_att) if p_att > 0.0 else None,
                               shift_len=ctx.shift_len)
            if f is None:
                raise ops._lib.EamdError("attention recompute: the fused forward declined operands it took in forward")
            return f[0], (f[1] if p_att > 0.0 else None)
        if ctx.fused:
            qkv = k
            dqkv = torch.empty(B * T1, 3 * D, device=do.device, dtype=adt)
            quv = qu if rel else _MV(qkv, 0, 3 * D)
            if P is None:
                P, Pd = again(quv, qv, _MV(qkv, D, 3 * D), _MV(qkv, 2 * D, 3 * D))
            dqu, dqv, _, _, dp = attn_core_bwd(dctx, P, quv, qv, _MV(qkv, D, 3 * D), _MV(qkv, 2 * D, 3 * D), p, B, T1,
                                               T2, H, dk, Pd=Pd, attn_drop=(p_att, s_att), dqkv=dqkv, dp_out=dp_out,
                                               shift_len=ctx.shift_len)
            if rel:
                wpos = params[10]
                # dq = dqu + dqv (operand dtype, into the fused buffer) + both bias gradients in one pass
                ops.add_cast_colsum2(dqu, dqv, sink.buf(11).view(-1), sink.buf(12).view(-1), out=dqkv, out_off=0,
                                     ld_out=3 * D)
                if dp_out is None:
                    ops.linear_bwd_w(ops.to_act(dp), pos2, sink.buf(10))
            ops.linear_bwd_w(dqkv, xn, _span3(sink.buf(2), (3 * D, D)), db=_span3(sink.buf(3), (3 * D,)))
            if rp is not None:      # dxn = dqkv W3 and the LayerNorm backward (+ residual gradient, + the dropped copy) in one launch
                dx = _rp_ln_bwd(dqkv, rp[1], x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T1, D))
            else:
                dxn = ops.linear_bwd_x(dqkv, _span3(ops.wshadow(wq), (3 * D, D)))
                dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T1, D))
            return (dx, None, None, None, None, None, None, None, None, None) + sink.results()
        dkv_out = None
        if pre_kv is not None:
            k, v = pre_kv[1].block(pre_kv[2], 0), pre_kv[1].block(pre_kv[2], D)
            dkv_out = pre_kv[1].grad_block(pre_kv[2])
        if P is None:
            P, Pd = again(qu, qv, k, v)
        dqu, dqv, dkk, dv, dp = attn_core_bwd(dctx, P, qu, qv, k, v, p, B, T1, T2, H, dk,
                                              Pd=Pd, attn_drop=(p_att, s_att), dkv_out=dkv_out, dp_out=dp_out,
                                              shift_len=ctx.shift_len)
        if rel:
            wpos = params[10]
            dq = ops.add_cast_colsum2(dqu, dqv, sink.buf(11).view(-1), sink.buf(12).view(-1))
            if dp_out is None:
                ops.linear_bwd_w(ops.to_act(dp), pos2, sink.buf(10))
        else:
            dq = ops.to_act(dqu)
        kv_in = mem2 if cross else xn
        ops.linear_bwd_w(dq, xn, sink.buf(2), db=sink.buf(3))
        dxn = ops.linear_bwd_x(dq, ops.wshadow(wq))
        dmem = None
        if pre_kv is not None:     # dWk / dWv / dmemory: one GEMM each for the whole decoder stack (SharedProjFn.backward)
            dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T1, D))
            return (dx, None, None, None, None, None, None, None, None, None) + sink.results()
        ops.linear_bwd_w(dkk, kv_in, sink.buf(4), db=sink.buf(5))
        ops.linear_bwd_w(dv, kv_in, sink.buf(6), db=sink.buf(7))
        if cross:
            dmem = ops.linear_bwd_x(dkk, ops.wshadow(wk))
            ops.linear_bwd_x(dv, ops.wshadow(wv), out=dmem, beta=1.0)
            dmem = dmem.view(B, T2, D)
        else:
            ops.linear_bwd_x(dkk, ops.wshadow(wk), out=dxn, beta=1.0)
            ops.linear_bwd_x(dv, ops.wshadow(wv), out=dxn, beta=1.0)
        dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T1, D))
        return (dx, dmem, None, None, None, None, None, None, None, None) + sink.results()


# =================================================================================================
# Conformer convolution module with pre-LayerNorm and residual
# reference: conformer/convolution.py:13-79, conformer/encoder_layer.py:132-138
# params: ln_w, ln_b, pw1_w [2C,C,1], pw1_b, dw_w [C,1,K], dw_b, bn_w, bn_b, pw2_w [C,C,1], pw2_b
# buffers: running_mean, running_var (updated in training mode)
# =================================================================================================
FUSE_GLU_DWCONV = True   # tests flip this to reach the stand-alone GLU kernels


class ConvModuleBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, running_mean, running_var, training, act, eps, bn_eps, bn_momentum, drop, *params):
        p_out, s_out = drop          # block-output dropout (conformer/encoder_layer.py:136)
        ln_w, ln_b, w1, b1, wd, bd_, g, be, w2, b2 = params
        B, T, D = x.shape
        Cc = w2.shape[0]
        K = wd.shape[-1]
        M = B * T
        adt = ops.act_dtype()
        x2 = x.reshape(M, D).contiguous()
        rp = None
        if eps is not None and D == 256 and Cc == 256 and x2.dtype == torch.float32 and adt == torch.float32 \
                and ops.rowproj_ok(M, D, 2 * Cc) and (p_out <= 0.0 or ops.f32_epilogue_drop()):
            # row-block projections (csrc/rowproj_f32.hip): LayerNorm + pointwise conv 1 in one launch, pointwise conv 2 with
            # dropout + residual in one, and in backward de = dy W2 and dxn = da W1 + LayerNorm backward
            w1v, w2v = w1.detach().view(2 * Cc, D), w2.detach().view(Cc, Cc)
            rp = ops.rowproj_images(w1, [(w1v, False), (w1v, True), (w2v, False), (w2v, True)])
        if rp is not None:
            xn = torch.empty_like(x2)
            mean = torch.empty(M, device=x2.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            a = ops.rowproj(xn, rp[0], 2 * Cc, bias=b1, ln=(x2, ln_w, ln_b, eps, mean, rstd))
        else:
            xn, mean, rstd = _ln_fwd_in(x2, ln_w, ln_b, eps, adt)
            a = ops.linear_fwd(xn, ops.wshadow(w1).view(2 * Cc, D), b1)     # pointwise conv 1  [M, 2C] fp32
        assert eps is not None or p_out <= 0.0
        # a batch padded beyond its own longest utterance (ops.set_time_bound): frames from the bound on are zeroed in front of
        # the depthwise convolution (GLU(0, 0) = 0: the zero padding the reference has there) and left out of the BatchNorm
        tb = ops.time_bound()
        bound = (T, tb) if tb is not None else None
        if bound is not None:
            ops.mask_time(a, T, tb)
        # GLU + depthwise conv over time in one launch (GLU(a) formed while the conv loads its rows, never written)
        # (in training mode the same launch leaves the BatchNorm partial statistics of its output tiles behind).
        # The BatchNorm's num_batches_tracked buffer rides on running_mean (ConvolutionModule sets it): += 1 in the
        # same launch as the running-statistics update instead of a 1-element add kernel per layer
        gl = d = bmean = None
        nbt = getattr(running_mean, "_eamd_nbt", None)
        if FUSE_GLU_DWCONV:
            d = ops.dwconv_glu_fwd(a, wd.view(Cc, K), bd_, B, T, Cc, K,
                                   bn=(bn_eps, bn_momentum, running_mean, running_var, nbt) if (training and bound is None) else None)
            if d is not None and training and bound is None:
                d, bmean, brstd = d
        if d is None:
            gl = ops.glu_fwd(a, Cc)                                      # [M, C]
            d = ops.dwconv_fwd(gl, wd.view(Cc, K), bd_, B, T, Cc, K)
        if training:
            if bmean is None:
                bmean, brstd = ops.bn_stats(d, M, Cc, bn_eps, bn_momentum, running_mean, running_var, nbt, bound=bound)
        else:
            bmean = running_mean
            brstd = ops.axpby(running_var, None, 1.0, 0.0)
            brstd = torch.rsqrt_(brstd.add_(bn_eps))  # tiny [C] host-issued op on eval path only
        e = ops.bn_apply(d, bmean, brstd, g, be, M, Cc, act, adt)
        ctx.rp = rp
        if rp is not None:
            out = ops.rowproj(e, rp[2], Cc, bias=b2, R=x2, drop=(p_out, s_out) if p_out > 0.0 else None)
        elif p_out > 0.0 and (ops.fast() or ops.f32_epilogue_drop()):
            out = ops.linear_fwd(e, ops.wshadow(w2).view(Cc, Cc), b2, R=x2, drop=(p_out, s_out))
        elif p_out > 0.0:
            br = ops.linear_fwd(e, ops.wshadow(w2).view(Cc, Cc), b2)
            out = ops.axpby(x2, ops.dropout(br, p_out, s_out), 1.0, 1.0)
        else:
            out = ops.linear_fwd(e, ops.wshadow(w2).view(Cc, Cc), b2, R=x2 if eps is not None else None)
        ctx.save_for_backward(x2, mean, rstd, xn, a, gl, d, bmean, brstd, e)
        ctx.pr = GradSink.use(params)
        ctx.cfg = (B, T, D, Cc, K, act, training, drop)
        ctx.bound = bound
        ctx.prev = _prev_drop(x)
        return _tag_out(out.view(B, T, D), p_out, s_out)

    @staticmethod
    def backward(ctx, dout):
        x2, mean, rstd, xn, a, gl, d, bmean, brstd, e = ctx.saved_tensors
        ln_w, ln_b, w1, b1, wd, bd_, g, be, w2, b2 = ctx.pr
        B, T, D, Cc, K, act, training, (p_out, s_out) = ctx.cfg
        M = B * T
        adt = ops.act_dtype()
        sink = GradSink(ctx.pr)
        do = dout.reshape(M, D).contiguous()
        dob, g_drop = _grad_operand(dout, do, p_out, s_out)
        rp = ctx.rp if g_drop is None else None
        ops.linear_bwd_w(dob, e, sink.buf(8).view(Cc, Cc), db=sink.buf(9), a_drop=g_drop)
        if rp is not None:
            de = ops.rowproj(dob, rp[3], Cc)
        else:
            de = ops.linear_bwd_x(dob, ops.wshadow(w2).view(Cc, Cc), a_drop=g_drop)
        # (time bound: the bounded BatchNorm backward takes dy = 0 and writes dx = 0 from the bound on)
        dd = ops.bn_bwd(de, d, bmean, brstd, g, be, sink.buf(6), sink.buf(7), M, Cc, act, training, bound=ctx.bound)
        if gl is None:       # GLU fused into the depthwise kernels: its derivative in the input-gradient store, GLU(a) on load
            da = ops.dwconv_glu_bwd_x(dd, wd.view(Cc, K), a, B, T, Cc, K, adt)
            if ctx.bound is not None:
                ops.mask_time(da, T, ctx.bound[1])      # the forward zeroed these rows of a
            ops.dwconv_glu_bwd_w(dd, a, sink.buf(4).view(Cc, K), sink.buf(5), B, T, Cc, K)
        else:
            dgl = ops.dwconv_bwd_x(dd, wd.view(Cc, K), B, T, Cc, K)
            ops.dwconv_bwd_w(dd, gl, sink.buf(4).view(Cc, K), sink.buf(5), B, T, Cc, K)
            da = ops.glu_bwd(dgl, a, Cc, adt)
            if ctx.bound is not None:
                ops.mask_time(da, T, ctx.bound[1])
        ops.linear_bwd_w(da, xn, sink.buf(2).view(2 * Cc, D), db=sink.buf(3))
        if rp is not None and da.dtype == torch.float32:
            dx = _rp_ln_bwd(da, rp[1], x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T, D))
        else:
            dxn = ops.linear_bwd_x(da, ops.wshadow(w1).view(2 * Cc, D))
            dx = _ln_bwd_out(dxn, x2, ln_w, mean, rstd, do, sink.buf(0), sink.buf(1), ctx.prev, (B, T, D))
        return (dx, None, None, None, None, None, None, None, None) + sink.results()


# =================================================================================================
# Conv2dSubsampling (1/4 length): conv3x3 s2 + ReLU, conv3x3 s2 + ReLU, Linear, x * sqrt(d)
# reference: transformer/subsampling.py:14-59; embedding.py:80-91,143-161 (x * xscale)
# params: c1_w [C,1,3,3], c1_b, c2_w [C,C,3,3], c2_b, lin_w [D, C*W2], lin_b
# =================================================================================================
_TAPS_FWD = [(kh, kw) for kh in range(3) for kw in range(3)]
# stride-parity classes of the input gradient: (ph, pw) -> [(kh, kw)] in eamd_conv2_weight_prep order
_CLASSES = [((0, 0), [(0, 0), (0, 2), (2, 0), (2, 2)]), ((0, 1), [(0, 1), (2, 1)]),
            ((1, 0), [(1, 0), (1, 2)]), ((1, 1), [(1, 1)])]


def _conv3s2_fwd(y_in, w, b, B, Hi, Wi, Cc, adt):
    """3x3 stride-2 convolution + ReLU on an NHWC activation [B*Hi*Wi, Cc] as one implicit GEMM (rows gathered tap by
    tap) -> (y [B*Ho*Wo, Cc], input-gradient weight image wd, Ho, Wo)"""
    Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
    wf, wd = ops.conv2_weight_prep(w, adt)
    g = ops.make_gather(Cc, _TAPS_FWD, Ho, Wo, Hi, Wi, 2, 2)
    M = B * Ho * Wo
    y = torch.empty(M, Cc, device=y_in.device, dtype=adt)
    ops.gemm(y_in, wf, y, M, Cc, 9 * Cc, 9 * Cc, Cc, Cc, transB=1, bias=b, epilogue=EPI_RELU, gather=g)
    return y, wd, Ho, Wo


def _conv3s2_bwd(dy, y_in, wd, dw_buf, db_buf, B, Hi, Wi, Ho, Wo, Cc, adt):
    """dy [B*Ho*Wo, Cc] (already multiplied by this stage's ReLU mask) -> gradient w.r.t. y_in, multiplied by the ReLU
    mask of y_in in the epilogue; weight / bias gradients are accumulated into dw_buf / db_buf"""
    dev = dy.device
    M = B * Ho * Wo
    ops.colsum(dy, db_buf)
    # weight gradient: dwf[(tap, ci), co] = sum_pos col[pos, (tap, ci)] * dy[pos, co]
    dwf = torch.zeros(9 * Cc, Cc, device=dev, dtype=torch.float32)
    g = ops.make_gather(Cc, _TAPS_FWD, Ho, Wo, Hi, Wi, 2, 2)
    # bf16 operands: 64x64 tiles, 1024 workgroups.  fp32-MFMA kernel (measured at config 2, tools/gemm_breakdown.py fp32): the
    # 128x128 tile at just under two rounds of two workgroups per CU - 36 tiles x split-K 28 = 1008 workgroups - takes
    # 1.58 ms; 64x64 tiles x split-K 16 = 2304 workgroups 1.70 ms; 128x128 at 1.5 or 3 rounds 1.63-1.73 ms
    f32_big = (not ops.fast()) and Cc % 128 == 0
    tile = int(os.environ.get("EAMD_CONV_DW_TILE", "128" if f32_big else "64"))
    ntile = (9 * Cc // tile) * ((Cc + tile - 1) // tile)
    want = int(os.environ.get("EAMD_CONV_DW_WGS", "0")) or (1024 if ops.fast() else 1008 if f32_big else 2304)
    sk = max(2, min(64, (want + ntile - 1) // ntile, max(1, M // 256)))
    ops.gemm(y_in, dy, dwf, 9 * Cc, Cc, M, 9 * Cc, Cc, Cc, transA=1, transB=1, gather=g, splitk=sk, tile=tile)
    ops.conv2_weight_grad(dwf, dw_buf, Cc, Cc)
    # input gradient, one implicit GEMM per stride-parity class; the classes write disjoint rows of dy_in and leave together
    # (ops.gemm_multi: one launch in fp32 mode)
    dy_in = torch.empty(y_in.shape, device=dev, dtype=adt)
    q0, classes = 0, []
    for (ph, pw), taps in _CLASSES:
        Hc, Wc = (Hi - ph + 1) // 2, (Wi - pw + 1) // 2
        if Hc > 0 and Wc > 0:
            gt = ops.make_gather(Cc, [((ph - kh) // 2, (pw - kw) // 2) for kh, kw in taps], Hc, Wc, Ho, Wo, 1, 1)
            cm = ops.make_rowmap(Hc, Wc, Hi, Wi, 2, ph, 2, pw)
            nt = len(taps)
            ops.gemm(dy, wd, dy_in, B * Hc * Wc, Cc, nt * Cc, nt * Cc, Cc, Cc, transB=1, b_off=q0 * Cc * Cc,
                     gather=gt, cmap=cm, epilogue=EPI_MUL_RELU_MASK, aux=y_in, ldaux=Cc, defer=classes)
        q0 += len(taps)
    ops.gemm_multi(classes)
    return dy_in


def _taps(k):
    return [(kh, kw) for kh in range(k) for kw in range(k)]


def _conv_ks_fwd(y_in, w, b, B, Hi, Wi, Cc, adt, k, st):
    """general k x k stride-st convolution + ReLU on an NHWC activation (Conv2dSubsampling6's 5x5 stride 3): the gather
    descriptor holds 9 taps, so the k*k taps run as ceil(k*k/9) implicit GEMMs accumulating into an fp32 buffer."""
    Ho, Wo = (Hi - k) // st + 1, (Wi - k) // st + 1
    KK = k * k
    wf = torch.empty(KK, Cc, Cc, device=y_in.device, dtype=torch.float32)          # [tap][ci][co]
    ops.permute4(w.contiguous(), wf, (Cc, Cc, KK, 1), (1, Cc, Cc * Cc, 0))
    wd = torch.empty(KK, Cc, Cc, device=y_in.device, dtype=torch.float32)          # [tap][co][ci]
    ops.permute4(w.contiguous(), wd, (Cc, Cc, KK, 1), (Cc, 1, Cc * Cc, 0))
    wf, wd = ops.to_act(wf), ops.to_act(wd)
    M = B * Ho * Wo
    y32 = torch.empty(M, Cc, device=y_in.device, dtype=torch.float32)
    taps = _taps(k)
    for t0 in range(0, KK, 9):
        ch = taps[t0:t0 + 9]
        g = ops.make_gather(Cc, ch, Ho, Wo, Hi, Wi, st, st)
        ops.gemm(y_in, wf, y32, M, Cc, len(ch) * Cc, len(ch) * Cc, Cc, Cc, transB=1, bias=b if t0 == 0 else None,
                 beta=0.0 if t0 == 0 else 1.0, gather=g, b_off=t0 * Cc * Cc)
    y = ops.dropout(y32, 0.0, 0, act=ACT_RELU, out_dtype=adt)                       # p = 0: ReLU + cast
    return y, wd, Ho, Wo


def _conv_ks_bwd(dy, y_in, wd, dw_buf, db_buf, B, Hi, Wi, Ho, Wo, Cc, adt, k, st):
    dev = dy.device
    M = B * Ho * Wo
    KK = k * k
    taps = _taps(k)
    ops.colsum(dy, db_buf)
    dwf = torch.zeros(KK * Cc, Cc, device=dev, dtype=torch.float32)                 # [(tap, ci), co]
    for t0 in range(0, KK, 9):
        ch = taps[t0:t0 + 9]
        g = ops.make_gather(Cc, ch, Ho, Wo, Hi, Wi, st, st)
        ntile = (len(ch) * Cc // 64) * ((Cc + 63) // 64)
        sk = max(2, min(64, (1024 + ntile - 1) // ntile, max(1, M // 256)))
        ops.gemm(y_in, dy, dwf, len(ch) * Cc, Cc, M, len(ch) * Cc, Cc, Cc, transA=1, transB=1, gather=g, splitk=sk, tile=64,
                 c_off=t0 * Cc * Cc)
    ops.permute4(dwf, dw_buf, (KK, Cc, Cc, 1), (1, KK, Cc * KK, 0), accumulate=True)      # -> dw[co][ci][tap]
    # input gradient: one implicit GEMM per stride-parity class (ph, pw); its taps are kh = ph (mod st), kw = pw (mod st)
    dy_in = torch.empty(y_in.shape, device=dev, dtype=adt)
    order, classes = [], []
    for ph in range(st):
        for pw in range(st):
            ct = [(kh, kw) for kh in range(ph, k, st) for kw in range(pw, k, st)]
            classes.append(((ph, pw), ct))
            order += [kh * k + kw for kh, kw in ct]
    wd_cls = wd.index_select(0, torch.tensor(order, device=dev)).contiguous()          # class-ordered taps (data movement)
    q0, jobs = 0, []
    for (ph, pw), ct in classes:
        Hc, Wc = (Hi - ph + st - 1) // st, (Wi - pw + st - 1) // st
        if Hc > 0 and Wc > 0 and ct:
            gt = ops.make_gather(Cc, [((ph - kh) // st, (pw - kw) // st) for kh, kw in ct], Hc, Wc, Ho, Wo, 1, 1)
            cm = ops.make_rowmap(Hc, Wc, Hi, Wi, st, ph, st, pw)
            nt = len(ct)
            ops.gemm(dy, wd_cls, dy_in, B * Hc * Wc, Cc, nt * Cc, nt * Cc, Cc, Cc, transB=1, b_off=q0 * Cc * Cc,
                     gather=gt, cmap=cm, epilogue=EPI_MUL_RELU_MASK, aux=y_in, ldaux=Cc, defer=jobs)
        q0 += len(ct)
    ops.gemm_multi(jobs)
    return dy_in


class Conv2dSubsamplingFn(torch.autograd.Function):
    """Conv2d(1, C, 3, 2) -> ReLU -> n x [Conv2d(C, C, 3, 2) -> ReLU] -> Linear(C * W', D) * xscale.
    n = 1: Conv2dSubsampling (subsampling.py:17-66); n = 2: Conv2dSubsampling8 (:123-168); a 5x5 stride-3 second stage:
    Conv2dSubsampling6 (:69-120).  convs = (w, b) pairs of the C -> C stages; the stride of a stage follows from its
    kernel size (3 -> 2, 5 -> 3) as in the reference's three layouts."""

    @staticmethod
    def forward(ctx, x, xscale, c1_w, c1_b, lin_w, lin_b, *convs):
        B, T, F = x.shape
        Cc = c1_w.shape[0]
        D = lin_w.shape[0]
        H1, W1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
        adt = ops.act_dtype()
        x = x.contiguous()
        ys = [ops.conv1_fwd(x, c1_w, c1_b, B, T, F, Cc, adt)]               # [B,H1,W1,C] NHWC, ReLU'd
        dims, wds = [(H1, W1)], []
        for i in range(0, len(convs), 2):
            k = convs[i].shape[-1]
            if k == 3:
                y, wd, Ho, Wo = _conv3s2_fwd(ys[-1].view(-1, Cc), convs[i], convs[i + 1], B, dims[-1][0], dims[-1][1], Cc, adt)
            else:
                y, wd, Ho, Wo = _conv_ks_fwd(ys[-1].view(-1, Cc), convs[i], convs[i + 1], B, dims[-1][0], dims[-1][1], Cc,
                                             adt, k, (k + 1) // 2)
            ys.append(y)
            wds.append(wd)
            dims.append((Ho, Wo))
        Hl, Wl = dims[-1]
        assert lin_w.shape[1] == Cc * Wl
        # Linear over (c, f) features: our rows are (f, c)-ordered, so permute the weight columns
        wl = torch.empty(D, Wl * Cc, device=x.device, dtype=torch.float32)
        ops.permute4(lin_w, wl, (D, Cc, Wl, 1), (Wl * Cc, 1, Cc, 0))
        wl = ops.to_act(wl)
        out = ops.linear_fwd(ys[-1].view(B * Hl, Wl * Cc), wl, lin_b, alpha=xscale)
        ctx.save_for_backward(x, wl, *ys, *wds)
        ctx.pr = GradSink.use((c1_w, c1_b, lin_w, lin_b) + tuple(convs))
        ctx.cfg = (B, T, F, Cc, D, dims, xscale)
        ctx.ks = [convs[i].shape[-1] for i in range(0, len(convs), 2)]
        return out.view(B, Hl, D)

    @staticmethod
    def backward(ctx, dout):
        B, T, F, Cc, D, dims, xscale = ctx.cfg
        n = len(dims) - 1
        x, wl = ctx.saved_tensors[:2]
        ys = ctx.saved_tensors[2:3 + n]
        wds = ctx.saved_tensors[3 + n:]
        adt = ops.act_dtype()
        sink = GradSink(ctx.pr)
        dev = x.device
        Hl, Wl = dims[-1]
        dob = ops.to_act(dout.reshape(B * Hl, D).contiguous())
        ylv = ys[-1].view(B * Hl, Wl * Cc)
        # Linear: weight grad in permuted column order, then un-permute-accumulate
        dwl = torch.zeros(D, Wl * Cc, device=dev, dtype=torch.float32)
        ops.linear_bwd_w(dob, ylv, dwl, alpha=xscale, db=sink.buf(3))
        ops.permute4(dwl, sink.buf(2), (D, Wl, Cc, 1), (Wl * Cc, 1, Wl, 0), accumulate=True)
        dy = ops.linear_bwd_x(dob, wl, epilogue=EPI_MUL_RELU_MASK, aux=ylv, alpha=xscale, out_dtype=adt)
        dy = dy.view(B * Hl * Wl, Cc)
        for i in range(n - 1, -1, -1):
            (Hi, Wi), (Ho, Wo) = dims[i], dims[i + 1]
            if ctx.ks[i] == 3:
                dy = _conv3s2_bwd(dy, ys[i].view(-1, Cc), wds[i], sink.buf(4 + 2 * i), sink.buf(5 + 2 * i), B, Hi, Wi, Ho,
                                  Wo, Cc, adt)
            else:
                dy = _conv_ks_bwd(dy, ys[i].view(-1, Cc), wds[i], sink.buf(4 + 2 * i), sink.buf(5 + 2 * i), B, Hi, Wi, Ho,
                                  Wo, Cc, adt, ctx.ks[i], (ctx.ks[i] + 1) // 2)
        ops.conv1_bwd_w(dy, x, sink.buf(0), sink.buf(1), B, T, F, Cc)
        return (None, None) + sink.results()


# =================================================================================================
# Embedding + absolute positional encoding (decoder input layer)
# reference: decoder.py:83-86 (Embedding, PositionalEncoding), embedding.py:80-91
# =================================================================================================
class EmbedPEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, table, pe, scale, pos_offset):
        B, U = tokens.shape
        if U == 1 and not torch.is_grad_enabled():       # a beam step's newest tokens: read in place (one column of the prefix buffer)
            return ops.embed_pe(tokens, table, pe, U, scale, pos_offset).view(B, U, table.shape[1])
        tok = tokens.contiguous()
        out = ops.embed_pe(tok, table, pe, U, scale, pos_offset)
        ctx.save_for_backward(tok)
        ctx.pr = GradSink.use((table,))
        ctx.scale = scale
        return out.view(B, U, table.shape[1])

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        sink = GradSink(ctx.pr)
        ops.embed_bwd(tok, dout.reshape(-1, dout.shape[-1]).contiguous(), sink.buf(0), ctx.scale)
        return (None,) + sink.results() + (None, None, None)


class PosEncFn(torch.autograd.Function):
    """x * scale + pe[:T]  (PositionalEncoding on float inputs, embedding.py:80-91)"""

    @staticmethod
    def forward(ctx, x, pe, scale):
        B, T, D = x.shape
        ctx.scale = scale
        return ops.posenc(x.reshape(-1, D).contiguous(), pe, T, scale).view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        return ops.axpby(dout.contiguous(), None, ctx.scale, 0.0), None, None


class ScaledPosEncFn(torch.autograd.Function):
    """x * scale + alpha * pe[:T]  (ScaledPositionalEncoding, embedding.py:95-128; scale = 1 there)"""

    @staticmethod
    def forward(ctx, x, pe, alpha, scale):
        B, T, D = x.shape
        ctx.save_for_backward(pe)
        ctx.pr = GradSink.use((alpha,))
        ctx.cfg = (T, scale)
        return ops.posenc_scaled(x.reshape(-1, D).contiguous(), pe, alpha.reshape(1), T, scale).view(B, T, D)

    @staticmethod
    def backward(ctx, dout):
        (pe,) = ctx.saved_tensors
        T, scale = ctx.cfg
        sink = GradSink(ctx.pr)
        do = dout.contiguous()
        ops.posenc_scaled_bwd(do.view(-1, do.shape[-1]), pe, sink.buf(0).view(1), T)
        dx = do if scale == 1.0 else ops.axpby(do, None, scale, 0.0)
        res = sink.results()
        return dx, None, res[0], None


# =================================================================================================
# Losses: gradients are produced by the forward kernels; backward only rescales by the upstream
# scalar, read on the device (no host synchronisation).
# =================================================================================================
class CTCLossFn(torch.autograd.Function):
    """sum_b -log p(y_b | x_b) / B on raw activations [B, T, V].
    reference: ctc.py:53-66,67-123 (loss_fn + forward), espnet2/asr/ctc.py:44-108."""

    @staticmethod
    def forward(ctx, acts, ys_pad, hlens, blank, ignore_id, time_major=False):
        """time_major: acts is (T, B, V) as warp-ctc takes it; read in place (strides), no transposed copy"""
        B = acts.shape[1] if time_major else acts.shape[0]
        nll, grad = ops.ctc_loss(acts.contiguous(), ys_pad, hlens, blank, ignore_id, 1.0 / B,
                                 want_grad=acts.requires_grad, time_major=time_major)
        ctx.save_for_backward(grad)
        ctx.nll = nll
        return ops.reduce_sum(nll, 1.0 / B)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return ops.scale_dev(grad, g.contiguous(), 1.0, out=grad), None, None, None, None, None


class LabelSmoothingLossFn(torch.autograd.Function):
    """reference: transformer/label_smoothing_loss.py:44-63 (normalize_length=False => / batch)."""

    @staticmethod
    def forward(ctx, logits, target, smoothing, ignore_id, denom):
        V = logits.shape[-1]
        lg = logits.reshape(-1, V).contiguous()
        tg = target.reshape(-1).contiguous()
        loss_rows, correct, grad = ops.lsm_loss(lg, tg, smoothing, 1.0 / denom, ignore_id,
                                                want_grad=logits.requires_grad)
        ctx.save_for_backward(grad)
        ctx.shp = logits.shape
        ctx.mark_non_differentiable(correct)
        return ops.reduce_sum(loss_rows, 1.0 / denom), correct

    @staticmethod
    def backward(ctx, g, _gc):
        (grad,) = ctx.saved_tensors
        return ops.scale_dev(grad, g.contiguous(), 1.0, out=grad).view(ctx.shp), None, None, None, None


class ScaleFn(torch.autograd.Function):
    """a * x for a host constant a (rnn/decoders.py:272: loss *= mean(len(ys_in)) - 1)"""

    @staticmethod
    def forward(ctx, x, a):
        ctx.a = a
        return ops.axpby(x.reshape(-1).contiguous(), None, a, 0.0).view(x.shape)

    @staticmethod
    def backward(ctx, g):
        return ops.axpby(g.reshape(-1).contiguous(), None, ctx.a, 0.0).view(g.shape), None


class WeightedSumFn(torch.autograd.Function):
    """alpha * a + (1 - alpha) * b on 0-dim device scalars (e2e_asr_transformer.py:219-232)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.alpha = alpha
        return ops.axpby(a.reshape(1), b.reshape(1), alpha, 1.0 - alpha).reshape(())

    @staticmethod
    def backward(ctx, g):
        g1 = g.reshape(1).contiguous()
        return (ops.axpby(g1, None, ctx.alpha, 0.0).reshape(()),
                ops.axpby(g1, None, 1.0 - ctx.alpha, 0.0).reshape(()), None)
