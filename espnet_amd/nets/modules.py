"""Host-side mirror of the reference's nn building blocks (same class names, constructor
arguments, parameter names/shapes, so reference checkpoints load key-for-key) whose forward
passes run on the espnet_amd HIP kernels.

reference files mirrored (espnet/nets/pytorch_backend/...):
  transformer/{layer_norm,embedding,subsampling,attention,positionwise_feed_forward,
  encoder_layer,encoder,decoder_layer,decoder,label_smoothing_loss,mask,add_sos_eos}.py,
  conformer/{convolution,encoder_layer,encoder,swish}.py, ctc.py, nets_utils.py
torch.nn.Linear / Conv / BatchNorm objects below are parameter containers only (default init and
state_dict layout identical to the reference); their torch forward is never called.
"""
import math

import numpy as np
import torch

from .. import functional as F_
from .. import ops
from .. import rnn_functional as R_
from .scorer_interface import BatchScorerInterface

_EPS_LN = 1e-12


def _act_id(activation_type):
    if activation_type in ("swish", None) or isinstance(activation_type, Swish):
        return ops.ACT_SWISH
    if activation_type == "relu" or isinstance(activation_type, torch.nn.ReLU):
        return ops.ACT_RELU
    if activation_type == "tanh" or isinstance(activation_type, torch.nn.Tanh):
        return ops.ACT_TANH
    if activation_type == "hardtanh" or isinstance(activation_type, torch.nn.Hardtanh):
        if isinstance(activation_type, torch.nn.Hardtanh) and (activation_type.min_val, activation_type.max_val) != (-1.0, 1.0):
            raise NotImplementedError("Hardtanh: only the default range [-1, 1] (get_activation's) has a HIP kernel")
        return ops.ACT_HARDTANH
    if activation_type == "selu" or isinstance(activation_type, torch.nn.SELU):
        return ops.ACT_SELU
    raise NotImplementedError(f"activation {activation_type!r} has no HIP kernel (hardtanh, tanh, relu, selu, swish)")


def _p(module, p):
    """effective dropout probability of a module (0 in eval mode)"""
    return float(p) if (module.training and p > 0.0) else 0.0


class Swish(torch.nn.Module):
    """reference: conformer/swish.py:13-18"""

    def forward(self, x):
        y = torch.empty_like(x)
        ops._lib.check(ops._lib.lib().eamd_act_fwd(ops.ptr(x.contiguous()), ops.ptr(y), x.numel(), ops.ACT_SWISH,
                                                   ops.stream_ptr()), "eamd_act_fwd")
        return y


def get_activation(act):
    """reference: nets_utils.py:485-498"""
    # the returned module is what the reference hands to PositionwiseFeedForward / ConvolutionModule; those shells map it
    # to an eamd_act id (_act_id) for their fused blocks, so torch's own kernels never run on the path
    table = {"hardtanh": torch.nn.Hardtanh, "tanh": torch.nn.Tanh, "relu": torch.nn.ReLU, "selu": torch.nn.SELU, "swish": Swish}
    return table[act]()


class LayerNorm(torch.nn.LayerNorm):
    """reference: transformer/layer_norm.py:12-38 (eps 1e-12, last dim)"""

    def __init__(self, nout, dim=-1):
        super().__init__(nout, eps=_EPS_LN)
        assert dim == -1, "only last-dim LayerNorm is on the ASR path"

    def forward(self, x):
        return F_.run(F_.LayerNormFn, x, self.weight, self.bias, self.eps)


# ---- masks / padding helpers (integer work, host side like the reference) ----------------------
def make_pad_mask(lengths, maxlen=None):
    """reference: nets_utils.py:64-176 (length_dim=-1, xs=None case). True at padded positions."""
    if not isinstance(lengths, (list, tuple)):
        lengths = [int(v) for v in lengths.tolist()]
    maxlen = int(max(lengths)) if maxlen is None else maxlen
    ar = np.arange(maxlen)[None, :]
    return torch.from_numpy(ar >= np.asarray(lengths)[:, None])


def make_non_pad_mask(lengths, maxlen=None):
    """reference: nets_utils.py:179-265"""
    return ~make_pad_mask(lengths, maxlen)


def subsequent_mask(size, device="cpu", dtype=torch.bool):
    """reference: transformer/mask.py:20-38"""
    return torch.tril(torch.ones(size, size, device=device, dtype=dtype))


def target_mask(ys_in_pad, ignore_id):
    """reference: transformer/mask.py:41-51"""
    ys_mask = ys_in_pad != ignore_id
    m = subsequent_mask(ys_mask.size(-1), device=ys_mask.device).unsqueeze(0)
    return ys_mask.unsqueeze(-2) & m


def pad_list(xs, pad_value):
    """reference: nets_utils.py:34-61"""
    n = len(xs)
    maxlen = max(x.size(0) for x in xs)
    pad = xs[0].new_full((n, maxlen) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        pad[i, : x.size(0)] = x
    return pad


def subsampled_lengths(ilens, tmax=None, stages=2):
    """valid frames after the mask slicing of the conv2d input layers: `stages` = 2 -> [:-2:2] twice (subsampling.py:59,
    Conv2dSubsampling), 3 -> three times (:166, Conv2dSubsampling8), or an explicit list of (cut, step) slices, e.g.
    [(2, 2), (4, 3)] for Conv2dSubsampling6's [:-2:2][:-4:3] (:118)."""
    slices = [(2, 2)] * stages if isinstance(stages, int) else list(stages)
    out = []
    for n in ilens:
        n = int(n)
        t = n if tmax is None else tmax
        for cut, step in slices:
            n = max(0, -(-min(n, t - cut) // step))   # true entries of mask[:-cut:step]
            t = max(0, -(-(t - cut) // step))
        out.append(n)
    return out


def subsampled_stride(embed):
    """input frames per encoder frame of an input layer (product of the mask-slicing steps)"""
    if isinstance(embed, Conv2dSubsampling8):
        return 8
    if isinstance(embed, Conv2dSubsampling6):
        return 6
    if isinstance(embed, Conv2dSubsampling):
        return 4
    return 1


def embed_output_lengths(embed, ilens, tmax=None):
    """valid frames behind an encoder input layer: the conv2d family subsamples the mask, linear / embed keep it"""
    if isinstance(embed, Conv2dSubsampling8):
        return subsampled_lengths(ilens, tmax, 3)
    if isinstance(embed, Conv2dSubsampling6):
        return subsampled_lengths(ilens, tmax, [(2, 2), (4, 3)])
    if isinstance(embed, Conv2dSubsampling):
        return subsampled_lengths(ilens, tmax, 2)
    return [int(n) for n in ilens]


def _mask_u8(mask, device):
    """bool/uint8 mask of shape (B,1,T2) or (B,T1,T2) -> contiguous uint8 on device (None passes)."""
    if mask is None:
        return None
    m = mask.to(device=device, dtype=torch.uint8, non_blocking=True)
    return m.contiguous()


# ---- positional encodings -----------------------------------------------------------------------
class PositionalEncoding(torch.nn.Module):
    """reference: transformer/embedding.py:35-91.  Table is built on the host in fp32 exactly as the
    reference does and cached on the device; x*sqrt(d)+pe runs in eamd_posenc."""

    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model = d_model
        self.reverse = reverse
        self.xscale = math.sqrt(d_model)
        self.dropout_rate = dropout_rate
        self.pe = None
        self.salt, self.salt2 = ops.new_salt(), ops.new_salt()
        self.extend_pe(max_len, torch.device("cpu"))
        self._register_load_state_dict_pre_hook(self._pre_hook)

    @staticmethod
    def _pre_hook(state_dict, prefix, *args):
        state_dict.pop(prefix + "pe", None)   # embedding.py:14-32 (back-compat)

    def extend_pe(self, length, device):
        if self.pe is not None and self.pe.size(0) >= length:
            if self.pe.device != device:
                self.pe = self.pe.to(device)
            return
        pe = torch.zeros(length, self.d_model)
        if self.reverse:
            position = torch.arange(length - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
        else:
            position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32)
                             * -(math.log(10000.0) / self.d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.to(device)

    def forward(self, x):
        self.extend_pe(x.size(1), x.device)
        return F_.dropout(F_.PosEncFn.apply(x, self.pe, self.xscale), self.dropout_rate, self.salt, self.training)


class ScaledPositionalEncoding(PositionalEncoding):
    """reference: transformer/embedding.py:95-128: x + alpha * pe[:T] with a learnable scalar alpha (state_dict key
    `...alpha`); the input is NOT multiplied by sqrt(d) - xscale is 1 here, which is also what the subsampling
    Linear's epilogue applies when this class sits behind it."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model, dropout_rate, max_len)
        self.xscale = 1.0
        self.alpha = torch.nn.Parameter(torch.tensor(1.0))

    def reset_parameters(self):
        self.alpha.data = torch.tensor(1.0, device=self.alpha.device)

    def forward(self, x):
        self.extend_pe(x.size(1), x.device)
        return F_.dropout(F_.ScaledPosEncFn.apply(x, self.pe, self.alpha, 1.0), self.dropout_rate, self.salt, self.training)


class RelPositionalEncoding(PositionalEncoding):
    """reference: transformer/embedding.py:131-161 (legacy: reversed table of max_len, first T rows)."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model, dropout_rate, max_len, reverse=True)

    def pos_emb(self, T, device):
        self.extend_pe(T, device)
        return self.pe[:T]

    def forward(self, x):
        y = ops.axpby(x.contiguous(), None, self.xscale, 0.0)
        pos = self.pos_emb(x.size(1), x.device).unsqueeze(0)
        return (F_.dropout(y, self.dropout_rate, self.salt, self.training),
                F_.dropout(pos, self.dropout_rate, self.salt2, self.training))


# ---- input layers -------------------------------------------------------------------------------
class Conv2dSubsampling(torch.nn.Module):
    """reference: transformer/subsampling.py:14-59"""

    def __init__(self, idim, odim, dropout_rate, pos_enc=None):
        super().__init__()
        self.conv = torch.nn.Sequential(torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(),
                                        torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU())
        self.out = torch.nn.Sequential(
            torch.nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim),
            pos_enc if pos_enc is not None else PositionalEncoding(odim, dropout_rate))

    def _params(self):
        """(c1_w, c1_b, lin_w, lin_b, [w, b of every C -> C stage]).  The implicit-GEMM convolutions gather whole 64-channel
        pieces, so an odim that is not a multiple of 64 (subsampling.py:14-59 takes any) runs with its channel axis
        zero-padded to the next one: zero filters give ReLU(0) = 0 channels that meet zero weights downstream, and autograd
        slices the padded gradients back into the parameters."""
        c1_w, c1_b, lin_w, lin_b = self.conv[0].weight, self.conv[0].bias, self.out[0].weight, self.out[0].bias
        stages = [t for i in range(2, len(self.conv), 2) for t in (self.conv[i].weight, self.conv[i].bias)]
        C = c1_w.shape[0]
        pad = (-C) % 64
        if pad:
            pd = torch.nn.functional.pad
            c1_w, c1_b = pd(c1_w, (0, 0, 0, 0, 0, 0, 0, pad)), pd(c1_b, (0, pad))
            stages = [pd(t, (0, 0, 0, 0, 0, pad, 0, pad)) if t.dim() == 4 else pd(t, (0, pad)) for t in stages]
            D, CW = lin_w.shape
            lin_w = pd(lin_w.view(D, C, CW // C), (0, 0, 0, pad)).reshape(D, (C + pad) * (CW // C))      # columns are (c, f)
        return c1_w, c1_b, lin_w, lin_b, stages

    def forward(self, x, x_mask):
        pos = self.out[1]
        c1_w, c1_b, lin_w, lin_b, stages = self._params()
        y = F_.Conv2dSubsamplingFn.apply(x, pos.xscale, c1_w, c1_b, lin_w, lin_b, *stages)
        if isinstance(pos, RelPositionalEncoding):
            pe = pos.pos_emb(y.size(1), y.device).unsqueeze(0)
            y = (F_.dropout(y, pos.dropout_rate, pos.salt, pos.training),
                 F_.dropout(pe, pos.dropout_rate, pos.salt2, pos.training))
        else:
            pos.extend_pe(y.size(1), y.device)
            if isinstance(pos, ScaledPositionalEncoding):
                y = F_.ScaledPosEncFn.apply(y, pos.pe, pos.alpha, 1.0)
            else:
                y = F_.PosEncFn.apply(y, pos.pe, 1.0)   # x*xscale already applied in the Linear epilogue
            y = F_.dropout(y, pos.dropout_rate, pos.salt, pos.training)
        if x_mask is None:
            return y, None
        return y, x_mask[:, :, :-2:2][:, :, :-2:2].contiguous()


class Conv2dSubsampling8(Conv2dSubsampling):
    """reference: transformer/subsampling.py:123-168 (three 3x3 stride-2 convolutions, 1/8 length)"""

    def __init__(self, idim, odim, dropout_rate, pos_enc=None):
        torch.nn.Module.__init__(self)
        self.conv = torch.nn.Sequential(torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(),
                                        torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU(),
                                        torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU())
        self.out = torch.nn.Sequential(
            torch.nn.Linear(odim * ((((idim - 1) // 2 - 1) // 2 - 1) // 2), odim),
            pos_enc if pos_enc is not None else PositionalEncoding(odim, dropout_rate))

    def forward(self, x, x_mask):
        y, m = super().forward(x, None if x_mask is None else x_mask)
        if x_mask is None:
            return y, None
        return y, x_mask[:, :, :-2:2][:, :, :-2:2][:, :, :-2:2].contiguous()


class Conv2dSubsampling6(Conv2dSubsampling):
    """reference: transformer/subsampling.py:69-120 (3x3 stride 2, then 5x5 stride 3: 1/6 length)"""

    def __init__(self, idim, odim, dropout_rate, pos_enc=None):
        torch.nn.Module.__init__(self)
        self.conv = torch.nn.Sequential(torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(),
                                        torch.nn.Conv2d(odim, odim, 5, 3), torch.nn.ReLU())
        self.out = torch.nn.Sequential(
            torch.nn.Linear(odim * (((idim - 1) // 2 - 2) // 3), odim),
            pos_enc if pos_enc is not None else PositionalEncoding(odim, dropout_rate))

    def forward(self, x, x_mask):
        y, _ = super().forward(x, None)
        if x_mask is None:
            return y, None
        return y, x_mask[:, :, :-2:2][:, :, :-4:3].contiguous()


# ---- attention ---------------------------------------------------------------------------------
class MultiHeadedAttention(torch.nn.Module):
    """reference: transformer/attention.py:16-114.  Standalone forward = attention without the
    surrounding LayerNorm/residual (used by external callers); encoder/decoder layers call the
    fused block instead."""

    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k = n_feat // n_head
        self.h = n_head
        self.linear_q = torch.nn.Linear(n_feat, n_feat)
        self.linear_k = torch.nn.Linear(n_feat, n_feat)
        self.linear_v = torch.nn.Linear(n_feat, n_feat)
        self.linear_out = torch.nn.Linear(n_feat, n_feat)
        self.attn = None
        self.dropout_rate = dropout_rate
        self.salt_attn, self.salt_out = ops.new_salt(), ops.new_salt()

    def block_params(self):
        return (self.linear_q.weight, self.linear_q.bias, self.linear_k.weight, self.linear_k.bias,
                self.linear_v.weight, self.linear_v.bias, self.linear_out.weight, self.linear_out.bias)

    def _bare(self, query, key, value, pos_emb, mask):
        """the module on its own (attention.py:94-114 / :164-206): same kernels as the fused block, without the
        LayerNorm in front and the residual behind (eps = None)"""
        assert key is value or (key.shape == value.shape and key.data_ptr() == value.data_ptr()), \
            "key and value must be the same tensor (as at every call site of the reference)"
        memory = None if (key is query or key.data_ptr() == query.data_ptr()) else key.contiguous()
        drop = (_p(self, self.dropout_rate), self.salt_attn, 0.0, self.salt_out)
        n_tap = len(F_.ATTN_TAP) if F_.ATTN_TAP is not None else 0
        out = F_.MHABlockFn.apply(query.contiguous(), memory, pos_emb, _mask_u8(mask, query.device), self.h, None,
                                  False, drop, None, None, None, None, *self.block_params())
        if F_.ATTN_TAP is not None and len(F_.ATTN_TAP) > n_tap:
            self.attn = F_.ATTN_TAP[-1]
        return out

    def forward(self, query, key, value, mask):
        return self._bare(query, key, value, None, mask)


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    """reference: transformer/attention.py:117-206 (legacy rel_shift, zero_triu=False)"""

    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__(n_head, n_feat, dropout_rate)
        self.linear_pos = torch.nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = torch.nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = torch.nn.Parameter(torch.Tensor(self.h, self.d_k))
        torch.nn.init.xavier_uniform_(self.pos_bias_u)
        torch.nn.init.xavier_uniform_(self.pos_bias_v)

    def block_params(self):
        return super().block_params() + (self.linear_pos.weight, self.pos_bias_u, self.pos_bias_v)

    def forward(self, query, key, value, pos_emb, mask):
        return self._bare(query, key, value, pos_emb, mask)


def shared_stack_proj(kind, layers, attn_of, inp):
    """One projection GEMM for a whole layer stack (F_.SharedProjFn): kind "kv" = linear_k / linear_v of every
    decoder layer's source attention applied to the encoder memory, kind "pos" = linear_pos of every encoder layer's
    self-attention applied to the positional embedding.  Hands every layer its (kind, SharedProj, block, token) in
    `layer._pre`; returns False (nothing set) when the parameters do not form one run in the arenas."""
    atts = [attn_of(m) for m in layers]
    if kind == "kv":
        ws = [w for a in atts for w in (a.linear_k.weight, a.linear_v.weight)]
        bs = [b for a in atts for b in (a.linear_k.bias, a.linear_v.bias)]
    else:
        if not all(hasattr(a, "linear_pos") for a in atts):
            return False
        ws, bs = [a.linear_pos.weight for a in atts], None
    if not inp.is_cuda or not F_.shared_proj_ok(ws, bs):
        return False
    box = []
    token = F_.SharedProjFn.apply(inp, box, kind == "pos", len(atts), len(ws), *(ws + (bs or [])))
    for i, m in enumerate(layers):
        m._pre = (kind, box[0], i, token)
    return True


def mha_block(norm, attn, x, memory, pos_emb, mask, last_query_only=False, p_out=0.0, pre=None):
    """x + drop(attn(LN(x)[, memory])) through the fused HIP block (p_out = the layer's dropout rate).
    pre = (kind, SharedProj, block, token): this layer's share of a projection the whole stack ran as one GEMM."""
    drop = (_p(attn, attn.dropout_rate), attn.salt_attn, p_out if attn.training else 0.0, attn.salt_out)
    n_tap = len(F_.ATTN_TAP) if F_.ATTN_TAP is not None else 0
    out = F_.run(F_.MHABlockFn, x.contiguous(), memory, pos_emb, _mask_u8(mask, x.device), attn.h, norm.eps,
                              last_query_only, drop, pre[:3] if pre is not None else None,
                              pre[3] if pre is not None else None, norm.weight, norm.bias, *attn.block_params())
    if F_.ATTN_TAP is not None and len(F_.ATTN_TAP) > n_tap:
        attn.attn = F_.ATTN_TAP[-1]        # attention.py:90 (self.attn, kept for calculate_all_attentions / plotting)
    return out


class PositionwiseFeedForward(torch.nn.Module):
    """reference: transformer/positionwise_feed_forward.py:12-32"""

    def __init__(self, idim, hidden_units, dropout_rate, activation=None):
        super().__init__()
        self.w_1 = torch.nn.Linear(idim, hidden_units)
        self.w_2 = torch.nn.Linear(hidden_units, idim)
        self.dropout_rate = dropout_rate
        self.activation = activation if activation is not None else torch.nn.ReLU()
        self.act_id = _act_id(self.activation)
        self.salt_in, self.salt_out = ops.new_salt(), ops.new_salt()

    def forward(self, x):
        """the module on its own (positionwise_feed_forward.py:30-32): no LayerNorm in front, no residual behind"""
        drop = (_p(self, self.dropout_rate), self.salt_in, 0.0, self.salt_out)
        return F_.FFNBlockFn.apply(x.contiguous(), None, None, self.w_1.weight, self.w_1.bias, self.w_2.weight,
                                   self.w_2.bias, 1.0, self.act_id, None, drop)


class MultiLayeredConv1d(torch.nn.Module):
    """reference: transformer/multi_layer_conv.py:13-58 (Conv1d -> ReLU -> dropout -> Conv1d along time)"""

    def __init__(self, in_chans, hidden_chans, kernel_size, dropout_rate):
        super().__init__()
        self.w_1 = torch.nn.Conv1d(in_chans, hidden_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.w_2 = torch.nn.Conv1d(hidden_chans, in_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.dropout_rate = dropout_rate
        self.salt_in, self.salt_out = ops.new_salt(), ops.new_salt()
        if kernel_size % 2 == 0:
            raise NotImplementedError("even conv1d kernels change the sequence length; odd kernels are on the HIP path")


class Conv1dLinear(torch.nn.Module):
    """reference: transformer/multi_layer_conv.py:61-105 (Conv1d -> ReLU -> dropout -> Linear)"""

    def __init__(self, in_chans, hidden_chans, kernel_size, dropout_rate):
        super().__init__()
        self.w_1 = torch.nn.Conv1d(in_chans, hidden_chans, kernel_size, stride=1, padding=(kernel_size - 1) // 2)
        self.w_2 = torch.nn.Linear(hidden_chans, in_chans)
        self.dropout_rate = dropout_rate
        self.salt_in, self.salt_out = ops.new_salt(), ops.new_salt()
        if kernel_size % 2 == 0:
            raise NotImplementedError("even conv1d kernels change the sequence length; odd kernels are on the HIP path")


def positionwise_layer(layer_type, attention_dim, linear_units, dropout_rate, conv_kernel_size=1, activation=None):
    """reference: encoder.py:258-285 (get_positionwise_layer); the conv1d variants take no activation argument"""
    if layer_type == "linear":
        return PositionwiseFeedForward(attention_dim, linear_units, dropout_rate, activation)
    if layer_type == "conv1d":
        return MultiLayeredConv1d(attention_dim, linear_units, conv_kernel_size, dropout_rate)
    if layer_type == "conv1d-linear":
        return Conv1dLinear(attention_dim, linear_units, conv_kernel_size, dropout_rate)
    raise NotImplementedError("Support only linear or conv1d.")


def ffn_block(norm, ff, x, scale, p_out=0.0):
    drop = (_p(ff, ff.dropout_rate), ff.salt_in, p_out if ff.training else 0.0, ff.salt_out)
    if not isinstance(ff, PositionwiseFeedForward):
        return F_.Conv1dFFNBlockFn.apply(x.contiguous(), norm.weight, norm.bias, ff.w_1.weight, ff.w_1.bias,
                                         ff.w_2.weight, ff.w_2.bias, scale, norm.eps, drop)
    return F_.run(F_.FFNBlockFn, x.contiguous(), norm.weight, norm.bias, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight,
                               ff.w_2.bias, scale, ff.act_id, norm.eps, drop)


class ConvolutionModule(torch.nn.Module):
    """reference: conformer/convolution.py:13-79"""

    def __init__(self, channels, kernel_size, activation=None, bias=True):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.pointwise_conv1 = torch.nn.Conv1d(channels, 2 * channels, kernel_size=1, stride=1, padding=0, bias=bias)
        self.depthwise_conv = torch.nn.Conv1d(channels, channels, kernel_size, stride=1,
                                              padding=(kernel_size - 1) // 2, groups=channels, bias=bias)
        self.norm = torch.nn.BatchNorm1d(channels)
        self.pointwise_conv2 = torch.nn.Conv1d(channels, channels, kernel_size=1, stride=1, padding=0, bias=bias)
        self.activation = activation if activation is not None else torch.nn.ReLU()
        self.act_id = _act_id(self.activation)
        self.salt_out = ops.new_salt()
        assert bias, "bias=False variant is not on the path"

    def forward(self, x):
        """the module on its own (convolution.py:53-79): no LayerNorm in front, no residual behind"""
        bn = self.norm
        bn.running_mean._eamd_nbt = bn.num_batches_tracked      # incremented by the statistics kernel in training mode
        return F_.ConvModuleBlockFn.apply(
            x.contiguous(), bn.running_mean, bn.running_var, self.training, self.act_id, None, bn.eps, bn.momentum,
            (0.0, self.salt_out), None, None, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
            self.depthwise_conv.weight, self.depthwise_conv.bias, bn.weight, bn.bias, self.pointwise_conv2.weight,
            self.pointwise_conv2.bias)


def conv_block(norm, cm, x, p_out=0.0):
    bn = cm.norm
    bn.running_mean._eamd_nbt = bn.num_batches_tracked          # incremented by the statistics kernel in training mode
    return F_.ConvModuleBlockFn.apply(
        x.contiguous(), bn.running_mean, bn.running_var, cm.training, cm.act_id, norm.eps, bn.eps, bn.momentum,
        (p_out if cm.training else 0.0, cm.salt_out), norm.weight, norm.bias, cm.pointwise_conv1.weight, cm.pointwise_conv1.bias, cm.depthwise_conv.weight,
        cm.depthwise_conv.bias, bn.weight, bn.bias, cm.pointwise_conv2.weight, cm.pointwise_conv2.bias)


# ---- encoder layers ----------------------------------------------------------------------------
class ConformerEncoderLayer(torch.nn.Module):
    """reference: conformer/encoder_layer.py:16-157 (normalize_before=True, concat_after=False)"""

    def __init__(self, size, self_attn, feed_forward, feed_forward_macaron, conv_module, dropout_rate,
                 normalize_before=True, concat_after=False):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.feed_forward_macaron = feed_forward_macaron
        self.conv_module = conv_module
        if concat_after:
            self.concat_linear = torch.nn.Linear(size + size, size)
        self.salts = [ops.new_salt() for _ in range(4)]       # dropout sites of the composed (post-norm / concat) form
        self.norm_ff = LayerNorm(size)
        self.norm_mha = LayerNorm(size)
        if feed_forward_macaron is not None:
            self.norm_ff_macaron = LayerNorm(size)
            self.ff_scale = 0.5
        else:
            self.ff_scale = 1.0
        if self.conv_module is not None:
            self.norm_conv = LayerNorm(size)
            self.norm_final = LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.size = size
        self.normalize_before = normalize_before
        self.concat_after = concat_after

    _pre = None      # set by ConformerEncoder.forward for the duration of one pass (shared linear_pos of the stack)

    def forward(self, x_input, mask, cache=None):
        assert cache is None, "encoder-side cache is not used on the ASR path"
        p = self.dropout_rate
        if isinstance(x_input, tuple):
            x, pos_emb = x_input
        else:
            x, pos_emb = x_input, None
        if not self.normalize_before or self.concat_after:
            x = self._forward_composed(x, pos_emb, mask)
            return ((x, pos_emb), mask) if pos_emb is not None else (x, mask)
        if self.feed_forward_macaron is not None:
            x = ffn_block(self.norm_ff_macaron, self.feed_forward_macaron, x, self.ff_scale, p)
        x = mha_block(self.norm_mha, self.self_attn, x, None, pos_emb, mask, p_out=p, pre=self._pre)
        if self.conv_module is not None:
            x = conv_block(self.norm_conv, self.conv_module, x, p)
        x = ffn_block(self.norm_ff, self.feed_forward, x, self.ff_scale, p)
        if self.conv_module is not None:
            x = self.norm_final(x)
        if pos_emb is not None:
            return (x, pos_emb), mask
        return x, mask


    def _forward_composed(self, x, pos_emb, mask):
        """encoder_layer.py:99-157 with normalize_before=False (LayerNorm behind each residual sum) and / or concat_after
        (x + concat_linear([x, att(x)])): composed from the modules' own forwards - the fused blocks are the pre-norm form"""
        nb, tr, p = self.normalize_before, self.training, self.dropout_rate
        drop = lambda y, i: F_.dropout(y, p, self.salts[i], tr)  # noqa: E731
        if self.feed_forward_macaron is not None:
            x = x + self.ff_scale * drop(self.feed_forward_macaron(self.norm_ff_macaron(x) if nb else x), 0)
            x = x if nb else self.norm_ff_macaron(x)
        xn = self.norm_mha(x) if nb else x
        att = self.self_attn(xn, xn, xn, pos_emb, mask) if pos_emb is not None else self.self_attn(xn, xn, xn, mask)
        x = x + (self.concat_linear(torch.cat([xn, att], -1)) if self.concat_after else drop(att, 1))
        x = x if nb else self.norm_mha(x)
        if self.conv_module is not None:
            x = x + drop(self.conv_module(self.norm_conv(x) if nb else x), 2)
            x = x if nb else self.norm_conv(x)
        x = x + self.ff_scale * drop(self.feed_forward(self.norm_ff(x) if nb else x), 3)
        x = x if nb else self.norm_ff(x)
        if self.conv_module is not None:
            x = self.norm_final(x)
        return x


class TransformerEncoderLayer(torch.nn.Module):
    """reference: transformer/encoder_layer.py:16-101"""

    def __init__(self, size, self_attn, feed_forward, dropout_rate, normalize_before=True, concat_after=False):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.size = size
        self.normalize_before = normalize_before
        self.concat_after = concat_after
        if concat_after:
            self.concat_linear = torch.nn.Linear(size + size, size)
        self.salts = [ops.new_salt() for _ in range(2)]

    def _forward_composed(self, x, mask, cache):
        """encoder_layer.py:53-101 with normalize_before=False and / or concat_after, from the modules' own forwards"""
        nb, tr, p = self.normalize_before, self.training, self.dropout_rate
        res = x
        xn = self.norm1(x) if nb else x
        xq = xn
        if cache is not None:       # only the newest position queries (:70-77)
            xq, res, mask = xn[:, -1:, :], res[:, -1:, :], (None if mask is None else mask[:, -1:, :])
        att = self.self_attn(xq, xn, xn, mask)
        x = res + (self.concat_linear(torch.cat([xq, att], -1)) if self.concat_after else F_.dropout(att, p, self.salts[0], tr))
        x = x if nb else self.norm1(x)
        x = x + F_.dropout(self.feed_forward(self.norm2(x) if nb else x), p, self.salts[1], tr)
        x = x if nb else self.norm2(x)
        return torch.cat([cache, x], dim=1) if cache is not None else x

    def forward(self, x, mask, cache=None):
        p = self.dropout_rate
        if not self.normalize_before or self.concat_after:
            return self._forward_composed(x, mask, cache), mask
        if cache is None:
            x = mha_block(self.norm1, self.self_attn, x, None, None, mask, p_out=p)
        else:   # incremental scoring: only the newest position queries (encoder_layer.py:70-77)
            assert cache.shape == (x.shape[0], x.shape[1] - 1, self.size)
            mask = None if mask is None else mask[:, -1:, :]
            x = mha_block(self.norm1, self.self_attn, x, None, None, mask, last_query_only=True, p_out=p)
        x = ffn_block(self.norm2, self.feed_forward, x, 1.0, p)
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x, mask


class GradCuts:
    """Phased backward for data-parallel training (train.GraphedDataParallelStep): while `active` is a list, every cut
    point replaces the activation by a detached leaf and records (upstream tensor, leaf).  loss.backward() then stops
    at the last cut; upstream.backward(leaf.grad) continues segment by segment, so that the gradients of the layers
    already finished can be all-reduced while the rest of backward runs.  Values are unchanged."""

    active = None


def grad_cut(x):
    """x: tensor, or the (x, pos_emb) pair the relative-position layers pass along"""
    if GradCuts.active is None:
        return x
    t = x[0] if isinstance(x, tuple) else x
    if not (torch.is_tensor(t) and t.requires_grad):
        return x
    leaf = t.detach().requires_grad_(True)
    tag = getattr(t, "_eamd_out_drop", None)      # the next block's LayerNorm backward may fuse this block's dropout
    if tag is not None:
        leaf._eamd_out_drop = tag
    GradCuts.active.append((t, leaf))
    return (leaf,) + tuple(x[1:]) if isinstance(x, tuple) else leaf


class MultiSequential(torch.nn.ModuleList):
    """reference: transformer/repeat.py:12-33 (state_dict keys `N.<name>` like nn.Sequential)"""

    cut_before = ()      # layer indices in front of which a gradient cut is placed while GradCuts.active

    def forward(self, *args):
        for i, m in enumerate(self):
            if GradCuts.active is not None and i in self.cut_before:
                args = (grad_cut(args[0]),) + tuple(args[1:])
            args = m(*args)
        return args


def repeat(N, fn):
    return MultiSequential([fn(n) for n in range(N)])


class ConformerEncoder(torch.nn.Module):
    """reference: conformer/encoder.py:33-227"""

    def __init__(self, idim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 normalize_before=True, concat_after=False, positionwise_layer_type="linear",
                 positionwise_conv_kernel_size=1, macaron_style=False, pos_enc_layer_type="abs_pos",
                 selfattention_layer_type="selfattn", activation_type="swish", use_cnn_module=False,
                 cnn_module_kernel=31, padding_idx=-1):
        super().__init__()
        if pos_enc_layer_type == "abs_pos":
            pos_enc_class = PositionalEncoding
        elif pos_enc_layer_type == "scaled_abs_pos":       # conformer/encoder.py:98-99
            pos_enc_class = ScaledPositionalEncoding
        elif pos_enc_layer_type == "rel_pos":
            assert selfattention_layer_type == "rel_selfattn"
            pos_enc_class = RelPositionalEncoding
        else:
            raise NotImplementedError("pos_enc_layer_type " + pos_enc_layer_type)
        if input_layer not in ("conv2d", "conv2d6", "conv2d8"):
            raise NotImplementedError("input_layer=%r: conv2d / conv2d6 / conv2d8 are on the HIP path" % (input_layer,))
        self.embed = {"conv2d": Conv2dSubsampling, "conv2d6": Conv2dSubsampling6, "conv2d8": Conv2dSubsampling8}[input_layer](
            idim, attention_dim, dropout_rate, pos_enc_class(attention_dim, positional_dropout_rate))

        def pw():
            return positionwise_layer(positionwise_layer_type, attention_dim, linear_units, dropout_rate,
                                      positionwise_conv_kernel_size, get_activation(activation_type))
        self.normalize_before = normalize_before
        if selfattention_layer_type == "selfattn":
            attn_class = MultiHeadedAttention
        elif selfattention_layer_type == "rel_selfattn":
            assert pos_enc_layer_type == "rel_pos"
            attn_class = RelPositionMultiHeadedAttention
        else:
            raise NotImplementedError("selfattention_layer_type " + selfattention_layer_type)
        self.encoders = repeat(
            num_blocks,
            lambda lnum: ConformerEncoderLayer(
                attention_dim,
                attn_class(attention_heads, attention_dim, attention_dropout_rate),
                pw(),
                pw()
                if macaron_style else None,
                ConvolutionModule(attention_dim, cnn_module_kernel, get_activation(activation_type))
                if use_cnn_module else None,
                dropout_rate, normalize_before, concat_after))
        if self.normalize_before:
            self.after_norm = LayerNorm(attention_dim)

    def forward(self, xs, masks):
        xs, masks = self.embed(xs, masks)
        # relative positions: linear_pos of all layers on the one positional embedding as one [T, D*layers] GEMM
        shared = isinstance(xs, tuple) and shared_stack_proj("pos", list(self.encoders), lambda m: m.self_attn, xs[1])
        x0 = xs[0] if isinstance(xs, tuple) else xs
        self._rowproj_prepack(x0.shape[0] * x0.shape[1], x0.shape[2])
        try:
            xs, masks = self.encoders(xs, masks)
        finally:
            ops.rowproj_prepack_end()
            if shared:
                for m in self.encoders:
                    m._pre = None
        if isinstance(xs, tuple):
            xs = xs[0]
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, masks


def _conformer_rowproj_prepack(self, M, D):
    """the packed weight images of ALL layers' row-block projections (q/k/v, attention output, both pointwise convolutions;
    csrc/rowproj_f32.hip) in two launches at the start of the pass instead of two per layer; the blocks pick them up through
    ops.rowproj_images"""
    if not ops.rowproj_ok(M, D, D) or D != 256:
        return
    ffns = [f for m in self.encoders for f in (m.feed_forward_macaron, m.feed_forward)
            if isinstance(f, PositionwiseFeedForward) and ops.FUSED_FFN and M >= ops.FUSED_FFN_MIN_ROWS]
    groups = []
    for m in self.encoders:
        att, cv = m.self_attn, m.conv_module
        if F_._qkv_adjacent(att.linear_q.weight, att.linear_k.weight, att.linear_v.weight, att.linear_q.bias, att.linear_k.bias,
                            att.linear_v.bias):
            w3 = F_._span3(ops.wshadow(att.linear_q.weight), (3 * D, D))
            wo = att.linear_out.weight.detach()
            groups.append((att.linear_q.weight, [(w3, False), (w3, True), (wo, False), (wo, True)]))
        if cv is not None and cv.pointwise_conv2.weight.shape[0] == D:
            w1 = cv.pointwise_conv1.weight.detach().view(2 * D, D)
            w2 = cv.pointwise_conv2.weight.detach().view(D, D)
            groups.append((cv.pointwise_conv1.weight, [(w1, False), (w1, True), (w2, False), (w2, True)]))
    ops.rowproj_prepack(groups)
    ops.ffn_prepack([(f.w_1.weight, f.w_2.weight) for f in ffns])        # ... and the fused feed-forward blocks' images in one more


ConformerEncoder._rowproj_prepack = _conformer_rowproj_prepack


class _LinearInput(torch.nn.Sequential):
    """input_layer="linear": Linear -> LayerNorm -> Dropout -> ReLU -> pos_enc (encoder.py:95-102); the
    children keep the reference's indices so the state_dict keys are embed.0.* / embed.1.*"""

    def __init__(self, idim, odim, dropout_rate, pos_enc):
        super().__init__(torch.nn.Linear(idim, odim), LayerNorm(odim), torch.nn.Dropout(dropout_rate),
                         torch.nn.ReLU(), pos_enc)
        self.salt = ops.new_salt()

    def forward(self, x):
        y = F_.LinearFn.apply(x, self[0].weight, self[0].bias)
        y = F_.dropout(self[1](y), self[2].p, self.salt, self.training)
        y = R_.ActFn.apply(y, ops.ACT_RELU)
        return self[4](y)


class _EmbedInput(torch.nn.Sequential):
    """input_layer="embed": Embedding -> pos_enc (encoder.py:123-127)"""

    def __init__(self, idim, odim, padding_idx, pos_enc):
        super().__init__(torch.nn.Embedding(idim, odim, padding_idx=padding_idx), pos_enc)

    def forward(self, tokens):
        pad = self[0].padding_idx
        return self[1](R_.PlainEmbedFn.apply(tokens, self[0].weight, -1 if pad is None else pad))


class TransformerEncoder(torch.nn.Module):
    """reference: transformer/encoder.py:48-332 (selfattn + linear positionwise; conv2d / linear / embed
    input layers; `forward_one_step` with per-layer caches for language-model scoring)"""

    def __init__(self, idim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 pos_enc_class=PositionalEncoding, normalize_before=True, concat_after=False,
                 positionwise_layer_type="linear", positionwise_conv_kernel_size=1, padding_idx=-1, **unused):
        super().__init__()
        pos = pos_enc_class(attention_dim, positional_dropout_rate)
        if input_layer == "conv2d":
            self.embed = Conv2dSubsampling(idim, attention_dim, dropout_rate, pos)
        elif input_layer == "conv2d8":
            self.embed = Conv2dSubsampling8(idim, attention_dim, dropout_rate, pos)
        elif input_layer == "conv2d6":
            self.embed = Conv2dSubsampling6(idim, attention_dim, dropout_rate, pos)
        elif input_layer == "linear":
            self.embed = _LinearInput(idim, attention_dim, dropout_rate, pos)
        elif input_layer == "embed":
            self.embed = _EmbedInput(idim, attention_dim, padding_idx, pos)
        else:
            raise NotImplementedError("input_layer %r: conv2d / conv2d6 / conv2d8 / linear / embed are on the HIP path" % (input_layer,))
        self.normalize_before = normalize_before
        self.encoders = repeat(
            num_blocks,
            lambda lnum: TransformerEncoderLayer(
                attention_dim, MultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate),
                positionwise_layer(positionwise_layer_type, attention_dim, linear_units, dropout_rate,
                                   positionwise_conv_kernel_size), dropout_rate,
                normalize_before, concat_after))
        if self.normalize_before:
            self.after_norm = LayerNorm(attention_dim)

    def _embed(self, xs, masks):
        if isinstance(self.embed, Conv2dSubsampling):
            return self.embed(xs, masks)
        return self.embed(xs), masks

    def forward(self, xs, masks):
        xs, masks = self._embed(xs, masks)
        xs, masks = self.encoders(xs, masks)
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, masks

    def forward_one_step(self, xs, masks, cache=None):
        """reference: encoder.py:306-332 -> (ys [B, L, D], masks, new per-layer caches)"""
        xs, masks = self._embed(xs, masks)
        if cache is None:
            cache = [None] * len(self.encoders)
        new_cache = []
        for c, e in zip(cache, self.encoders):
            xs, masks = e(xs, masks, cache=c)
            new_cache.append(xs)
        if self.normalize_before:
            xs = self.after_norm(xs.contiguous())
        return xs, masks, new_cache


# ---- decoder ------------------------------------------------------------------------------------
class DecoderLayer(torch.nn.Module):
    """reference: transformer/decoder_layer.py:15-134"""

    def __init__(self, size, self_attn, src_attn, feed_forward, dropout_rate, normalize_before=True,
                 concat_after=False):
        super().__init__()
        self.size = size
        if concat_after:
            self.concat_linear1 = torch.nn.Linear(size + size, size)
            self.concat_linear2 = torch.nn.Linear(size + size, size)
        self.salts = [ops.new_salt() for _ in range(3)]
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)
        self.norm3 = LayerNorm(size)
        self.dropout_rate = dropout_rate
        self.normalize_before = normalize_before
        self.concat_after = concat_after

    _pre = None      # set by Decoder.forward for the duration of one pass (shared k / v projection of the stack)

    def _forward_composed(self, tgt, tgt_mask, memory, memory_mask, cache):
        """decoder_layer.py:60-134 with normalize_before=False and / or concat_after, from the modules' own forwards"""
        nb, tr, p = self.normalize_before, self.training, self.dropout_rate
        drop = lambda y, i: F_.dropout(y, p, self.salts[i], tr)  # noqa: E731
        res = tgt
        xn = self.norm1(tgt) if nb else tgt
        xq, q_mask = xn, tgt_mask
        if cache is not None:       # only the newest position queries (:81-93)
            xq, res, q_mask = xn[:, -1:, :], res[:, -1:, :], (None if tgt_mask is None else tgt_mask[:, -1:, :])
        att = self.self_attn(xq, xn, xn, q_mask)
        x = res + (self.concat_linear1(torch.cat([xq, att], -1)) if self.concat_after else drop(att, 0))
        x = x if nb else self.norm1(x)
        xn = self.norm2(x) if nb else x
        if memory.shape[0] != x.shape[0]:       # beam search on the memory of G utterances: one copy per hypothesis
            g = x.shape[0] // memory.shape[0]
            memory_x = memory.repeat_interleave(g, 0)
            mmask_x = None if memory_mask is None else memory_mask.repeat_interleave(g, 0)
        else:
            memory_x, mmask_x = memory, memory_mask
        att = self.src_attn(xn, memory_x, memory_x, mmask_x)
        x = x + (self.concat_linear2(torch.cat([xn, att], -1)) if self.concat_after else drop(att, 1))
        x = x if nb else self.norm2(x)
        x = x + drop(self.feed_forward(self.norm3(x) if nb else x), 2)
        x = x if nb else self.norm3(x)
        return torch.cat([cache, x], dim=1) if cache is not None else x

    def forward(self, tgt, tgt_mask, memory, memory_mask, cache=None):
        p = self.dropout_rate
        if not self.normalize_before or self.concat_after:
            return self._forward_composed(tgt, tgt_mask, memory, memory_mask, cache), tgt_mask, memory, memory_mask
        if cache is None:
            x = mha_block(self.norm1, self.self_attn, tgt, None, None, tgt_mask, p_out=p)
        else:
            assert cache.shape == (tgt.shape[0], tgt.shape[1] - 1, self.size)
            q_mask = None if tgt_mask is None else tgt_mask[:, -1:, :]
            x = mha_block(self.norm1, self.self_attn, tgt, None, None, q_mask, last_query_only=True, p_out=p)
        if memory.shape[0] != x.shape[0]:
            # beam search: memory [G, T, D] of G utterances for n = G * g hypotheses with ONE query position each (row r
            # belongs to utterance r // g).  The g hypotheses of an utterance are g query positions over ITS memory: keys and
            # values are read once per utterance instead of once per hypothesis (and come from Decoder's per-search memo)
            G, n = memory.shape[0], x.shape[0]
            assert x.shape[1] == 1 and n % G == 0
            xg = mha_block(self.norm2, self.src_attn, x.reshape(G, n // G, self.size), memory, None, memory_mask, p_out=p,
                           pre=self._pre)
            x = xg.reshape(n, 1, self.size)
        else:
            x = mha_block(self.norm2, self.src_attn, x, memory, None, memory_mask, p_out=p,
                          pre=self._pre if cache is None else None)
        x = ffn_block(self.norm3, self.feed_forward, x, 1.0, p)
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x, tgt_mask, memory, memory_mask


class Decoder(torch.nn.Module, BatchScorerInterface):
    """reference: transformer/decoder.py:47-370 (embed input layer, selfattn layers)"""

    def __init__(self, odim, selfattention_layer_type="selfattn", attention_dim=256, attention_heads=4,
                 conv_wshare=4, conv_kernel_length=11, conv_usebias=False, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, self_attention_dropout_rate=0.0,
                 src_attention_dropout_rate=0.0, input_layer="embed", use_output_layer=True,
                 pos_enc_class=PositionalEncoding, normalize_before=True, concat_after=False):
        super().__init__()
        if input_layer != "embed" or selfattention_layer_type != "selfattn":
            raise NotImplementedError("only embed input + selfattn decoder layers are on the HIP path")
        self._register_load_state_dict_pre_hook(self._pre_hook)
        self.embed = torch.nn.Sequential(torch.nn.Embedding(odim, attention_dim),
                                         pos_enc_class(attention_dim, positional_dropout_rate))
        self.normalize_before = normalize_before
        self.decoders = repeat(
            num_blocks,
            lambda lnum: DecoderLayer(
                attention_dim, MultiHeadedAttention(attention_heads, attention_dim, self_attention_dropout_rate),
                MultiHeadedAttention(attention_heads, attention_dim, src_attention_dropout_rate),
                PositionwiseFeedForward(attention_dim, linear_units, dropout_rate), dropout_rate,
                normalize_before, concat_after))
        self.selfattention_layer_type = selfattention_layer_type
        if self.normalize_before:
            self.after_norm = LayerNorm(attention_dim)
        self.output_layer = torch.nn.Linear(attention_dim, odim) if use_output_layer else None
        self.odim = odim

    @staticmethod
    def _pre_hook(state_dict, prefix, *args):
        # reference: decoder.py:34-44 (output_norm -> after_norm rename)
        for k in list(state_dict):
            if k.startswith(prefix + "output_norm."):
                state_dict[k.replace("output_norm.", "after_norm.")] = state_dict.pop(k)

    def _embed(self, tgt, pos_offset=0):
        pos = self.embed[1]
        pos.extend_pe(tgt.size(1) + pos_offset, self.embed[0].weight.device)
        y = F_.run(F_.EmbedPEFn, tgt, self.embed[0].weight, pos.pe, pos.xscale, pos_offset)
        return F_.dropout(y, pos.dropout_rate, pos.salt, pos.training)

    def forward(self, tgt, tgt_mask, memory, memory_mask):
        x = self._embed(tgt)
        # linear_k / linear_v of all layers' source attention on the memory: one [B*T, 2*D*layers] GEMM
        shared = shared_stack_proj("kv", list(self.decoders), lambda m: m.src_attn, memory)
        try:
            x, tgt_mask, memory, memory_mask = self.decoders(x, tgt_mask, memory, memory_mask)
        finally:
            if shared:
                for m in self.decoders:
                    m._pre = None
        if self.normalize_before:
            x = self.after_norm(x)
        if self.output_layer is not None:
            x = F_.LinearFn.apply(x, self.output_layer.weight, self.output_layer.bias)
        return x, tgt_mask

    def forward_one_step(self, tgt, tgt_mask, memory, cache=None, memory_mask=None):
        """reference: decoder.py:283-321 (memory_mask [n, 1, T]: padded frames of a batch of utterances - batched beam search)"""
        x = self._embed(tgt)
        if cache is None:
            cache = [None] * len(self.decoders)
        new_cache = []
        grouped = memory.shape[0] != x.shape[0]       # memory of G utterances for G * g hypotheses (DecoderLayer.forward)
        if grouped:
            sp = self._memory_kv(memory)
            for i, decoder in enumerate(self.decoders):
                decoder._pre = ("kv", sp, i, None)
        try:
            for c, decoder in zip(cache, self.decoders):
                x, tgt_mask, memory, memory_mask = decoder(x, tgt_mask, memory, memory_mask, cache=c)
                new_cache.append(x)
        finally:
            if grouped:
                for decoder in self.decoders:
                    decoder._pre = None
        y = x[:, -1]
        if self.normalize_before:
            y = self.after_norm(y.contiguous())
        if self.output_layer is not None:
            y = F_.run(F_.LinearFn, y, self.output_layer.weight, self.output_layer.bias)
            y = ops.log_softmax_rows(y.contiguous())
        return y, new_cache

    # ---- beam-search scorer API (reference: decoder.py:323-370, scorer_interface.py) ----
    def init_state(self, x):
        return None

    def batch_init_state(self, x):
        self._kv_memo = None          # a new search: the memory (and possibly the weights) changed
        return None

    # The device-resident beam searches hand score_tree the memory of G utterances ([G, T, D], not repeated per hypothesis)
    shared_memory_ok = True
    _kv_memo = None

    def _memory_kv(self, memory):
        """keys and values of EVERY layer's source attention for the memory of a search, as one projection (the reference
        recomputes linear_k / linear_v of the same memory in every layer at every step: decoder_layer.py:103-115) ->
        F_.SharedProj whose column block i holds layer i's [k | v]; kept until batch_init_state() announces the next search"""
        key = (memory.data_ptr(), tuple(memory.shape), memory.dtype, ops.act_dtype())
        if self._kv_memo is not None and self._kv_memo[0] == key:
            return self._kv_memo[1]
        atts = [m.src_attn for m in self.decoders]
        D = memory.shape[-1]
        with torch.no_grad():
            W = torch.cat([ops.wshadow(w).detach() for a in atts for w in (a.linear_k.weight, a.linear_v.weight)], 0)
            b = torch.cat([v.detach() for a in atts for v in (a.linear_k.bias, a.linear_v.bias)], 0)
            out = ops.linear_fwd(ops.to_act_shared(memory).reshape(-1, D), W, b, out_dtype=ops.act_dtype())
        sp = F_.SharedProj(out, len(atts), False, ops.act_dtype())
        self._kv_memo = (key, sp)
        return sp

    def select_state(self, state, i, new_id=None):
        return None if state is None else state[i]

    def final_score(self, state):
        return 0.0

    def score(self, ys, state, x):
        ys_mask = subsequent_mask(len(ys), device=x.device).unsqueeze(0)
        logp, state = self.forward_one_step(ys.unsqueeze(0), ys_mask, x.unsqueeze(0), cache=state)
        return logp.squeeze(0), state

    # ---- cached decoding on key / value caches (csrc/decode.hip) ----------------------------------------------------------
    # The reference's cached step (decoder_layer.py:81-134) keeps every layer's OUTPUTS and, at every step, normalises and projects
    # keys / values of the WHOLE prefix again: 14 launches per layer here, their work growing with the prefix.  score_tree keeps
    # each layer's keys and values instead (same numbers: LayerNorm and the projections are row-wise) in time-major [Lcap, n, D]
    # buffers that a beam step never re-orders (a slot table is), folds every pre-norm into the product behind it, and runs a
    # layer as: norm1 + q/k/v | append + self-attention | out + residual | norm2 + q | source attention | out + residual |
    # norm3 + w_1 + ReLU | w_2 + residual = 8 launches.  fp32 mode, d_k = 64, memory shared by the hypotheses of an utterance.
    DECODE_KV = True

    def _kv_ok(self, ys, tree, xs):
        n, D = ys.shape[0], self.embed[0].weight.shape[1]
        att = self.decoders[0].self_attn
        return (self.DECODE_KV and not self.training and not torch.is_grad_enabled() and ops._infer > 0 and xs.is_cuda
                and ops.get_precision() == "fp32" and att.d_k == 64 and self.normalize_before and not self.decoders[0].concat_after
                and self.output_layer is not None
                and xs.shape[0] != n and n % xs.shape[0] == 0 and (isinstance(tree, dict) or (tree is None and ys.shape[1] == 1))
                and isinstance(self.decoders[0].feed_forward, PositionwiseFeedForward)
                and self.decoders[0].feed_forward.act_id in (ops.ACT_RELU, ops.ACT_SWISH))

    def _kv_weights(self):
        """per layer: the q/k/v weights of the self-attention back to back ([3D, D], [3D]); rebuilt when a weight changes"""
        ver = sum(int(m.self_attn.linear_q.weight._version) + int(m.self_attn.linear_k.weight._version) for m in self.decoders)
        memo = getattr(self, "_kv_w3", None)
        if memo is None or memo[0] != (ver, self.decoders[0].self_attn.linear_q.weight.data_ptr()):
            with torch.no_grad():
                w = [(torch.cat([a.linear_q.weight, a.linear_k.weight, a.linear_v.weight], 0).contiguous(),
                      torch.cat([a.linear_q.bias, a.linear_k.bias, a.linear_v.bias], 0).contiguous())
                     for a in (m.self_attn for m in self.decoders)]
            memo = self._kv_w3 = ((ver, self.decoders[0].self_attn.linear_q.weight.data_ptr()), w)
        return memo[1]

    @staticmethod
    def _ln_rows(x, norm, W, b, act=ops.EPI_NONE):
        """act(LayerNorm(x) W^T + b): one launch for <= 16 rows, LayerNorm + product otherwise"""
        y = ops.linear_rows_ln(x, norm.weight, norm.bias, norm.eps, W, b, act=act)
        if y is None:
            y = ops.linear_fwd(ops.layernorm_fwd(x, norm.weight, norm.bias, norm.eps)[0], W, b, act=act)
        return y

    def _score_tree_kv(self, ys, tree, xs, memory_mask):
        n, L = ys.shape
        pos = L - 1
        D = self.embed[0].weight.shape[1]
        G = xs.shape[0]
        H = self.decoders[0].self_attn.h
        # tree["dyn"] = (step on the device, newest tokens [n]): ONE captured graph serves every step - the position is read from
        # device memory by the kernels (pos below becomes the offset 0), the caches were made for the whole prefix buffer
        dyn = tree.get("dyn") if isinstance(tree, dict) else None
        pos_dev = None
        if dyn is not None:
            pos_dev, newest = dyn
            pos = 0
        if tree is None:
            Lcap = max(int(ys.stride(0)), 8) if (ys.dim() == 2 and ys.stride(1) == 1 and ys.stride(0) >= L) else 64
            tree = dict(K=[torch.empty(Lcap, n, D, device=xs.device) for _ in self.decoders],
                        V=[torch.empty(Lcap, n, D, device=xs.device) for _ in self.decoders],
                        slot=torch.zeros(n, Lcap, dtype=torch.int32, device=xs.device), Lcap=Lcap)
            self.embed[1].extend_pe(Lcap + 1, xs.device)
        elif pos >= tree["Lcap"]:          # a prefix longer than the caches were made for: double them
            Lcap = 2 * tree["Lcap"]
            grow = lambda t: torch.cat([t, torch.empty_like(t)], 0)  # noqa: E731
            tree = dict(K=[grow(t) for t in tree["K"]], V=[grow(t) for t in tree["V"]],
                        slot=torch.cat([tree["slot"], torch.zeros_like(tree["slot"])], 1).contiguous(), Lcap=Lcap)
        w3 = self._kv_weights()
        sp = self._memory_kv(xs)
        T = xs.shape[1]
        mask = _mask_u8(memory_mask, xs.device)
        if dyn is not None:
            pe = self.embed[1]
            x = ops.embed_pe(newest, self.embed[0].weight, pe.pe, 1, pe.xscale, 0, pos_dev=pos_dev)
        else:
            x = self._embed(ys[:, -1:], pos_offset=pos).reshape(n, D)
        for i, m in enumerate(self.decoders):
            qkv = self._ln_rows(x, m.norm1, w3[i][0], w3[i][1])
            ctx = ops.decode_self_attn(qkv, tree["K"][i], tree["V"][i], tree["slot"], pos, H, pos_dev=pos_dev)
            x = ops.linear_fwd(ctx, m.self_attn.linear_out.weight, m.self_attn.linear_out.bias, R=x)
            a = m.src_attn
            q2 = self._ln_rows(x, m.norm2, a.linear_q.weight, a.linear_q.bias)
            k2, v2 = sp.block(i, 0), sp.block(i, D)
            # a handful of hypotheses: one wave group per (hypothesis, head); many (32 utterances x beam 10) re-read an utterance's
            # keys / values once per hypothesis that way (431 -> 360 utt/s) - those keep the 64-queries-per-workgroup kernel
            # (... 32 utterances: one workgroup per (utterance, head) for all its hypotheses, keys / values read once)
            cx = ops.decode_src_attn(q2, k2.t, k2.off, v2.off, k2.ld, mask, G, n // G, T, H, group=n > 32) \
                if (k2.t.dtype == torch.float32 and (mask is None or mask.shape[1] == 1)) else None
            fwd = None
            if cx is None and F_.FUSE_ATTN and ops.attn_fwd_supported(n // G, T, a.d_k, False):
                fwd = F_.attn_fwd_fused(q2, None, k2, v2, None, mask, G, n // G, T, H, a.d_k)
            if cx is not None:
                pass
            elif fwd is not None:
                cx = fwd[2]
            else:
                P = F_.attn_scores_fwd(q2, None, k2, None, mask, G, n // G, T, H, a.d_k)
                cx = F_.attn_context_fwd(P, v2, G, n // G, T, H, a.d_k)
            x = ops.linear_fwd(cx, a.linear_out.weight, a.linear_out.bias, R=x)
            ff = m.feed_forward
            h = self._ln_rows(x, m.norm3, ff.w_1.weight, ff.w_1.bias, act=ff.act_id)      # (ACT_RELU / ACT_SWISH = EPI_RELU / EPI_SWISH)
            x = ops.linear_fwd(h, ff.w_2.weight, ff.w_2.bias, R=x)
        y = self._ln_rows(x, self.after_norm, self.output_layer.weight, self.output_layer.bias)
        new = dict(tree)
        new["pos"] = pos
        new["pos_dev"] = pos_dev
        new.pop("dyn", None)
        return ops.log_softmax_rows(y.contiguous()), new

    @staticmethod
    def reorder_tree(tree, hyp_i):
        """the batched state behind a beam step's selection (BeamSearch._reorder): the layer-output caches of the reference-shaped
        step are gathered; the key / value caches stay where they are, only the slot table follows the hypotheses"""
        if not isinstance(tree, dict):
            return None if tree is None else [t.index_select(0, hyp_i) for t in tree]
        new = dict(tree)
        new["slot"] = ops.beam_slots(tree["slot"], hyp_i, tree["pos"], pos_dev=tree.get("pos_dev"))
        return new

    def score_tree(self, ys, tree, xs, memory_mask=None):
        """batch_score on a BATCHED state (list per layer of [n, L-1, D], or None): no per-hypothesis stacking / slicing;
        the search reorders it with index_select (BeamSearch device loop)"""
        if self._kv_ok(ys, tree, xs):
            return self._score_tree_kv(ys, tree, xs, memory_mask)
        # all hypotheses of a search have the same length and only the NEWEST position queries (cached decoding): its row of the
        # causal mask is all ones, i.e. no mask at all - the reference builds subsequent_mask(L) and slices that row every step
        # (decoder.py:342-343, decoder_layer.py:88-101); here that was a tril, a fill and six mask conversions per step
        ys_mask = None if (tree is not None or ys.size(-1) == 1) else subsequent_mask(ys.size(-1), device=xs.device).unsqueeze(0)
        return self.forward_one_step(ys, ys_mask, xs, cache=tree, memory_mask=memory_mask)

    def final_tree(self, tree):
        return 0.0

    def batch_score(self, ys, states, xs):
        n_batch = len(ys)
        n_layers = len(self.decoders)
        if states[0] is None:
            batch_state = None
        else:
            batch_state = [torch.stack([states[b][i] for b in range(n_batch)]) for i in range(n_layers)]
        ys_mask = subsequent_mask(ys.size(-1), device=xs.device).unsqueeze(0)
        logp, states = self.forward_one_step(ys, ys_mask, xs, cache=batch_state)
        state_list = [[states[i][b] for i in range(n_layers)] for b in range(n_batch)]
        return logp, state_list


# ---- losses -------------------------------------------------------------------------------------
class LabelSmoothingLoss(torch.nn.Module):
    """reference: transformer/label_smoothing_loss.py:13-63.  forward returns the loss; the per-row
    argmax-correct flags of the same pass are kept in `self.correct_rows` for th_accuracy."""

    def __init__(self, size, padding_idx, smoothing, normalize_length=False, criterion=None):
        super().__init__()
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.normalize_length = normalize_length
        self.correct_rows = None

    def forward(self, x, target, n_tokens=None):
        assert x.size(2) == self.size
        batch_size = x.size(0)
        if self.normalize_length:
            if n_tokens is None:   # same host sync as label_smoothing_loss.py:58
                n_tokens = int((target != self.padding_idx).sum().item())
            denom = n_tokens
        else:
            denom = batch_size
        loss, self.correct_rows = F_.LabelSmoothingLossFn.apply(x, target, self.smoothing, self.padding_idx,
                                                                float(denom))
        return loss


class CTC(torch.nn.Module):
    """reference: ctc.py:12-151 (espnet1) / espnet2/asr/ctc.py:6-111.
    ctc_type is accepted for interface parity ("warpctc" and "builtin" compute the same quantity:
    sum_b -log p / B); both run the espnet_amd HIP kernel."""

    def __init__(self, odim, eprojs, dropout_rate, ctc_type="warpctc", reduce=True):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.loss = None
        self.ctc_lo = torch.nn.Linear(eprojs, odim)
        self.probs = None
        if ctc_type not in ("builtin", "warpctc"):
            raise ValueError('ctc_type must be "builtin" or "warpctc": {}'.format(ctc_type))
        self.ctc_type = ctc_type
        self.ignore_id = -1
        self.reduce = reduce
        self.salt = ops.new_salt()
        if not reduce:
            raise NotImplementedError("reduce=False is not on the path")

    def logits(self, hs_pad, loss_path=False):
        if loss_path:   # ctc.py:85: F.dropout(hs_pad, p) without `training=` => active in eval mode too
            hs_pad = F_.dropout(hs_pad, self.dropout_rate, self.salt, True)
        return F_.LinearFn.apply(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias)

    def forward(self, hs_pad, hlens, ys_pad):
        """hs_pad (B,T,D); hlens list/tensor of valid frames; ys_pad (B,L) int64 padded with -1."""
        ys_hat = self.logits(hs_pad, loss_path=True)
        if isinstance(hlens, torch.Tensor):
            hl = hlens.to(device=hs_pad.device, dtype=torch.int32)
        else:
            hl = ops.h2d_cached("ctclens", np.asarray([int(v) for v in hlens], dtype=np.int32), hs_pad.device)
        if not ys_pad.is_cuda:     # host labels (kept on the host for the decoder's label parsing): cached upload
            ys_pad = ops.h2d_cached("ctc_ys", ys_pad.numpy(), hs_pad.device)
        self.loss = F_.CTCLossFn.apply(ys_hat, ys_pad.contiguous(), hl, 0, self.ignore_id)
        return self.loss

    def softmax(self, hs_pad):
        lp = self.log_softmax(hs_pad)
        self.probs = torch.exp(lp)
        return self.probs

    def log_softmax(self, hs_pad):
        y = self.logits(hs_pad)
        return ops.log_softmax_rows(y.reshape(-1, y.shape[-1]).contiguous()).view(y.shape)

    def argmax(self, hs_pad):
        y = self.logits(hs_pad)
        return ops.argmax_rows(y.reshape(-1, y.shape[-1]).contiguous()).view(y.shape[:-1]).long()


def th_accuracy(correct_rows, pad_targets, ignore_label):
    """reference: nets_utils.py:299-319; numerator comes from the fused loss kernel."""
    num = ops.reduce_sum(correct_rows)
    den = (pad_targets != ignore_label).sum()
    return num / den
