"""Decoder-side attention of the RNN path.  reference: espnet/nets/pytorch_backend/rnn/attentions.py.
All twelve attention types listed by `initial_att` (:1722-1771) run on the espnet_amd HIP kernels."""
import math

import numpy as np
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_


class AttLoc(torch.nn.Module):
    """reference: rnn/attentions.py:250-380 (same parameters: mlp_enc, mlp_dec, mlp_att, loc_conv, gvec)"""

    def __init__(self, eprojs, dunits, att_dim, aconv_chans, aconv_filts, han_mode=False):
        super().__init__()
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.mlp_att = torch.nn.Linear(aconv_chans, att_dim, bias=False)
        self.loc_conv = torch.nn.Conv2d(1, aconv_chans, (1, 2 * aconv_filts + 1), padding=(0, aconv_filts), bias=False)
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim = dunits, eprojs, att_dim
        self.han_mode = han_mode
        self.reset()

    def reset(self):
        """reset states (attentions.py:291-296)"""
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.mask = None
        self._lens = None
        self._acc = None

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0, last_attended_idx=None,
                backward_window=1, forward_window=3):
        if last_attended_idx is not None:
            raise NotImplementedError("attention constraint (TTS) is not on the ASR path")
        batch = enc_hs_pad.shape[0]
        dev = enc_hs_pad.device
        first = False
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            lens = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens_host = lens
            self._lens = ops.h2d_cached("attlens", lens, dev)
            # the first step of a sequence that starts from no attention history: every later step depends on it, so its
            # backward runs last and the steps can keep the gradients of enc_h / pre_compute_enc_h as one running sum
            first = att_prev is None and not self.han_mode
            self._acc = {} if first else None
        if dec_z is None:
            dec_z = enc_hs_pad.new_zeros(batch, self.dunits)
        else:
            dec_z = dec_z.view(batch, self.dunits)
        if att_prev is None:
            # uniform over the valid frames (attentions.py:331-337); a constant, built on the host
            keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
            att_prev = ops.h2d_cached("attuniform", keep / self._lens_host[:, None].astype(np.float32), dev)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, att_prev, self._lens, float(scaling),
                                     self.loc_conv.weight, self.mlp_att.weight, self.gvec.weight, self.gvec.bias,
                                     self._acc, first)
        return c, w


class AttAdd(torch.nn.Module):
    """additive attention.  reference: rnn/attentions.py:167-247 (e = gvec . tanh(mlp_enc(h) + mlp_dec(z)),
    softmax(2.0 * e)): the location-aware kernels with the location term switched off."""

    def __init__(self, eprojs, dunits, att_dim, han_mode=False):
        super().__init__()
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim, self.han_mode = dunits, eprojs, att_dim, han_mode
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.mask = None
        self._lens = None

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0):
        batch = enc_hs_pad.shape[0]
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            self._lens = ops.h2d_cached("attlens", np.asarray([int(v) for v in enc_hs_len], dtype=np.int32),
                                        enc_hs_pad.device)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None)
        return R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, None, self._lens, float(scaling),
                                     None, None, self.gvec.weight, self.gvec.bias)


class AttDot(torch.nn.Module):
    """dot-product attention.  reference: rnn/attentions.py:91-164"""

    def __init__(self, eprojs, dunits, att_dim, han_mode=False):
        super().__init__()
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim)
        self.dunits, self.eprojs, self.att_dim, self.han_mode = dunits, eprojs, att_dim, han_mode
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.mask = None
        self._lens = None

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0):
        batch = enc_hs_pad.shape[0]
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = R_.ActFn.apply(
                F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias), ops.ACT_TANH)
            self._lens = ops.h2d_cached("attlens", np.asarray([int(v) for v in enc_hs_len], dtype=np.int32),
                                        enc_hs_pad.device)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        q = R_.ActFn.apply(F_.LinearFn.apply(dec_z, self.mlp_dec.weight, self.mlp_dec.bias), ops.ACT_TANH)
        return R_.AttDotStepFn.apply(self.pre_compute_enc_h, q, self.enc_h, self._lens, float(scaling))


class _MultiHeadBase(torch.nn.Module):
    """shared part of the multi-head attentions (rnn/attentions.py:845-1385): per head k = mlp_k(h) (no bias),
    v = mlp_v(h) (no bias), q = mlp_q(z) (bias); the H context vectors are concatenated and mixed by mlp_o."""

    def __init__(self, eprojs, dunits, aheads, att_dim_k, att_dim_v, han_mode=False):
        super().__init__()
        self.mlp_q = torch.nn.ModuleList([torch.nn.Linear(dunits, att_dim_k) for _ in range(aheads)])
        self.mlp_k = torch.nn.ModuleList([torch.nn.Linear(eprojs, att_dim_k, bias=False) for _ in range(aheads)])
        self.mlp_v = torch.nn.ModuleList([torch.nn.Linear(eprojs, att_dim_v, bias=False) for _ in range(aheads)])
        self.gvec = torch.nn.ModuleList([torch.nn.Linear(att_dim_k, 1) for _ in range(aheads)])
        self.mlp_o = torch.nn.Linear(aheads * att_dim_v, eprojs, bias=False)
        self.dunits, self.eprojs, self.aheads = dunits, eprojs, aheads
        self.att_dim_k, self.att_dim_v = att_dim_k, att_dim_v
        self.scaling = 1.0 / math.sqrt(att_dim_k)
        self.han_mode = han_mode
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_k = None
        self.pre_compute_v = None
        self.mask = None
        self._lens = None

    def _precompute(self, enc_hs_pad, enc_hs_len):
        if self.pre_compute_k is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_k = [F_.LinearFn.apply(self.enc_h, self.mlp_k[h].weight, None) for h in range(self.aheads)]
            self.pre_compute_v = [F_.LinearFn.apply(self.enc_h, self.mlp_v[h].weight, None) for h in range(self.aheads)]
            lens = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens_host = lens
            self._lens = ops.h2d_cached("attlens", lens, enc_hs_pad.device)

    def _uniform(self, dev):
        keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
        return ops.h2d_cached("attuniform", keep / self._lens_host[:, None].astype(np.float32), dev)

    def _head(self, h, dec_z, att_prev_h, scaling):
        raise NotImplementedError

    def _forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling, needs_prev):
        batch = enc_hs_pad.shape[0]
        self._precompute(enc_hs_pad, enc_hs_len)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        if needs_prev and att_prev is None:
            u = self._uniform(enc_hs_pad.device)
            att_prev = [u for _ in range(self.aheads)]
        c, w = [], []
        for h in range(self.aheads):
            q = F_.LinearFn.apply(dec_z, self.mlp_q[h].weight, self.mlp_q[h].bias)
            ch, wh = self._head(h, q, att_prev[h] if needs_prev else None, scaling)
            c.append(ch)
            w.append(wh)
        c = F_.LinearFn.apply(torch.cat(c, dim=1), self.mlp_o.weight, None)
        return c, w


class AttMultiHeadDot(_MultiHeadBase):
    """reference: rnn/attentions.py:845-990 (k = tanh(mlp_k h), q = tanh(mlp_q z), e = k . q, softmax(e / sqrt(d_k)));
    this variant has no gvec parameters"""

    def __init__(self, eprojs, dunits, aheads, att_dim_k, att_dim_v, han_mode=False):
        super().__init__(eprojs, dunits, aheads, att_dim_k, att_dim_v, han_mode)
        del self.gvec

    def _precompute(self, enc_hs_pad, enc_hs_len):
        fresh = self.pre_compute_k is None or self.han_mode
        super()._precompute(enc_hs_pad, enc_hs_len)
        if fresh:
            self.pre_compute_k = [R_.ActFn.apply(k, ops.ACT_TANH) for k in self.pre_compute_k]

    def _head(self, h, q, att_prev_h, scaling):
        return R_.AttDotStepFn.apply(self.pre_compute_k[h], R_.ActFn.apply(q, ops.ACT_TANH), self.pre_compute_v[h],
                                     self._lens, scaling)

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev):
        return self._forward(enc_hs_pad, enc_hs_len, dec_z, att_prev, self.scaling, False)


class AttMultiHeadAdd(_MultiHeadBase):
    """reference: rnn/attentions.py:993-1107 (per head: gvec . tanh(k + q), softmax(e / sqrt(d_k)))"""

    def _head(self, h, q, att_prev_h, scaling):
        return R_.AttLocStepFn.apply(self.pre_compute_v[h], self.pre_compute_k[h], q, None, self._lens, scaling, None,
                                     None, self.gvec[h].weight, self.gvec[h].bias)

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev):
        return self._forward(enc_hs_pad, enc_hs_len, dec_z, att_prev, self.scaling, False)


class AttMultiHeadLoc(_MultiHeadBase):
    """reference: rnn/attentions.py:1110-1258 (per head location-aware energies, softmax(scaling * e), scaling 2.0)"""

    def __init__(self, eprojs, dunits, aheads, att_dim_k, att_dim_v, aconv_chans, aconv_filts, han_mode=False):
        super().__init__(eprojs, dunits, aheads, att_dim_k, att_dim_v, han_mode)
        self.loc_conv = torch.nn.ModuleList([
            torch.nn.Conv2d(1, aconv_chans, (1, 2 * aconv_filts + 1), padding=(0, aconv_filts), bias=False)
            for _ in range(aheads)])
        self.mlp_att = torch.nn.ModuleList([torch.nn.Linear(aconv_chans, att_dim_k, bias=False) for _ in range(aheads)])

    def _head(self, h, q, att_prev_h, scaling):
        return R_.AttLocStepFn.apply(self.pre_compute_v[h], self.pre_compute_k[h], q, att_prev_h, self._lens, scaling,
                                     self.loc_conv[h].weight, self.mlp_att[h].weight, self.gvec[h].weight,
                                     self.gvec[h].bias)

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0):
        return self._forward(enc_hs_pad, enc_hs_len, dec_z, att_prev, float(scaling), True)


class AttMultiHeadMultiResLoc(AttMultiHeadLoc):
    """reference: rnn/attentions.py:1261-1385: head h uses filter half-width aconv_filts * (h + 1) // aheads,
    softmax(e / sqrt(d_k))"""

    def __init__(self, eprojs, dunits, aheads, att_dim_k, att_dim_v, aconv_chans, aconv_filts, han_mode=False):
        _MultiHeadBase.__init__(self, eprojs, dunits, aheads, att_dim_k, att_dim_v, han_mode)
        self.loc_conv = torch.nn.ModuleList()
        self.mlp_att = torch.nn.ModuleList()
        for h in range(aheads):
            afilts = aconv_filts * (h + 1) // aheads
            self.loc_conv += [torch.nn.Conv2d(1, aconv_chans, (1, 2 * afilts + 1), padding=(0, afilts), bias=False)]
            self.mlp_att += [torch.nn.Linear(aconv_chans, att_dim_k, bias=False)]

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev):
        return self._forward(enc_hs_pad, enc_hs_len, dec_z, att_prev, self.scaling, True)


class NoAtt(torch.nn.Module):
    """reference: rnn/attentions.py:46-91: the context is the mean of the valid encoder frames, computed once.
    Runs the dot-attention kernels on zero keys (softmax of zeros over the valid frames = the uniform weights)."""

    def __init__(self):
        super().__init__()
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.c = None

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev):
        batch = enc_hs_pad.shape[0]
        if self.pre_compute_enc_h is None:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
        if att_prev is None:
            dev = enc_hs_pad.device
            lens = ops.h2d_cached("attlens", np.asarray([int(v) for v in enc_hs_len], dtype=np.int32), dev)
            zk = torch.zeros(batch, self.h_length, 1, device=dev)
            self.c, att_prev = R_.AttDotStepFn.apply(zk, torch.zeros(batch, 1, device=dev), self.enc_h, lens, 1.0)
            att_prev = att_prev.detach()
        return self.c, att_prev


class _Coverage(list):
    """att_prev_list of the coverage attentions with its running sum kept alongside (the reference re-adds the whole
    list at every step, attentions.py:433)"""

    total = None


class AttCov(torch.nn.Module):
    """coverage attention.  reference: rnn/attentions.py:383-482: e = gvec . tanh(wvec(sum of previous weights) +
    mlp_enc(h) + mlp_dec(z)).  Maps onto the location-aware kernels: a 1-channel, 1-tap unit "convolution" of the
    coverage vector, wvec.weight as the channel projection and wvec.bias folded into the decoder projection."""

    def __init__(self, eprojs, dunits, att_dim, han_mode=False):
        super().__init__()
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.wvec = torch.nn.Linear(1, att_dim)
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim, self.han_mode = dunits, eprojs, att_dim, han_mode
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.mask = None
        self._lens = None

    def _prepare(self, enc_hs_pad, enc_hs_len, dec_z, att_prev_list):
        batch = enc_hs_pad.shape[0]
        dev = enc_hs_pad.device
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            self._lens_host = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens = ops.h2d_cached("attlens", self._lens_host, dev)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        if att_prev_list is None:
            keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
            att_prev_list = _Coverage([ops.h2d_cached("attuniform", keep / self._lens_host[:, None].astype(np.float32), dev)])
            att_prev_list.total = att_prev_list[0]
        elif getattr(att_prev_list, "total", None) is None:      # a plain list handed in by an external caller
            tot = att_prev_list[0]
            for w in att_prev_list[1:]:
                tot = R_.AddFn.apply(tot, w)
            att_prev_list = _Coverage(att_prev_list)
            att_prev_list.total = tot
        return dec_z, att_prev_list

    @staticmethod
    def _extend(att_prev_list, w):
        out = _Coverage(list(att_prev_list) + [w])      # a new list: beam search keeps the old one for sibling hypotheses
        out.total = R_.AddFn.apply(att_prev_list.total, w)
        return out

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev_list, scaling=2.0):
        dec_z, att_prev_list = self._prepare(enc_hs_pad, enc_hs_len, dec_z, att_prev_list)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, self.wvec.bias)
        one = ops.h2d_cached("attone", np.ones((1, 1, 1, 1), dtype=np.float32), enc_hs_pad.device)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, att_prev_list.total, self._lens,
                                     float(scaling), one, self.wvec.weight, self.gvec.weight, self.gvec.bias)
        return c, self._extend(att_prev_list, w)


class AttCovLoc(AttCov):
    """coverage + location attention.  reference: rnn/attentions.py:729-842 (the location convolution is applied to
    the coverage vector instead of the previous weights)"""

    def __init__(self, eprojs, dunits, att_dim, aconv_chans, aconv_filts, han_mode=False):
        torch.nn.Module.__init__(self)
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.mlp_att = torch.nn.Linear(aconv_chans, att_dim, bias=False)
        self.loc_conv = torch.nn.Conv2d(1, aconv_chans, (1, 2 * aconv_filts + 1), padding=(0, aconv_filts), bias=False)
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim, self.aconv_chans, self.han_mode = dunits, eprojs, att_dim, aconv_chans, han_mode
        self.reset()

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev_list, scaling=2.0):
        dec_z, att_prev_list = self._prepare(enc_hs_pad, enc_hs_len, dec_z, att_prev_list)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, att_prev_list.total, self._lens,
                                     float(scaling), self.loc_conv.weight, self.mlp_att.weight, self.gvec.weight,
                                     self.gvec.bias)
        return c, self._extend(att_prev_list, w)


class AttLoc2D(AttLoc):
    """location-aware attention over a window of the last att_win weight vectors.  reference: rnn/attentions.py:485-603
    (Conv2d(1, C, (att_win, 2F+1)) spanning the whole window = the location kernels with att_win history rows)"""

    def __init__(self, eprojs, dunits, att_dim, att_win, aconv_chans, aconv_filts, han_mode=False):
        torch.nn.Module.__init__(self)
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.mlp_att = torch.nn.Linear(aconv_chans, att_dim, bias=False)
        self.loc_conv = torch.nn.Conv2d(1, aconv_chans, (att_win, 2 * aconv_filts + 1), padding=(0, aconv_filts),
                                        bias=False)
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim, self.aconv_chans, self.att_win = dunits, eprojs, att_dim, aconv_chans, att_win
        self.han_mode = han_mode
        self.reset()

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0):
        first = att_prev is None      # the window starts as att_win copies of the uniform weights (:560-566)
        batch = enc_hs_pad.shape[0]
        dev = enc_hs_pad.device
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            self._lens_host = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens = ops.h2d_cached("attlens", self._lens_host, dev)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        if first:
            keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
            uni = keep / self._lens_host[:, None].astype(np.float32)
            att_prev = ops.h2d_cached("attuniform2d", np.repeat(uni[:, None, :], self.att_win, axis=1), dev)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, att_prev, self._lens, float(scaling),
                                     self.loc_conv.weight, self.mlp_att.weight, self.gvec.weight, self.gvec.bias)
        # slide the window: drop the oldest row, append the new weights (attentions.py:598-601; data movement only)
        att_prev = torch.cat([att_prev[:, 1:], w.unsqueeze(1)], dim=1)
        return c, att_prev


class AttLocRec(torch.nn.Module):
    """location-aware recurrent attention.  reference: rnn/attentions.py:606-726: the location features are max-pooled
    over time and fed to an LSTMCell whose output replaces the per-frame location term; the energy is then the additive
    one with the decoder projection shifted by that output."""

    def __init__(self, eprojs, dunits, att_dim, aconv_chans, aconv_filts, han_mode=False):
        super().__init__()
        self.mlp_enc = torch.nn.Linear(eprojs, att_dim)
        self.mlp_dec = torch.nn.Linear(dunits, att_dim, bias=False)
        self.loc_conv = torch.nn.Conv2d(1, aconv_chans, (1, 2 * aconv_filts + 1), padding=(0, aconv_filts), bias=False)
        self.att_lstm = torch.nn.LSTMCell(aconv_chans, att_dim, bias=False)      # parameter container only
        self.gvec = torch.nn.Linear(att_dim, 1)
        self.dunits, self.eprojs, self.att_dim, self.han_mode = dunits, eprojs, att_dim, han_mode
        self.reset()

    def reset(self):
        self.h_length = None
        self.enc_h = None
        self.pre_compute_enc_h = None
        self.mask = None
        self._lens = None

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev_states, scaling=2.0):
        batch = enc_hs_pad.shape[0]
        dev = enc_hs_pad.device
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            self._lens_host = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens = ops.h2d_cached("attlens", self._lens_host, dev)
        dec_z = enc_hs_pad.new_zeros(batch, self.dunits) if dec_z is None else dec_z.view(batch, self.dunits)
        if att_prev_states is None:
            keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
            att_prev = ops.h2d_cached("attuniform", keep / self._lens_host[:, None].astype(np.float32), dev)
            att_states = (enc_hs_pad.new_zeros(batch, self.att_dim), enc_hs_pad.new_zeros(batch, self.att_dim))
        else:
            att_prev, att_states = att_prev_states
        pooled = R_.ConvMaxFn.apply(att_prev, self.loc_conv.weight)
        gx = F_.LinearFn.apply(pooled, self.att_lstm.weight_ih, None)
        att_h, att_c = R_.LSTMCellFn.apply(gx, att_states[0], att_states[1], self.att_lstm.weight_hh, None)
        dec_proj = R_.AddFn.apply(F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None), att_h)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, None, self._lens, float(scaling),
                                     None, None, self.gvec.weight, self.gvec.bias)
        return c, (w, (att_h, att_c))


def initial_att(atype, eprojs, dunits, aheads, adim, awin, aconv_chans, aconv_filts, han_mode=False):
    """reference: rnn/attentions.py:1722-1771"""
    if atype == "location2d":
        return AttLoc2D(eprojs, dunits, adim, awin, aconv_chans, aconv_filts, han_mode)
    if atype == "location_recurrent":
        return AttLocRec(eprojs, dunits, adim, aconv_chans, aconv_filts, han_mode)
    if atype == "noatt":
        return NoAtt()
    if atype == "coverage":
        return AttCov(eprojs, dunits, adim, han_mode)
    if atype == "coverage_location":
        return AttCovLoc(eprojs, dunits, adim, aconv_chans, aconv_filts, han_mode)
    if atype == "location":
        return AttLoc(eprojs, dunits, adim, aconv_chans, aconv_filts, han_mode)
    if atype == "dot":
        return AttDot(eprojs, dunits, adim, han_mode)
    if atype == "add":
        return AttAdd(eprojs, dunits, adim, han_mode)
    if atype == "multi_head_dot":
        return AttMultiHeadDot(eprojs, dunits, aheads, adim, adim, han_mode)
    if atype == "multi_head_add":
        return AttMultiHeadAdd(eprojs, dunits, aheads, adim, adim, han_mode)
    if atype == "multi_head_loc":
        return AttMultiHeadLoc(eprojs, dunits, aheads, adim, adim, aconv_chans, aconv_filts, han_mode)
    if atype == "multi_head_multi_res_loc":
        return AttMultiHeadMultiResLoc(eprojs, dunits, aheads, adim, adim, aconv_chans, aconv_filts, han_mode)
    raise ValueError("unknown attention type %r" % atype)


def att_for(args, num_att=1, han_mode=False):
    """reference: rnn/attentions.py:1661-1719 (single-encoder case)"""
    if getattr(args, "num_encs", 1) != 1:
        raise NotImplementedError("multi-encoder attention is out of the hot-path scope")
    att_list = torch.nn.ModuleList()
    for _ in range(num_att):
        att_list.append(initial_att(args.atype, args.eprojs, args.dunits, getattr(args, "aheads", None), args.adim,
                                    getattr(args, "awin", None), getattr(args, "aconv_chans", None),
                                    getattr(args, "aconv_filts", None)))
    return att_list
