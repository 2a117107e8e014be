"""Decoder-side attention of the RNN path.  reference: espnet/nets/pytorch_backend/rnn/attentions.py.
AttLoc (location-aware attention, :250-380) runs on the espnet_amd HIP kernels; the other attention
types listed by `initial_att` (:1722-1771) have no kernel yet and raise NotImplementedError."""
import numpy as np
import torch

from ... import functional as F_
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

    def forward(self, enc_hs_pad, enc_hs_len, dec_z, att_prev, scaling=2.0, last_attended_idx=None,
                backward_window=1, forward_window=3):
        if last_attended_idx is not None:
            raise NotImplementedError("attention constraint (TTS) is not on the ASR path")
        batch = enc_hs_pad.shape[0]
        dev = enc_hs_pad.device
        if self.pre_compute_enc_h is None or self.han_mode:
            self.enc_h = enc_hs_pad.contiguous()
            self.h_length = self.enc_h.size(1)
            self.pre_compute_enc_h = F_.LinearFn.apply(self.enc_h, self.mlp_enc.weight, self.mlp_enc.bias)
            lens = np.asarray([int(v) for v in enc_hs_len], dtype=np.int32)
            self._lens_host = lens
            self._lens = torch.from_numpy(lens).to(dev, non_blocking=True)
        if dec_z is None:
            dec_z = enc_hs_pad.new_zeros(batch, self.dunits)
        else:
            dec_z = dec_z.view(batch, self.dunits)
        if att_prev is None:
            # uniform over the valid frames (attentions.py:331-337); a constant, built on the host
            keep = (np.arange(self.h_length)[None, :] < self._lens_host[:, None]).astype(np.float32)
            att_prev = torch.from_numpy(keep / self._lens_host[:, None].astype(np.float32)).to(dev, non_blocking=True)
        dec_proj = F_.LinearFn.apply(dec_z, self.mlp_dec.weight, None)
        c, w = R_.AttLocStepFn.apply(self.enc_h, self.pre_compute_enc_h, dec_proj, att_prev, self._lens, float(scaling),
                                     self.loc_conv.weight, self.mlp_att.weight, self.gvec.weight, self.gvec.bias)
        return c, w


def initial_att(atype, eprojs, dunits, aheads, adim, awin, aconv_chans, aconv_filts, han_mode=False):
    """reference: rnn/attentions.py:1722-1771"""
    if atype == "location":
        return AttLoc(eprojs, dunits, adim, aconv_chans, aconv_filts, han_mode)
    raise NotImplementedError("atype %r: only 'location' has HIP kernels (SURVEY.md 8f lists the rest as next)" % atype)


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
