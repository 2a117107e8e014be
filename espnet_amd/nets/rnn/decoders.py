"""Attention decoder of the RNN path.  reference: espnet/nets/pytorch_backend/rnn/decoders.py:28-311
(Decoder.__init__/zero_state/rnn_forward/forward, lstm, single encoder, no scheduled sampling)."""
import math

import numpy as np
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_
from ..modules import pad_list, th_accuracy


class LSTMCell(torch.nn.Module):
    """parameter container with torch.nn.LSTMCell's names; forward = x-projection GEMM + fused cell"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        k = 1.0 / math.sqrt(hidden_size)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = torch.nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = torch.nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))
        self.bias_hh = torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))

    def forward(self, x, state):
        h, c = state
        gx = F_.LinearFn.apply(x, self.weight_ih, self.bias_ih)
        return R_.LSTMCellFn.apply(gx, h, c, self.weight_hh, self.bias_hh)


class GRUCell(torch.nn.Module):
    """parameter container with torch.nn.GRUCell's names; forward(x, h) -> h'"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        k = 1.0 / math.sqrt(hidden_size)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = torch.nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = torch.nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = torch.nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh = torch.nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))

    def forward(self, x, h):
        gx = F_.LinearFn.apply(x, self.weight_ih, self.bias_ih)
        return R_.GRUCellFn.apply(gx, h, self.weight_hh, self.bias_hh)


class Decoder(torch.nn.Module):
    """reference: rnn/decoders.py:28-311"""

    def __init__(self, eprojs, odim, dtype, dlayers, dunits, sos, eos, att, verbose=0, char_list=None,
                 labeldist=None, lsm_weight=0.0, sampling_probability=0.0, dropout=0.0, context_residual=False,
                 replace_sos=False, num_encs=1):
        super().__init__()
        if dtype not in ("lstm", "gru"):
            raise NotImplementedError("dtype %r: lstm and gru decoders have HIP kernels" % dtype)
        if num_encs != 1 or replace_sos or labeldist is not None or sampling_probability > 0.0:
            raise NotImplementedError("multi-encoder / replace_sos / label-dist smoothing / scheduled sampling "
                                      "are outside the hot-path scope")
        self.dtype, self.dunits, self.dlayers, self.context_residual = dtype, dunits, dlayers, context_residual
        self.embed = torch.nn.Embedding(odim, dunits)
        cell = LSTMCell if dtype == "lstm" else GRUCell
        self.decoder = torch.nn.ModuleList([cell(dunits + eprojs, dunits)] + [cell(dunits, dunits) for _ in range(1, dlayers)])
        self.ignore_id = -1
        self.output = torch.nn.Linear(dunits + eprojs if context_residual else dunits, odim)
        self.loss = None
        self.att = att
        self.sos, self.eos, self.odim = sos, eos, odim
        self.verbose, self.char_list = verbose, char_list
        self.lsm_weight = lsm_weight
        self.sampling_probability = sampling_probability
        self.dropout = dropout
        self.num_encs = num_encs
        self.logzero = -10000000000.0
        self.salt_emb = ops.new_salt()
        self.salts = [ops.new_salt() for _ in range(dlayers)]

    def zero_state(self, hs_pad):
        return hs_pad.new_zeros(hs_pad.size(0), self.dunits)

    def _drop(self, k, x, step):
        # one salt per (layer, step): torch draws a fresh mask at every call of dropout_dec[k]
        return F_.dropout(x, self.dropout, self.salts[k] + 131 * (step + 1), self.training)

    def rnn_forward(self, ey, z_list, c_list, z_prev, c_prev, step=0):
        if self.dtype == "lstm":
            z_list[0], c_list[0] = self.decoder[0](ey, (z_prev[0], c_prev[0]))
            for i in range(1, self.dlayers):
                z_list[i], c_list[i] = self.decoder[i](self._drop(i - 1, z_list[i - 1], step), (z_prev[i], c_prev[i]))
        else:
            z_list[0] = self.decoder[0](ey, z_prev[0])
            for i in range(1, self.dlayers):
                z_list[i] = self.decoder[i](self._drop(i - 1, z_list[i - 1], step), z_prev[i])
        return z_list, c_list

    def forward(self, hs_pad, hlens, ys_pad, strm_idx=0, lang_ids=None):
        """hs_pad (B,T,D), hlens list[int], ys_pad (B,Lmax) padded with -1 -> (loss, acc, ppl)"""
        hlens = [int(v) for v in hlens]
        ys = [y[y != self.ignore_id] for y in ys_pad.cpu()]       # host-side label parsing (decoders.py:167);
        # pass ys_pad as a CPU tensor to keep this free of a device->host sync (needed under hipGraph capture)
        eos = ys[0].new([self.eos])
        sos = ys[0].new([self.sos])
        ys_in = [torch.cat([sos, y], dim=0) for y in ys]
        ys_out = [torch.cat([y, eos], dim=0) for y in ys]
        dev = hs_pad.device
        ys_in_pad = ops.h2d_cached("ys_in", pad_list(ys_in, self.eos).numpy(), dev)
        ys_out_pad = ops.h2d_cached("ys_out", pad_list(ys_out, self.ignore_id).numpy(), dev)
        batch, olength = ys_out_pad.size(0), ys_out_pad.size(1)
        c_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        z_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        z_all = []
        att_w = None
        att = self.att[min(strm_idx, len(self.att) - 1)]
        att.reset()
        eys = F_.dropout(R_.PlainEmbedFn.apply(ys_in_pad, self.embed.weight, -1), self.dropout, self.salt_emb,
                         self.training)
        for i in range(olength):
            att_c, att_w = att(hs_pad, hlens, self._drop(0, z_list[0], i + olength), att_w)
            ey = torch.cat((eys[:, i, :], att_c), dim=1)
            z_list, c_list = self.rnn_forward(ey, z_list, c_list, z_list, c_list, step=i)
            top = self._drop(self.dlayers - 1, z_list[-1], i + 2 * olength)
            z_all.append(torch.cat((top, att_c), dim=-1) if self.context_residual else top)
        z_all = torch.stack(z_all, dim=1).view(batch * olength, -1)
        y_all = F_.LinearFn.apply(z_all, self.output.weight, self.output.bias)
        n_tok = sum(len(y) for y in ys_out)
        ce, correct = F_.LabelSmoothingLossFn.apply(y_all.view(batch, olength, -1), ys_out_pad, 0.0, self.ignore_id,
                                                    float(n_tok))
        ppl = torch.exp(ce.detach())
        # -1: eos, which is removed in the loss computation (decoders.py:271-272)
        self.loss = F_.ScaleFn.apply(ce, float(np.mean([len(x) for x in ys_in]) - 1))
        acc = th_accuracy(correct, ys_out_pad, self.ignore_id)
        return self.loss, acc, ppl


def decoder_for(args, odim, sos, eos, att, labeldist):
    """reference: rnn/decoders.py:1199-1218"""
    return Decoder(args.eprojs, odim, args.dtype, args.dlayers, args.dunits, sos, eos, att, args.verbose,
                   args.char_list, labeldist, args.lsm_weight, args.sampling_probability, args.dropout_rate_decoder,
                   getattr(args, "context_residual", False), getattr(args, "replace_sos", False),
                   getattr(args, "num_encs", 1))
