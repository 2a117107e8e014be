"""VGG / (B)LSTM(P) encoders.  reference: espnet/nets/pytorch_backend/rnn/encoders.py (same class names,
constructor arguments and parameter names; torch.nn.LSTM containers are replaced by `LSTM` below, whose
parameters carry torch.nn.LSTM's names so reference checkpoints load key-for-key).  LSTM and GRU cells, with
and without projection layers and frame subsampling, are on the HIP path."""
import math

import numpy as np
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_
from ..modules import make_pad_mask


class LSTM(torch.nn.Module):
    """single-layer (bi)directional LSTM (gates = 4) or GRU (gates = 3), batch_first, on padded batches + lengths
    (what pack_padded_sequence -> torch.nn.LSTM / GRU -> pad_packed_sequence computes, encoders.py:62-71)."""

    gates = 4
    seq_fn = R_.LSTMSeqFn

    def __init__(self, input_size, hidden_size, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        k = 1.0 / math.sqrt(hidden_size)
        g = self.gates
        for sfx in ([""] + (["_reverse"] if bidirectional else [])):
            for name, shape in (("weight_ih_l0", (g * hidden_size, input_size)),
                                ("weight_hh_l0", (g * hidden_size, hidden_size)),
                                ("bias_ih_l0", (g * hidden_size,)), ("bias_hh_l0", (g * hidden_size,))):
                self.register_parameter(name + sfx, torch.nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def flatten_parameters(self):
        pass

    def forward(self, xs_pad, ilens):
        """xs_pad (B,T,I), ilens list[int] -> (B, max(ilens), H or 2H), zeros after each length"""
        B = xs_pad.shape[0]
        T = int(max(ilens))
        x_tm = xs_pad[:, :T].transpose(0, 1).contiguous()                   # time-major for per-frame views
        live = None
        if min(ilens) < T:
            live = ops.h2d_cached("live", (np.arange(T)[:, None] < np.asarray(ilens)[None, :]).astype(np.uint8),
                                  xs_pad.device)
        outs = _directions(self.seq_fn, x_tm, live, self.bidirectional, lambda n, sfx: getattr(self, n + "_l0" + sfx))
        y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
        return y.transpose(0, 1).contiguous()


class GRU(LSTM):
    """torch.nn.GRU counterpart (gate order r, z, n)"""

    gates = 3
    seq_fn = R_.GRUSeqFn


def _directions(seq_fn, x_tm, live, bidir, param):
    """input projection + recurrence of one layer for the forward (and reverse) direction -> [y_fwd (, y_rev)].
    The two directions are independent chains of hundreds of small step launches (each far from filling the chip): the
    reverse one runs on a second stream (a parallel branch of a captured graph); autograd runs its backward there too."""
    def one(sfx, rev):
        gx = F_.LinearFn.apply(x_tm, param("weight_ih", sfx), param("bias_ih", sfx))
        return seq_fn.apply(gx, param("weight_hh", sfx), param("bias_hh", sfx), live, rev)
    sfxs = [("", False), ("_reverse", True)] if bidir else [("", False)]
    T, B = x_tm.shape[0], x_tm.shape[1]
    H = param("weight_hh", "").shape[1]
    if seq_fn is R_.LSTMSeqFn and x_tm.is_cuda and ops.lstm_seq_ok(len(sfxs), B, H):
        # all time steps of the layer - both directions side by side - in one persistent launch (csrc/lstm_seq.hip)
        flat = []
        for sfx, rev in sfxs:
            gx = F_.LinearFn.apply(x_tm, param("weight_ih", sfx), param("bias_ih", sfx))
            flat += [gx, param("weight_hh", sfx), param("bias_hh", sfx), rev]
        return list(R_.LSTMSeqGroupFn.apply(live, len(sfxs), *flat))
    if not bidir:
        return [one("", False)]
    if not (ops.TWO_STREAM_BIRNN and x_tm.is_cuda):
        return [one("", False), one("_reverse", True)]
    cur = torch.cuda.current_stream(x_tm.device)
    side = ops.side_stream(x_tm.device)
    side.wait_stream(cur)                       # the layer input is ready
    with torch.cuda.stream(side):
        y_rev = one("_reverse", True)
    y_fwd = one("", False)
    cur.wait_stream(side)
    y_rev.record_stream(cur)                    # consumed (concatenated) on the main stream
    x_tm.record_stream(side)
    return [y_fwd, y_rev]


def _lstm_or_raise(typ):
    if "lstm" not in typ and "gru" not in typ:
        raise NotImplementedError("etype %r: lstm / gru recurrent layers are on the HIP path" % typ)


class RNNP(torch.nn.Module):
    """reference: rnn/encoders.py:15-100"""

    def __init__(self, idim, elayers, cdim, hdim, subsample, dropout, typ="blstm"):
        super().__init__()
        _lstm_or_raise(typ)
        bidir = typ[0] == "b"
        for i in range(elayers):
            inputdim = idim if i == 0 else hdim
            cls = LSTM if "lstm" in typ else GRU
            setattr(self, "%s%d" % ("birnn" if bidir else "rnn", i), cls(inputdim, cdim, bidirectional=bidir))
            setattr(self, "bt%d" % i, torch.nn.Linear(2 * cdim if bidir else cdim, hdim))
        self.elayers, self.cdim, self.subsample, self.typ, self.bidir, self.dropout = \
            elayers, cdim, subsample, typ, bidir, dropout
        self.salts = [ops.new_salt() for _ in range(elayers)]

    def forward(self, xs_pad, ilens, prev_state=None):
        assert prev_state is None, "streaming states are not on the training path"
        ilens = [int(v) for v in ilens]
        for layer in range(self.elayers):
            rnn = getattr(self, ("birnn" if self.bidir else "rnn") + str(layer))
            ys_pad = rnn(xs_pad, ilens)
            sub = int(self.subsample[layer + 1])
            if sub > 1:
                ys_pad = ys_pad[:, ::sub].contiguous()
                ilens = [(i + 1) // sub for i in ilens]
            bt = getattr(self, "bt%d" % layer)
            xs_pad = F_.LinearFn.apply(ys_pad, bt.weight, bt.bias)
            if layer < self.elayers - 1:
                # encoders.py:98: tanh(F.dropout(x, p)) - functional dropout, active in eval mode too
                xs_pad = R_.ActFn.apply(F_.dropout(xs_pad, self.dropout, self.salts[layer], True), ops.ACT_TANH)
        return xs_pad, ilens, None


class RNN(torch.nn.Module):
    """reference: rnn/encoders.py:103-162 (stacked (B)LSTM + tanh(Linear)); parameters keep
    torch.nn.LSTM's `nbrnn.weight_ih_l{k}[_reverse]` names."""

    def __init__(self, idim, elayers, cdim, hdim, dropout, typ="blstm"):
        super().__init__()
        _lstm_or_raise(typ)
        bidir = typ[0] == "b"
        rnn_cls = torch.nn.LSTM if "lstm" in typ else torch.nn.GRU          # parameter container only
        self.seq_fn = R_.LSTMSeqFn if "lstm" in typ else R_.GRUSeqFn
        self.nbrnn = rnn_cls(idim, cdim, elayers, batch_first=True, dropout=dropout, bidirectional=bidir)
        self.l_last = torch.nn.Linear(cdim * 2 if bidir else cdim, hdim)
        self.typ, self.bidir, self.elayers, self.dropout = typ, bidir, elayers, dropout
        self.salts = [ops.new_salt() for _ in range(elayers)]

    def forward(self, xs_pad, ilens, prev_state=None):
        assert prev_state is None
        ilens = [int(v) for v in ilens]
        B = xs_pad.shape[0]
        T = int(max(ilens))
        x = xs_pad[:, :T].transpose(0, 1).contiguous()
        live = None
        if min(ilens) < T:
            live = ops.h2d_cached("live", (np.arange(T)[:, None] < np.asarray(ilens)[None, :]).astype(np.uint8),
                                  xs_pad.device)
        for k in range(self.elayers):
            outs = _directions(self.seq_fn, x, live, self.bidir,
                               lambda n, sfx, k=k: getattr(self.nbrnn, "%s_l%d%s" % (n, k, sfx)))
            x = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
            if k < self.elayers - 1:   # torch.nn.LSTM(dropout=p): between layers, training mode only
                x = F_.dropout(x, self.dropout, self.salts[k], self.training)
        ys = x.transpose(0, 1).contiguous()
        proj = R_.ActFn.apply(F_.LinearFn.apply(ys, self.l_last.weight, self.l_last.bias), ops.ACT_TANH)
        return proj, ilens, None


class VGG2L(torch.nn.Module):
    """reference: rnn/encoders.py:178-237"""

    def __init__(self, in_channel=1):
        super().__init__()
        if in_channel != 1:
            raise NotImplementedError("VGG2L HIP path: in_channel = 1")
        self.conv1_1 = torch.nn.Conv2d(in_channel, 64, 3, stride=1, padding=1)
        self.conv1_2 = torch.nn.Conv2d(64, 64, 3, stride=1, padding=1)
        self.conv2_1 = torch.nn.Conv2d(64, 128, 3, stride=1, padding=1)
        self.conv2_2 = torch.nn.Conv2d(128, 128, 3, stride=1, padding=1)
        self.in_channel = in_channel

    def forward(self, xs_pad, ilens, **kwargs):
        y = R_.VGG2LFn.apply(xs_pad, self.conv1_1.weight, self.conv1_1.bias, self.conv1_2.weight, self.conv1_2.bias,
                             self.conv2_1.weight, self.conv2_1.bias, self.conv2_2.weight, self.conv2_2.bias)
        ilens = [int(math.ceil(math.ceil(int(v) / 2) / 2)) for v in ilens]
        return y, ilens, None


def get_vgg2l_odim(idim, in_channel=3, out_channel=128):
    """reference: espnet/nets/e2e_asr_common.py:238-249"""
    idim = idim / in_channel
    idim = np.ceil(np.array(idim, dtype=np.float32) / 2)
    idim = np.ceil(np.array(idim, dtype=np.float32) / 2)
    return int(idim) * out_channel


class Encoder(torch.nn.Module):
    """reference: rnn/encoders.py:240-326"""

    def __init__(self, etype, idim, elayers, eunits, eprojs, subsample, dropout, in_channel=1):
        super().__init__()
        typ = etype.lstrip("vgg").rstrip("p")
        if typ not in ["lstm", "gru", "blstm", "bgru"]:
            raise ValueError("Error: need to specify an appropriate encoder architecture")
        mods = []
        if etype.startswith("vgg"):
            mods.append(VGG2L(in_channel))
            idim = get_vgg2l_odim(idim, in_channel=in_channel)
        if etype[-1] == "p":
            mods.append(RNNP(idim, elayers, eunits, eprojs, subsample, dropout, typ=typ))
        else:
            mods.append(RNN(idim, elayers, eunits, eprojs, dropout, typ=typ))
        self.enc = torch.nn.ModuleList(mods)

    def forward(self, xs_pad, ilens, prev_states=None):
        assert prev_states is None
        ilens = [int(v) for v in (ilens.tolist() if torch.is_tensor(ilens) else ilens)]
        for module in self.enc:
            xs_pad, ilens, _ = module(xs_pad, ilens)
        # encoders.py:323-325: zero the padded frames (projection biases leak into them otherwise)
        keep = ops.h2d_cached("keep", ~make_pad_mask(ilens, xs_pad.shape[1]).numpy(), xs_pad.device).unsqueeze(-1)
        return F_.MaskRowsFn.apply(xs_pad, keep), ilens, None


def encoder_for(args, idim, subsample):
    """reference: rnn/encoders.py:329-372 (single encoder)"""
    num_encs = getattr(args, "num_encs", 1)
    if num_encs != 1:
        raise NotImplementedError("multi-encoder mode is out of the hot-path scope")
    return Encoder(args.etype, idim, args.elayers, args.eunits, args.eprojs, subsample, args.dropout_rate)
