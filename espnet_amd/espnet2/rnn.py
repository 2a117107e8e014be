"""espnet2 RNN encoders on the HIP kernels.

reference: espnet2/asr/encoder/rnn_encoder.py:14-113 (RNNEncoder), espnet2/asr/encoder/vgg_rnn_encoder.py:15-104
(VGGRNNEncoder): the espnet1 RNN / RNNP / VGG2L stacks of espnet_amd.nets.rnn.encoders behind the AbsEncoder call
signature, same constructor arguments and parameter names (`enc.0.birnn0.weight_ih_l0`, ...).
"""
import numpy as np
import torch

from .. import functional as F_
from .. import ops
from ..nets.modules import make_pad_mask
from ..nets.rnn.encoders import RNN, RNNP, VGG2L, get_vgg2l_odim
from .asr import AbsEncoder, _lens


class _RNNEncoderBase(AbsEncoder):
    def output_size(self):
        return self._output_size

    def forward(self, xs_pad, ilens, prev_states=None):
        """xs_pad (B, T, D), ilens (B,) -> (states (B, T', output_size) with padded frames zeroed, lengths, None)"""
        assert prev_states is None, "streaming states are not on the training path"
        il = _lens(ilens)
        for module in self.enc:
            xs_pad, il, _ = module(xs_pad, il)
        keep = ops.h2d_cached("keep", ~make_pad_mask(il, xs_pad.shape[1]).numpy(), xs_pad.device).unsqueeze(-1)
        olens = torch.tensor(il, dtype=torch.int64).to(xs_pad.device)
        return F_.MaskRowsFn.apply(xs_pad, keep), olens, [None] * len(self.enc)


class RNNEncoder(_RNNEncoderBase):
    def __init__(self, input_size, rnn_type="lstm", bidirectional=True, use_projection=True, num_layers=4,
                 hidden_size=320, output_size=320, dropout=0.0, subsample=(2, 2, 1, 1)):
        super().__init__()
        self._output_size = output_size
        self.rnn_type, self.bidirectional, self.use_projection = rnn_type, bidirectional, use_projection
        if rnn_type not in {"lstm", "gru"}:
            raise ValueError(f"Not supported rnn_type={rnn_type}")
        if subsample is None:
            subsample = np.ones(num_layers + 1, dtype=np.int64)
        else:
            subsample = list(subsample)[:num_layers]
            # the first entry belongs to the (absent) input layer; layers beyond the given list do not subsample
            subsample = np.pad(np.array(subsample, dtype=np.int64), [1, num_layers - len(subsample)], mode="constant",
                               constant_values=1)
        typ = ("b" if bidirectional else "") + rnn_type
        if use_projection:
            self.enc = torch.nn.ModuleList([RNNP(input_size, num_layers, hidden_size, output_size, subsample, dropout,
                                                 typ=typ)])
        else:
            self.enc = torch.nn.ModuleList([RNN(input_size, num_layers, hidden_size, output_size, dropout, typ=typ)])


class VGGRNNEncoder(_RNNEncoderBase):
    def __init__(self, input_size, rnn_type="lstm", bidirectional=True, use_projection=True, num_layers=4,
                 hidden_size=320, output_size=320, dropout=0.0, in_channel=1):
        super().__init__()
        self._output_size = output_size
        self.rnn_type, self.bidirectional, self.use_projection = rnn_type, bidirectional, use_projection
        if rnn_type not in {"lstm", "gru"}:
            raise ValueError(f"Not supported rnn_type={rnn_type}")
        subsample = np.ones(num_layers + 1, dtype=np.int64)      # VGG2L already subsamples by 4
        typ = ("b" if bidirectional else "") + rnn_type
        vgg_odim = get_vgg2l_odim(input_size, in_channel=in_channel)
        if use_projection:
            rnn = RNNP(vgg_odim, num_layers, hidden_size, output_size, subsample, dropout, typ=typ)
        else:
            rnn = RNN(vgg_odim, num_layers, hidden_size, output_size, dropout, typ=typ)
        self.enc = torch.nn.ModuleList([VGG2L(in_channel), rnn])
