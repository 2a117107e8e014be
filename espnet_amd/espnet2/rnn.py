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


# ---- decoder -------------------------------------------------------------------------------------------------------
def build_attention_list(eprojs, dunits, atype="location", num_att=1, num_encs=1, aheads=4, adim=320, awin=5,
                         aconv_chans=10, aconv_filts=100, han_mode=False, han_type=None, han_heads=4, han_dim=320,
                         han_conv_chans=-1, han_conv_filts=100, han_win=5):
    """reference: espnet2/asr/decoder/rnn_decoder.py:16-81 (single-encoder case)"""
    from ..nets.rnn.attentions import initial_att
    if num_encs != 1:
        raise NotImplementedError("multi-encoder attention is out of the hot-path scope")
    att_list = torch.nn.ModuleList()
    for _ in range(num_att):
        att_list.append(initial_att(atype, eprojs, dunits, aheads, adim, awin, aconv_chans, aconv_filts))
    return att_list


from ..nets.rnn.decoders import GRUCell, LSTMCell  # noqa: E402
from ..nets.scorer_interface import ScorerInterface  # noqa: E402
from .. import rnn_functional as R_  # noqa: E402
from .asr import AbsDecoder  # noqa: E402


class RNNDecoder(AbsDecoder, ScorerInterface):
    """attention LSTM / GRU decoder with the espnet2 call signature.  reference: espnet2/asr/decoder/rnn_decoder.py:84-334
    (forward returns the output-layer logits with padded positions zeroed; init_state / score for BeamSearch)."""

    def __init__(self, vocab_size, encoder_output_size, rnn_type="lstm", num_layers=1, hidden_size=320,
                 sampling_probability=0.0, dropout=0.0, context_residual=False, replace_sos=False, num_encs=1,
                 att_conf=None):
        if rnn_type not in {"lstm", "gru"}:
            raise ValueError(f"Not supported: rnn_type={rnn_type}")
        if num_encs != 1 or replace_sos or sampling_probability > 0.0:
            raise NotImplementedError("multi-encoder / replace_sos / scheduled sampling are outside the hot-path scope")
        super().__init__()
        eprojs = encoder_output_size
        self.dtype, self.dunits, self.dlayers = rnn_type, hidden_size, num_layers
        self.context_residual = context_residual
        self.sos = self.eos = vocab_size - 1
        self.odim = vocab_size
        self.sampling_probability, self.dropout, self.num_encs, self.replace_sos = sampling_probability, dropout, num_encs, replace_sos
        self.embed = torch.nn.Embedding(vocab_size, hidden_size)
        cell = LSTMCell if rnn_type == "lstm" else GRUCell
        self.decoder = torch.nn.ModuleList([cell(hidden_size + eprojs, hidden_size)]
                                           + [cell(hidden_size, hidden_size) for _ in range(1, num_layers)])
        self.output = torch.nn.Linear(hidden_size + eprojs if context_residual else hidden_size, vocab_size)
        self.att_list = build_attention_list(eprojs=eprojs, dunits=hidden_size, **(att_conf or {}))
        self.salt_emb = ops.new_salt()
        self.salts = [ops.new_salt() for _ in range(num_layers + 1)]

    def zero_state(self, hs_pad):
        return hs_pad.new_zeros(hs_pad.size(0), self.dunits)

    def _drop(self, k, x, step):
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

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens, strm_idx=0):
        """-> (logits (B, L, vocab) with positions >= ys_in_lens zeroed, ys_in_lens)"""
        att = self.att_list[min(strm_idx, len(self.att_list) - 1)]
        hl, yl = _lens(hlens), _lens(ys_in_lens)
        B, olength = ys_in_pad.shape
        c_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        z_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        att_w = None
        att.reset()
        eys = F_.dropout(R_.PlainEmbedFn.apply(ys_in_pad, self.embed.weight, -1), self.dropout, self.salt_emb, self.training)
        z_all = []
        for i in range(olength):
            att_c, att_w = att(hs_pad, hl, self._drop(self.dlayers, z_list[0], i), att_w)
            ey = torch.cat((eys[:, i, :], att_c), dim=1)
            z_list, c_list = self.rnn_forward(ey, z_list, c_list, z_list, c_list, step=i)
            top = self._drop(self.dlayers - 1, z_list[-1], i + olength)
            z_all.append(torch.cat((top, att_c), dim=-1) if self.context_residual else top)
        z_all = torch.stack(z_all, dim=1)
        logits = F_.LinearFn.apply(z_all.view(B * olength, -1), self.output.weight, self.output.bias)
        keep = ops.h2d_cached("keep", ~make_pad_mask(yl, olength).numpy().reshape(-1, 1), logits.device)
        return F_.MaskRowsFn.apply(logits, keep).view(B, olength, -1), ys_in_lens

    # ---- ScorerInterface (rnn_decoder.py:236-334) ----------------------------------------------------------------------
    def init_state(self, x):
        xs = x.unsqueeze(0)
        c_list = [self.zero_state(xs) for _ in range(self.dlayers)]
        z_list = [self.zero_state(xs) for _ in range(self.dlayers)]
        self.att_list[0].reset()
        return dict(c_prev=c_list[:], z_prev=z_list[:], a_prev=None, workspace=(0, z_list, c_list))

    def score(self, yseq, state, x):
        att_idx, z_list, c_list = state["workspace"]
        vy = yseq[-1].unsqueeze(0).to(x.device)
        ey = R_.PlainEmbedFn.apply(vy, self.embed.weight, -1)
        att_c, att_w = self.att_list[att_idx](x.unsqueeze(0), [x.size(0)], state["z_prev"][0], state["a_prev"])
        ey = torch.cat((ey, att_c), dim=1)
        z_list, c_list = self.rnn_forward(ey, [None] * self.dlayers, [None] * self.dlayers, state["z_prev"], state["c_prev"])
        top = torch.cat((z_list[-1], att_c), dim=-1) if self.context_residual else z_list[-1]
        logits = F_.LinearFn.apply(top, self.output.weight, self.output.bias)
        logp = ops.log_softmax_rows(logits.contiguous()).squeeze(0)
        return logp, dict(c_prev=c_list[:], z_prev=z_list[:], a_prev=att_w, workspace=(att_idx, z_list, c_list))
