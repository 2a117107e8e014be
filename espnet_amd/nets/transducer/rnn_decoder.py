"""reference: espnet/nets/pytorch_backend/transducer/rnn_decoder.py:10-166 (DecoderRNNT, lstm)"""
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_
from ..rnn.decoders import GRUCell, LSTMCell
from .joint_network import JointNetwork


class DecoderRNNT(torch.nn.Module):
    """Prediction network + joint network.  In training the label history is known up front, so each
    LSTM layer runs as one input-projection GEMM over all U steps followed by the recurrence
    (LSTMSeqFn) instead of U separate LSTMCell calls; the result is the same sequence h_dec (B,U,D)."""

    def __init__(self, eprojs, odim, dtype, dlayers, dunits, blank, embed_dim, joint_dim,
                 joint_activation_type="tanh", dropout=0.0, dropout_embed=0.0):
        super().__init__()
        if dtype not in ("lstm", "gru"):
            raise NotImplementedError("dtype %r: lstm and gru prediction networks have HIP kernels" % dtype)
        self.embed = torch.nn.Embedding(odim, embed_dim, padding_idx=blank)
        cell = LSTMCell if dtype == "lstm" else GRUCell
        self.decoder = torch.nn.ModuleList([cell(embed_dim, dunits)] + [cell(dunits, dunits) for _ in range(1, dlayers)])
        self.joint_network = JointNetwork(odim, eprojs, dunits, joint_dim, joint_activation_type)
        self.dlayers, self.dunits, self.dtype = dlayers, dunits, dtype
        self.embed_dim, self.joint_dim, self.odim = embed_dim, joint_dim, odim
        self.dropout, self.dropout_embed_rate = dropout, dropout_embed
        self.ignore_id = -1
        self.blank = blank
        self.salt_emb = ops.new_salt()
        self.salts = [ops.new_salt() for _ in range(dlayers)]

    def init_state(self, init_tensor):
        z = [init_tensor.new_zeros(init_tensor.size(0), self.dunits) for _ in range(self.dlayers)]
        c = [init_tensor.new_zeros(init_tensor.size(0), self.dunits) for _ in range(self.dlayers)]
        return (z, c)

    def rnn_forward(self, ey, state):
        """single step (decoding): ey (B, emb) -> (y (B, dunits), new state)   (rnn_decoder.py:106-138)"""
        z_prev, c_prev = state
        z_list, c_list = self.init_state(ey)
        if self.dtype == "lstm":
            z_list[0], c_list[0] = self.decoder[0](ey, (z_prev[0], c_prev[0]))
        else:
            z_list[0] = self.decoder[0](ey, z_prev[0])
        for i in range(1, self.dlayers):
            x = F_.dropout(z_list[i - 1], self.dropout, self.salts[i - 1], self.training)
            if self.dtype == "lstm":
                z_list[i], c_list[i] = self.decoder[i](x, (z_prev[i], c_prev[i]))
            else:
                z_list[i] = self.decoder[i](x, z_prev[i])
        y = F_.dropout(z_list[-1], self.dropout, self.salts[-1], self.training)
        return y, (z_list, c_list)

    # ---- decoding: batched single-step protocol of espnet_amd.nets.beam_search_transducer ----------------------------
    # A per-hypothesis state is a pair of (dlayers, dunits) tensors (hidden, cell); the searches keep one per label
    # prefix and hand any number of them to `step` as ONE batch through the cell kernels.
    def batch_states(self, states):
        """[(z (dlayers, dunits), c (dlayers, dunits))] * n -> the (z_list, c_list) form of rnn_forward, batch n"""
        z = torch.stack([s[0] for s in states])
        c = torch.stack([s[1] for s in states])
        return ([z[:, i].contiguous() for i in range(self.dlayers)], [c[:, i].contiguous() for i in range(self.dlayers)])

    def unbatch_state(self, state, idx):
        return (torch.stack([z[idx] for z in state[0]]), torch.stack([c[idx] for c in state[1]]))

    def step(self, tokens, state):
        """tokens (n,) int64 = the last label of n prefixes, state = batch_states(...) BEFORE those labels
        -> (y (n, dunits), state after)"""
        ey = R_.PlainEmbedFn.apply(tokens.view(-1), self.embed.weight, self.blank)
        return self.rnn_forward(ey, state)

    def score(self, hyp, cache, init_tensor=None):
        """the reference's per-hypothesis plug-in method (TransducerDecoderInterface.score, rnn_decoder.py:168-193):
        hyp.yseq / hyp.dec_state -> (y (1, dunits), state, last token (1,)), cached by the label prefix"""
        key = tuple(hyp.yseq)
        tok = torch.tensor([hyp.yseq[-1]], dtype=torch.long, device=self.embed.weight.device)
        if key not in cache:
            cache[key] = self.step(tok, hyp.dec_state)
        y, state = cache[key]
        return y, state, tok

    def forward(self, hs_pad, ys_in_pad, hlens=None):
        """hs_pad (B,Tmax,D), ys_in_pad (B,Lmax+1) -> joint logits (B,T,U,odim)   (rnn_decoder.py:140-166)"""
        return self.joint_network(hs_pad, self.hidden(ys_in_pad))

    def hidden(self, ys_in_pad):
        """prediction network over the whole label history: ys_in_pad (B,Lmax+1) -> h_dec (B,U,dunits)"""
        eys = R_.PlainEmbedFn.apply(ys_in_pad, self.embed.weight, self.blank)
        eys = F_.dropout(eys, self.dropout_embed_rate, self.salt_emb, self.training)
        x = eys.transpose(0, 1).contiguous()                              # (U,B,emb) time-major
        for i, cell in enumerate(self.decoder):
            gx = F_.LinearFn.apply(x, cell.weight_ih, cell.bias_ih)
            x = (R_.LSTMSeqFn if self.dtype == "lstm" else R_.GRUSeqFn).apply(gx, cell.weight_hh, cell.bias_hh, None, False)
            x = F_.dropout(x, self.dropout, self.salts[i], self.training)
        return x.transpose(0, 1).contiguous()                             # (B,U,dunits)
