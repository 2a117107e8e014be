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

    def score(self, hyp, cache, init_tensor=None):
        """one prediction-network step for a hypothesis, cached by its label prefix (rnn_decoder.py:168-193)
        -> (y (1, dunits), state, last token (1,))"""
        dev = self.embed.weight.device
        vy = torch.full((1, 1), hyp.yseq[-1], dtype=torch.long, device=dev)
        str_yseq = "".join([str(x) for x in hyp.yseq])
        if str_yseq in cache:
            y, state = cache[str_yseq]
        else:
            ey = R_.PlainEmbedFn.apply(vy, self.embed.weight, self.blank)
            y, state = self.rnn_forward(ey[0], hyp.dec_state)
            cache[str_yseq] = (y, state)
        return y, state, vy[0]

    def batch_score(self, hyps, batch_states, cache, init_tensor=None):
        """one prediction-network step for all hypotheses that are not cached yet, as ONE batch through the cell
        kernels (rnn_decoder.py:197-257) -> (batch_y (n, dunits), batch_states, last tokens (n,))"""
        dev = self.embed.weight.device
        final_batch = len(hyps)
        tokens, process = [], []
        done = [None] * final_batch
        for i, hyp in enumerate(hyps):
            str_yseq = "".join([str(x) for x in hyp.yseq])
            if str_yseq in cache:
                done[i] = cache[str_yseq]
            else:
                tokens.append(hyp.yseq[-1])
                process.append((str_yseq, hyp.dec_state))
        if process:
            batch = len(process)
            tok = torch.tensor(tokens, dtype=torch.long).to(dev).view(batch)
            dec_state = self.init_state(torch.zeros((batch, self.dunits), device=dev))
            dec_state = self.create_batch_states(dec_state, [p[1] for p in process])
            ey = R_.PlainEmbedFn.apply(tok, self.embed.weight, self.blank)
            y, dec_state = self.rnn_forward(ey, dec_state)
        j = 0
        for i in range(final_batch):
            if done[i] is None:
                new_state = self.select_state(dec_state, j)
                done[i] = (y[j], new_state)
                cache[process[j][0]] = (y[j], new_state)
                j += 1
        batch_states = self.create_batch_states(batch_states, [d[1] for d in done])
        batch_y = torch.stack([d[0] for d in done])
        lm_tokens = torch.tensor([h.yseq[-1] for h in hyps], dtype=torch.long).to(dev).view(final_batch)
        return batch_y, batch_states, lm_tokens

    def select_state(self, batch_states, idx):
        """rnn_decoder.py:251-266"""
        return ([batch_states[0][layer][idx] for layer in range(self.dlayers)],
                [batch_states[1][layer][idx] for layer in range(self.dlayers)])

    def create_batch_states(self, batch_states, l_states, l_tokens=None):
        """rnn_decoder.py:268-288"""
        for layer in range(self.dlayers):
            batch_states[0][layer] = torch.stack([s[0][layer] for s in l_states])
            batch_states[1][layer] = torch.stack([s[1][layer] for s in l_states])
        return batch_states

    def forward(self, hs_pad, ys_in_pad, hlens=None):
        """hs_pad (B,Tmax,D), ys_in_pad (B,Lmax+1) -> joint logits (B,T,U,odim)   (rnn_decoder.py:140-166)"""
        eys = R_.PlainEmbedFn.apply(ys_in_pad, self.embed.weight, self.blank)
        eys = F_.dropout(eys, self.dropout_embed_rate, self.salt_emb, self.training)
        x = eys.transpose(0, 1).contiguous()                              # (U,B,emb) time-major
        for i, cell in enumerate(self.decoder):
            gx = F_.LinearFn.apply(x, cell.weight_ih, cell.bias_ih)
            x = (R_.LSTMSeqFn if self.dtype == "lstm" else R_.GRUSeqFn).apply(gx, cell.weight_hh, cell.bias_hh, None, False)
            x = F_.dropout(x, self.dropout, self.salts[i], self.training)
        h_dec = x.transpose(0, 1).contiguous()                            # (B,U,dunits)
        return self.joint_network(hs_pad, h_dec)
