"""Language models for shallow fusion in the joint decoding loop, on the HIP kernels.

reference: espnet2/lm/transformer_lm.py:14-133 (TransformerLM: Embedding -> Encoder(input_layer="linear")
-> Linear; `batch_score` with per-layer caches), espnet2/lm/seq_rnn_lm.py:13-174 (SequentialRNNLM:
Embedding -> LSTM/GRU stack -> Linear; `batch_score` carrying (h, c)), espnet2/lm/abs_model.py.
Same constructor arguments, state_dict keys and scorer methods.
"""
import torch

from .. import functional as F_
from .. import ops
from .. import rnn_functional as R_
from ..nets.modules import PositionalEncoding, TransformerEncoder, subsequent_mask
from ..nets.scorer_interface import BatchScorerInterface


try:  # pragma: no cover - only when the reference package is importable
    from espnet2.lm.abs_model import AbsLM  # type: ignore
except Exception:  # noqa: BLE001

    class AbsLM(torch.nn.Module, BatchScorerInterface):
        """reference: espnet2/lm/abs_model.py:11-32"""

        def forward(self, input, hidden):
            raise NotImplementedError


def register_lm_choices(lm_task_module):
    """Add the HIP language models to the reference's registry (espnet2/tasks/lm.py:31-39):
    `--lm transformer_mi355x` / `--lm seq_rnn_mi355x`."""
    lm_task_module.lm_choices.classes["transformer_mi355x"] = TransformerLM
    lm_task_module.lm_choices.classes["seq_rnn_mi355x"] = SequentialRNNLM


class _NoPosEnc(torch.nn.Sequential):
    def forward(self, x):
        return x


class TransformerLM(AbsLM):
    def __init__(self, vocab_size, pos_enc=None, embed_unit=128, att_unit=256, head=2, unit=1024, layer=4,
                 dropout_rate=0.5):
        super().__init__()
        if pos_enc == "sinusoidal":
            pos_enc_class = PositionalEncoding
        elif pos_enc is None:
            def pos_enc_class(*args, **kwargs):
                return _NoPosEnc()
        else:
            raise ValueError(f"unknown pos-enc option: {pos_enc}")
        self.embed = torch.nn.Embedding(vocab_size, embed_unit)
        self.encoder = TransformerEncoder(idim=embed_unit, attention_dim=att_unit, attention_heads=head,
                                          linear_units=unit, num_blocks=layer, dropout_rate=dropout_rate,
                                          input_layer="linear", pos_enc_class=pos_enc_class)
        self.decoder = torch.nn.Linear(att_unit, vocab_size)

    def _target_mask(self, ys_in_pad):
        ys_mask = ys_in_pad != 0
        m = subsequent_mask(ys_mask.size(-1), device=ys_mask.device).unsqueeze(0)
        return ys_mask.unsqueeze(-2) & m

    def _emb(self, tokens):
        return R_.PlainEmbedFn.apply(tokens, self.embed.weight, -1)

    def forward(self, input, hidden=None):
        """input (B, L) token ids -> (logits (B, L, V), None)"""
        h, _ = self.encoder(self._emb(input), self._target_mask(input))
        return F_.LinearFn.apply(h, self.decoder.weight, self.decoder.bias), None

    def score(self, y, state, x):
        y = y.unsqueeze(0)
        h, _, cache = self.encoder.forward_one_step(self._emb(y), self._target_mask(y), cache=state)
        logits = F_.LinearFn.apply(h[:, -1].contiguous(), self.decoder.weight, self.decoder.bias)
        return ops.log_softmax_rows(logits.contiguous()).squeeze(0), cache

    def score_tree(self, ys, tree, xs):
        """batch_score on a batched cache (list per layer of [n, L-1, D], or None)"""
        h, _, new = self.encoder.forward_one_step(self._emb(ys), self._target_mask(ys), cache=tree)
        logits = F_.LinearFn.apply(h[:, -1].contiguous(), self.decoder.weight, self.decoder.bias)
        return ops.log_softmax_rows(logits.contiguous()), new

    def final_tree(self, tree):
        return 0.0

    def batch_score(self, ys, states, xs):
        n_batch, n_layers = len(ys), len(self.encoder.encoders)
        batch_state = None if states[0] is None else \
            [torch.stack([states[b][i] for b in range(n_batch)]) for i in range(n_layers)]
        h, _, new = self.encoder.forward_one_step(self._emb(ys), self._target_mask(ys), cache=batch_state)
        logits = F_.LinearFn.apply(h[:, -1].contiguous(), self.decoder.weight, self.decoder.bias)
        logp = ops.log_softmax_rows(logits.contiguous())
        return logp, [[new[i][b] for i in range(n_layers)] for b in range(n_batch)]


class SequentialRNNLM(AbsLM):
    def __init__(self, vocab_size, unit=650, nhid=None, nlayers=2, dropout_rate=0.0, tie_weights=False,
                 rnn_type="lstm", ignore_id=0):
        super().__init__()
        ninp = unit
        nhid = unit if nhid is None else nhid
        rnn_type = rnn_type.upper()
        if rnn_type not in ("LSTM", "GRU"):
            raise NotImplementedError("rnn_type %r: LSTM / GRU recurrences are on the HIP path" % rnn_type)
        self.drop = torch.nn.Dropout(dropout_rate)
        self.encoder = torch.nn.Embedding(vocab_size, ninp, padding_idx=ignore_id)
        # parameter container only: torch.nn.LSTM's names, its forward is never called
        self.rnn = getattr(torch.nn, rnn_type)(ninp, nhid, nlayers, dropout=dropout_rate, batch_first=True)
        self.decoder = torch.nn.Linear(nhid, vocab_size)
        if tie_weights:
            if nhid != ninp:
                raise ValueError("When using the tied flag, nhid must be equal to emsize")
            self.decoder.weight = self.encoder.weight
        self.rnn_type, self.nhid, self.nlayers = rnn_type, nhid, nlayers
        self.salts = [ops.new_salt() for _ in range(nlayers + 1)]

    def _p(self, name, k):
        return getattr(self.rnn, "%s_l%d" % (name, k))

    def _step(self, x, hidden):
        """one token: x (B, ninp), hidden (h[, c]) each (nlayers, B, nhid) -> (y (B, nhid), new hidden)"""
        lstm = self.rnn_type == "LSTM"
        hs, cs = [], []
        for k in range(self.nlayers):
            gx = F_.LinearFn.apply(x, self._p("weight_ih", k), self._p("bias_ih", k))
            if lstm:
                x, c = R_.LSTMCellFn.apply(gx, hidden[0][k].contiguous(), hidden[1][k].contiguous(),
                                           self._p("weight_hh", k), self._p("bias_hh", k))
                cs.append(c)
            else:
                x = R_.GRUCellFn.apply(gx, hidden[k].contiguous(), self._p("weight_hh", k), self._p("bias_hh", k))
            hs.append(x)
            if k < self.nlayers - 1:
                x = F_.dropout(x, self.rnn.dropout, self.salts[k], self.training)
        return x, ((torch.stack(hs), torch.stack(cs)) if lstm else torch.stack(hs))

    def forward(self, input, hidden=None):
        """input (B, L) -> (logits (B, L, V), hidden); seq_rnn_lm.py:81-96"""
        B, L = input.shape
        dev = input.device
        emb = R_.PlainEmbedFn.apply(input, self.encoder.weight, self.encoder.padding_idx)
        emb = F_.dropout(emb, self.drop.p, self.salts[-1], self.training)
        if hidden is None:
            z = torch.zeros(self.nlayers, B, self.nhid, device=dev)
            hidden = (z, z.clone()) if self.rnn_type == "LSTM" else z
        outs = []
        for t in range(L):
            y, hidden = self._step(emb[:, t].contiguous(), hidden)
            outs.append(y)
        out = F_.dropout(torch.stack(outs, dim=1), self.drop.p, self.salts[-1] + 1, self.training)
        return F_.LinearFn.apply(out, self.decoder.weight, self.decoder.bias), hidden

    def score(self, y, state, x):
        y, new_state = self(y[-1].view(1, 1), state)
        return ops.log_softmax_rows(y.view(1, -1).contiguous()).view(-1), new_state

    def score_tree(self, ys, tree, xs):
        """batch_score on a batched state with the hypothesis axis FIRST: (h, c) each (n, nlayers, nhid), or h alone, or None"""
        lstm = self.rnn_type == "LSTM"
        hidden = None
        if tree is not None:
            hidden = tuple(t.transpose(0, 1).contiguous() for t in tree) if lstm else tree.transpose(0, 1).contiguous()
        y, hidden = self(ys[:, -1:], hidden)
        logp = ops.log_softmax_rows(y.squeeze(1).contiguous())
        return logp, (tuple(t.transpose(0, 1) for t in hidden) if lstm else hidden.transpose(0, 1))

    def final_tree(self, tree):
        return 0.0

    def batch_score(self, ys, states, xs):
        """reference: seq_rnn_lm.py:126-174; state of one hypothesis = (h, c) each (nlayers, nhid)"""
        lstm = self.rnn_type == "LSTM"
        if states[0] is None:
            hidden = None
        elif lstm:
            hidden = (torch.stack([h for h, c in states], dim=1), torch.stack([c for h, c in states], dim=1))
        else:
            hidden = torch.stack(states, dim=1)
        y, hidden = self(ys[:, -1:], hidden)
        logp = ops.log_softmax_rows(y.squeeze(1).contiguous())
        if lstm:
            h, c = hidden
            return logp, [(h[:, i], c[:, i]) for i in range(h.size(1))]
        return logp, [hidden[:, i] for i in range(hidden.size(1))]
