"""espnet1 language models as beam-search scorers, on the HIP kernels.

reference: espnet/nets/pytorch_backend/lm/default.py:20-374 (DefaultRNNLM = ClassifierWithState(RNNLM): Embedding ->
LSTMCell / GRUCell stack -> Linear; state = {"c": [layer], "h": [layer]}), lm/transformer.py:18-189 (TransformerLM,
constructor (n_vocab, args); the network of espnet2's TransformerLM).  Same constructor arguments, state_dict keys
(`predictor.rnn.0.weight_ih`, ...), `forward(x, t) -> (loss, nll, count)` and scorer methods.
lm/seq_rnn.py's SequentialRNNLM only has the per-hypothesis `score`; its espnet2 twin (espnet2/lm.py) is the batched one.
"""
import torch

from .. import functional as F_
from .. import ops
from .. import rnn_functional as R_
from ..espnet2.lm import TransformerLM as _TransformerLM2
from .rnn.decoders import GRUCell, LSTMCell
from .scorer_interface import BatchScorerInterface


def _ce_sum(logits, target):
    """sum over rows of cross-entropy(logits, target), rows with target < 0 ignored"""
    loss, _ = F_.LabelSmoothingLossFn.apply(logits, target, 0.0, -1, 1.0)
    return loss


class RNNLM(torch.nn.Module):
    """reference: default.py:276-374"""

    def __init__(self, n_vocab, n_layers, n_units, n_embed=None, typ="lstm", dropout_rate=0.5):
        super().__init__()
        n_embed = n_units if n_embed is None else n_embed
        cell = LSTMCell if typ == "lstm" else GRUCell
        self.embed = torch.nn.Embedding(n_vocab, n_embed)
        self.rnn = torch.nn.ModuleList([cell(n_embed, n_units)] + [cell(n_units, n_units) for _ in range(n_layers - 1)])
        self.dropout = torch.nn.ModuleList([torch.nn.Dropout(dropout_rate) for _ in range(n_layers + 1)])
        self.lo = torch.nn.Linear(n_units, n_vocab)
        self.n_layers, self.n_units, self.typ = n_layers, n_units, typ
        self.salts = [ops.new_salt() for _ in range(n_layers + 1)]
        for param in self.parameters():
            param.data.uniform_(-0.1, 0.1)

    def zero_state(self, batchsize):
        p = next(self.parameters())
        return torch.zeros(batchsize, self.n_units, device=p.device, dtype=p.dtype)

    def forward(self, state, x):
        """state None | {"c": [..], "h": [..]}, x (B,) token ids -> (new state, logits (B, V))"""
        if state is None:
            state = {"h": [self.zero_state(x.size(0)) for _ in range(self.n_layers)]}
            if self.typ == "lstm":
                state["c"] = [self.zero_state(x.size(0)) for _ in range(self.n_layers)]
        y = R_.PlainEmbedFn.apply(x, self.embed.weight, -1)
        h, c = [None] * self.n_layers, [None] * self.n_layers
        for n in range(self.n_layers):
            y = F_.dropout(y, self.dropout[n].p, self.salts[n], self.training)
            if self.typ == "lstm":
                h[n], c[n] = self.rnn[n](y, (state["h"][n].contiguous(), state["c"][n].contiguous()))
            else:
                h[n] = self.rnn[n](y, state["h"][n].contiguous())
            y = h[n]
        y = F_.dropout(y, self.dropout[-1].p, self.salts[-1], self.training)
        new = {"c": c, "h": h} if self.typ == "lstm" else {"h": h}
        return new, F_.LinearFn.apply(y, self.lo.weight, self.lo.bias)


class ClassifierWithState(torch.nn.Module):
    """reference: default.py:172-273 (predictor + per-row cross entropy; `predict` = log-softmax of the logits)"""

    def __init__(self, predictor):
        super().__init__()
        self.predictor = predictor
        self.y = self.loss = None

    def forward(self, state, x, t):
        state, self.y = self.predictor(state, x)
        self.loss = _ce_sum(self.y.unsqueeze(1), t.view(-1, 1))      # sum over the batch of this position
        return state, self.loss

    def predict(self, state, x):
        state, z = self.predictor(state, x)
        return state, ops.log_softmax_rows(z.contiguous())

    def buff_predict(self, state, x, n):
        """reference: default.py:245-259 (an RNNLM predictor scores the whole batch at once)"""
        return self.predict(state, x)

    def final(self, state, index=None):
        return 0.0


class DefaultRNNLM(torch.nn.Module, BatchScorerInterface):
    def __init__(self, n_vocab, args):
        super().__init__()
        self.model = ClassifierWithState(RNNLM(n_vocab, args.layer, args.unit, getattr(args, "embed_unit", None),
                                               args.type, getattr(args, "dropout_rate", 0.0)))

    def state_dict(self, *a, **k):
        return self.model.state_dict(*a, **k)

    def load_state_dict(self, d, *a, **k):
        return self.model.load_state_dict(d, *a, **k)

    def forward(self, x, t):
        """reference: default.py:81-113, including its weighting: every position's batch-mean loss is multiplied by the
        number of non-zero inputs at that position, and `nll` by that count once more."""
        loss, count = 0, 0
        state = None
        batch_size, length = x.shape
        nz = (x != 0).sum(0).tolist()
        for i in range(length):
            state, loss_sum = self.model(state, x[:, i].contiguous(), t[:, i].contiguous())
            loss = loss + loss_sum * (nz[i] / batch_size)
            count += nz[i]
        return loss / batch_size, loss, torch.tensor(count, device=x.device)

    def score(self, y, state, x):
        new_state, scores = self.model.predict(state, y[-1].unsqueeze(0))
        return scores.squeeze(0), new_state

    def final_score(self, state):
        return self.model.final(state)

    def score_tree(self, ys, tree, xs):
        """batch_score on the merged state itself ({key: [layer tensors (n, n_units)]} or None)"""
        tree, logp = self.model.predict(tree, ys[:, -1].contiguous())
        return logp, tree

    def final_tree(self, tree):
        return self.model.final(tree)

    def batch_score(self, ys, states, xs):
        """reference: default.py:139-169; state of one hypothesis = {key: [layer tensors (n_units,)]}"""
        n_batch, n_layers = len(ys), self.model.predictor.n_layers
        keys = ("c", "h") if self.model.predictor.typ == "lstm" else ("h",)
        merged = None if states[0] is None else \
            {k: [torch.stack([states[b][k][i] for b in range(n_batch)]) for i in range(n_layers)] for k in keys}
        merged, logp = self.model.predict(merged, ys[:, -1].contiguous())
        return logp, [{k: [merged[k][i][b] for i in range(n_layers)] for k in keys} for b in range(n_batch)]


class TransformerLM(_TransformerLM2):
    """reference: lm/transformer.py:18-189"""

    def __init__(self, n_vocab, args):
        pos_enc = getattr(args, "pos_enc", "sinusoidal")
        super().__init__(n_vocab, pos_enc=None if pos_enc == "none" else pos_enc, embed_unit=args.embed_unit,
                         att_unit=args.att_unit, head=args.head, unit=args.unit, layer=args.layer,
                         dropout_rate=args.dropout_rate)

    def forward(self, x, t):
        """-> (mean nll over the non-zero inputs, summed nll, count); transformer.py:94-123"""
        y, _ = super().forward(x, None)
        tgt = torch.where(x != 0, t, torch.full_like(t, -1))
        logp = _ce_sum(y, tgt)
        count = (x != 0).sum()
        return logp / count, logp, count
